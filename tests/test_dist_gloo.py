"""Multi-process data-parallel path on CPU (gloo, world_size 2): prompt sharding matches the reference's
prepare_val_prompts partitions and the single end-of-run all-gather reassembles the global order."""
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, golden_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from invertible_cd_amd import dist_utils
    dist_utils.init(backend="gloo")
    assert dist_utils.get_world_size() == world and dist_utils.get_rank() == rank
    gold = [e for e in json.load(open(os.path.join(golden_dir, "sharding.json"))) if e["W"] == world and e["rank"] == rank]
    ok = True
    for e in gold:
        texts = [f"p{i}" for i in range(e["N"])]
        batches, index, allt = dist_utils.prepare_val_prompts(texts, bs=e["bs"], max_cnt=5000)
        ok &= [list(map(int, b)) for b in index] == e["index"]
        ok &= [list(map(str, b)) for b in batches] == e["batches"]
    # each rank "generates" its shard (sample i is filled with the value i), then ONE gather + reorder
    texts = [f"p{i}" for i in range(32)]
    _, index, _ = dist_utils.prepare_val_prompts(texts, bs=4, max_cnt=5000)
    ids = torch.tensor(np.concatenate(index), dtype=torch.int64)
    local = ids.to(torch.float16).reshape(-1, 1, 1, 1).expand(-1, 4, 8, 8).contiguous()
    allx, alli = dist_utils.gather_samples(local, ids)
    ok &= alli.tolist() == list(range(32))
    ok &= bool((allx[:, 0, 0, 0].float() == torch.arange(32).float()).all())
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharding_and_gather_world2(golden_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_single_process_defaults(golden_dir):
    from invertible_cd_amd import dist_utils
    assert dist_utils.get_world_size() == 1 and dist_utils.get_rank() == 0
    for e in json.load(open(os.path.join(golden_dir, "sharding.json"))):
        if e["W"] == 1:
            b, idx, _ = dist_utils.prepare_val_prompts([f"p{i}" for i in range(e["N"])], bs=e["bs"])
            assert [list(map(int, x)) for x in idx] == e["index"]
    x, i = dist_utils.gather_samples(torch.arange(6).reshape(6, 1).float(), torch.tensor([3, 1, 2, 0, 5, 4]))
    assert i.tolist() == [0, 1, 2, 3, 4, 5] and x[:, 0].tolist() == [3.0, 1.0, 2.0, 0.0, 5.0, 4.0]


def test_single_process_init_ignores_a_stray_agent_store_flag():
    """A process started FROM a torchrun worker (bench.py's counter passes are) inherits TORCHELASTIC_USE_AGENT_STORE: env:// rendezvous
    would then be a client of a store nobody runs and sit out its timeout (measured: 970 s of timeouts on a default bench run under
    torchrun).  dist_utils.init hosts its own store whenever no launcher gave it an address."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(TORCHELASTIC_USE_AGENT_STORE="True", PYTHONPATH=root)
    code = ("import torch.distributed as dist\nfrom invertible_cd_amd import dist_utils\n"
            "dist_utils.init(backend='gloo', timeout_s=20)\nassert dist.get_world_size() == 1\ndist.barrier()\n"
            "dist.destroy_process_group()\nprint('ALONE_OK')")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ALONE_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from invertible_cd_amd import dist_utils
    dist_utils.init(backend="gloo", timeout_s=60)
    calls = []

    def step():                                   # rank r "produces" 3 samples per pass, rank 1 is the slower one
        calls.append(1)
        import time
        time.sleep(0.02 * (rank + 1))
        return torch.full((3, 4, 2, 2), float(rank))
    dt, per_rank, prof, last = bench.time_leg(step, steps=4, warmup=2, batch=3, device="cpu", world=world, rank=rank)
    ok = len(calls) == 6 and prof is None and len(per_rank) == world and dt == max(per_rank) and per_rank[1] > per_rank[0] * 0.9
    ok &= dt >= 4 * 0.04 * 0.9 and tuple(last.shape) == (3, 4, 2, 2)
    dist.barrier()
    q.put((rank, bool(ok), per_rank))
    dist.destroy_process_group()


def test_bench_timed_leg_world2_takes_the_max_over_ranks_and_gathers_every_sample():
    """bench.py's timed leg under a 2-rank gloo group: W untimed + exactly K timed passes per rank, ONE gather of batch * K * world
    samples with the ids in global order (asserted inside time_leg), the reported time is the MAX over ranks and every rank sees
    the same per-rank list (what the JSON line prints as ms_per_step_per_rank)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(0, True), (1, True)] and res[0][2] == res[1][2]


def _bench_inflight_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    import threading
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from invertible_cd_amd import dist_utils
    dist_utils.init(backend="gloo", timeout_s=60)
    lock, calls, threads = threading.Lock(), [], set()

    def make(tag):                                # one "executor replica" per step callable: 2 samples per pass, rank 1 is slower
        def step():
            with lock:
                calls.append(tag)
                threads.add(threading.get_ident())
            time.sleep(0.02 * (rank + 1))
            return torch.full((2, 4, 2, 2), float(rank))
        return step
    flight = bench.InFlight([make("a"), make("b")], "cpu")
    dt, per_rank, prof, last = bench.time_leg(flight, steps=6, warmup=2, batch=2, device="cpu", world=world, rank=rank)
    # 2 x 2 warm-up (W per batch in flight) + 6 timed passes per RANK, drawn dynamically by the two host threads of this rank (both must have worked); the gather
    # inside time_leg has asserted batch * K * world samples with ids in global order; two passes overlap, so 6 passes take ~3 sleeps
    ok = len(calls) == 10 and len(threads) == 2 and set(calls) == {"a", "b"} and prof is None and len(per_rank) == world and dt == max(per_rank)
    ok &= dt >= 3 * 0.04 * 0.9                    # (the slower rank's three rounds of two overlapped passes bound the leg from below)
    ok &= tuple(last.shape) == (2, 4, 2, 2)
    dist.barrier()
    q.put((rank, bool(ok), per_rank, len(calls), len(threads)))
    dist.destroy_process_group()


def test_bench_timed_leg_world2_with_two_batches_in_flight_per_rank():
    """What `bench.py --gpus N` runs since round 4 - ranks x host threads x executor replicas - before its first contact with N > 1 GPUs:
    time_leg driven by an InFlight of two step callables under a 2-rank gloo group.  Every rank runs exactly W + K passes split over its
    two threads, the barrier / max-over-ranks / all-gather bookkeeping is that of the plain leg, and every rank reports the same list."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_inflight_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(0, True), (1, True)], res
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] == 10 and res[0][4] == res[1][4] == 2


def test_dist_init_picks_a_free_port_alone_and_the_reference_default_for_a_multi_rank_run(monkeypatch):
    """A lone process takes a port that is free right now; WORLD_SIZE > 1 without MASTER_PORT (launchers that export only RANK /
    WORLD_SIZE / MASTER_ADDR) meets on the reference's fixed 29500 (utils/dist_utils.py:11-12) and says so - every rank must agree on
    the port, so nothing can be picked there."""
    from invertible_cd_amd import dist_utils
    import pytest
    for k in ("MASTER_PORT", "MASTER_ADDR", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    called = {}
    monkeypatch.setattr(dist, "init_process_group", lambda **kw: called.update(kw))       # (no second rank will ever come)
    with pytest.warns(UserWarning, match="29500"):
        dist_utils.init(backend="gloo")
    assert os.environ["MASTER_PORT"] == "29500" and called["backend"] == "gloo" and called["init_method"] == "env://"
    monkeypatch.undo()
    for k in ("MASTER_PORT", "MASTER_ADDR", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "7")            # a value the launcher chose is kept
    dist_utils.init(backend="gloo", timeout_s=30)
    try:
        assert int(os.environ["MASTER_PORT"]) > 1024 and os.environ["MASTER_PORT"] != "29500" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "7"
        assert dist.is_initialized() and dist.get_world_size() == 1
        # a world of one can still run the collective (what the GPU box does with backend "nccl"): same result as the shortcut
        x, ids = torch.arange(12.0).reshape(3, 4), torch.tensor([2, 0, 1])
        a, ai = dist_utils.gather_samples(x, ids)
        b, bi = dist_utils.gather_samples(x, ids, always_collective=True)
        assert torch.equal(a, b) and torch.equal(ai, bi) and torch.equal(ai, torch.arange(3))
    finally:
        dist.destroy_process_group()


def test_in_flight_runs_exactly_n_passes_in_submission_order_and_propagates_errors():
    """bench.InFlight (several batches in flight: one host thread per executor replica, all pulling passes from one counter) on CPU
    callables: exactly n passes, each output at the index of its ticket, a failing pass re-raised on the caller's thread; time_leg
    accepts it in place of a plain step."""
    import threading
    import pytest
    import bench
    seen, lock = [], threading.Lock()

    def make(tag):
        def step():
            with lock:
                seen.append(tag)
                k = len(seen)                    # (read under the lock: the passes run on three threads)
            return torch.full((2, 3), float(k))
        return step
    fl = bench.InFlight([make("a"), make("b"), make("c")], "cpu")
    outs = fl.run(7)
    assert len(outs) == 7 and len(seen) == 7 and all(o.shape == (2, 3) for o in outs)
    assert sorted(float(o[0, 0]) for o in outs) == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
    dt, per_rank, prof, last = bench.time_leg(fl, steps=4, warmup=1, batch=2, device="cpu", world=1, rank=0)
    assert dt > 0 and per_rank == [dt] and prof is None and last.shape == (2, 3)

    def boom():
        raise ValueError("pass failed")
    import time

    def slow():                                   # (leaves the other thread time to draw a ticket: passes are handed out dynamically)
        time.sleep(0.05)
        return torch.zeros(2, 3)
    with pytest.raises(ValueError, match="pass failed"):
        bench.InFlight([slow, boom], "cpu").run(8)
    with pytest.raises(AssertionError):
        bench.time_leg(fl, 2, 0, 2, "cpu", 1, 0, events_family="gemm_conv")     # the event profiler serves one batch at a time
