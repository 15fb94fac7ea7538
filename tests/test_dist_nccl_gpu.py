"""The N > 1 path on real GPUs: one process per GPU, backend "nccl" (= RCCL on ROCm), the ONE end-of-run all-gather of
uint8 images + int64 ids (running/sd1.5/generate.py:372-397), and `bench.py --gpus 2` spawning its own ranks.
Skipped on a single-GPU box (the driver's multi-GPU tier and the CPU gloo test tests/test_dist_gloo.py cover it there)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on this node")

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["ICD_ROOT"])
import torch, torch.distributed as dist
from invertible_cd_amd import dist_utils
dist_utils.init("nccl")
r, W = dist.get_rank(), dist.get_world_size()
assert dist.get_backend() == "nccl" and torch.cuda.current_device() == int(os.environ["LOCAL_RANK"])
n = 3
ids = torch.arange(n, device="cuda", dtype=torch.int64) * W + r            # round-robin ownership, like prepare_val_prompts
imgs = (ids[:, None, None, None] % 251).to(torch.uint8).expand(n, 64, 64, 3).contiguous()
allx, alli = dist_utils.gather_samples(imgs, ids)
assert allx.dtype == torch.uint8 and allx.shape == (n * W, 64, 64, 3)
assert torch.equal(alli, torch.arange(n * W, device="cuda"))
assert torch.equal(allx[:, 0, 0, 0].long(), torch.arange(n * W, device="cuda") % 251)
dist.barrier()
if r == 0:
    print("NCCL_GATHER_OK", W)
dist.destroy_process_group()
'''


def _env():
    env = dict(os.environ)
    env["ICD_ROOT"] = ROOT
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)                 # bench.py / the tests below pick a free port
    return env


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@needs2
def test_world2_nccl_gather_of_uint8_images(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(w)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "NCCL_GATHER_OK 2" in r.stdout


@needs2
def test_bench_gpus_2_spawns_two_rccl_ranks():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-vae", "--no-ref-batching"]
    env = _env()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert d["rccl"]["rccl_world_size"] == 2 and d["ms_per_step_per_rank"]["ranks"] == 2


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stdout + r.stderr)
