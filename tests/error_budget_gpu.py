"""GPU side of the round-5 error budget (test infrastructure: imports oracle/; run as a script on the GPU box).

For every consumer class of residual mode 3 (icd_unet option split_mask, ICD_SPLIT_* bits) switched on alone on top of the carry and
switched off alone from the default set: rel-L2 of eps against the fp32 oracle on the reduced SD1.5 / SDXL topologies (B = 2, 32 x 32,
t = 779, the cases of tests/test_unet_gpu.py) and, with --time, the step time of the benchmark workloads (bench.py's SD1.5 B = 32 and
SDXL B = 8 legs, round-robin minimum).  tests/error_budget_sim.py is the CPU prediction of the same table.

    python tests/error_budget_gpu.py [--time] > profiles/r05_error_budget_gpu.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BITS = [("gn", 1), ("conv1", 2), ("shortcut", 4), ("proj_out", 8), ("down", 16), ("sampler_out", 32), ("up", 64), ("temb", 128), ("qk", 256)]
DEFAULT = 447


def variants():
    v = [("carry only (mode 2)", 2, 0), ("default (mode 3)", 3, DEFAULT), ("everything", 3, 1023)]
    for n, b in BITS:
        m = b | (1 if b == 4 else 0)                      # the split shortcut needs the GroupNorm that writes lo
        v.append((f"carry + {n} alone", 3, m))
    for n, b in BITS:
        if DEFAULT & b:
            off = DEFAULT & ~b & (~4 if b == 1 else ~0)
            v.append((f"default without {n}", 3, off))
    return v


def eps_table():
    from error_budget_sim import case, rel_l2
    from oracle import unet_ref
    from invertible_cd_amd import unet, unet_config as uc
    rows = {}
    for which in ("sd15", "sdxl"):
        sd, o, lat, t, ctx, cond, added = case(which, False)
        ref = unet_ref.unet_forward(sd, o, lat, t, ctx, timestep_cond=cond, added_cond=added)
        cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64) if which == "sd15" else uc.SDXL.scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8))
        model = unet.UNet2DConditionModel(cfg, sd)
        kw = dict(encoder_hidden_states=ctx.cuda(), timestep_cond=cond.cuda(),
                  added_cond_kwargs=None if added is None else {k: v.cuda() for k, v in added.items()})
        for name, mode, mask in variants():
            model.set_option("residual", mode)
            model.set_option("split_mask", mask)
            eps = model(lat.half().cuda(), torch.tensor(t), **kw).sample
            rows.setdefault(name, {})[which] = rel_l2(eps.float().cpu(), ref)
    return rows


def time_table():
    import bench
    dev = torch.device("cuda:0")
    out = {}
    for which, wl, batch in (("sd15", bench.SD15Workload(dev), 32), ("sdxl", bench.SDXLWorkload(dev), 8)):
        step = wl.reverse_step(batch)
        vs = variants()
        for _, mode, mask in vs:
            wl.net.set_option("residual", mode); wl.net.set_option("split_mask", mask); step()
        best = {n: 1e9 for n, _, _ in vs}
        for _ in range(3):
            for n, mode, mask in vs:
                wl.net.set_option("residual", mode); wl.net.set_option("split_mask", mask); step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); step(); e1.record(); torch.cuda.synchronize()
                best[n] = min(best[n], e0.elapsed_time(e1))
        for n in best:
            out.setdefault(n, {})[which] = best[n]
        del wl, step
        torch.cuda.empty_cache()
    return out


def main():
    torch.cuda.set_device(0)
    eps = eps_table()
    tm = time_table() if "--time" in sys.argv else None
    base = eps["carry only (mode 2)"]
    print("# rel-L2 of eps vs the fp32 oracle (reduced topologies, B=2 32x32 t=779)" + ("; ms per 4-step pass of the bench workloads (min of 3)" if tm else ""))
    print(f"{'variant':34s} {'sd15 eps':>10s} {'x':>5s} {'sdxl eps':>10s} {'x':>5s}" + (f" {'sd15 ms':>9s} {'%':>6s} {'sdxl ms':>9s} {'%':>6s}" if tm else ""))
    for name, _, _ in variants():
        e = eps[name]
        line = f"{name:34s} {e['sd15']:10.3e} {e['sd15'] / base['sd15']:5.2f} {e['sdxl']:10.3e} {e['sdxl'] / base['sdxl']:5.2f}"
        if tm:
            t0 = tm["carry only (mode 2)"]
            line += f" {tm[name]['sd15']:9.2f} {(tm[name]['sd15'] / t0['sd15'] - 1) * 100:+6.2f} {tm[name]['sdxl']:9.2f} {(tm[name]['sdxl'] / t0['sdxl'] - 1) * 100:+6.2f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
