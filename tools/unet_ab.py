"""Same-box A/B of per-handle UNet options on the benchmark workloads (bench.py's SD1.5 / SDXL legs), as a round-robin.

    python tools/unet_ab.py --arch sd15 --batch 32 --opt residual=0,1,2 [--opt xattn_fusion=0,2] [--rounds 3 --iters 2] [--leg reverse|edit]

Every combination of the listed option values is one variant.  Each round runs every variant `iters` times (one pass = the whole
4-step loop of the leg); minimum and median over rounds are reported (whatever runs first in a fresh process is 10 - 15 % slow while
the clocks settle, so sequential A/Bs lie).  With --families the per-family split of one pass (HIP events around every launch) is
printed per variant.  ICD_AMD_LIB=path selects another build of the library."""
import argparse
import itertools
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="sd15", choices=["sd15", "sdxl"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--leg", default="reverse", choices=["reverse", "edit"])
    ap.add_argument("--opt", action="append", default=[], help="name=v0,v1,... (UNet2DConditionModel.set_option)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--families", action="store_true")
    a = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    wl = bench.SD15Workload(dev) if a.arch == "sd15" else bench.SDXLWorkload(dev)
    batch = a.batch or (32 if a.arch == "sd15" else 8)
    step = wl.reverse_step(batch) if a.leg == "reverse" else wl.edit_step(batch)
    names = [o.split("=")[0] for o in a.opt]
    values = [[int(v, 0) for v in o.split("=")[1].split(",")] for o in a.opt]
    variants = [dict(zip(names, combo)) for combo in itertools.product(*values)] or [{}]

    def apply(v):
        for k, x in v.items():
            wl.net.set_option(k, x)

    for v in variants:                                   # warm-up: allocator, arena of every variant
        apply(v)
        step()
    torch.cuda.synchronize()
    times = {i: [] for i in range(len(variants))}
    for _ in range(a.rounds):
        for i, v in enumerate(variants):
            apply(v)
            step()                                        # one untimed pass after the switch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                step()
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / a.iters)
    base = min(times[0])
    for i, v in enumerate(variants):
        mn, md = min(times[i]), statistics.median(times[i])
        print(f"{a.arch} B={batch} {a.leg} {v}: min {mn:8.3f} ms  median {md:8.3f} ms  {batch / mn * 1e3:8.2f} images/s  "
              f"({(mn / base - 1) * 100:+.2f} % time vs the first variant)", flush=True)
    if a.families:
        for v in variants:
            apply(v)
            step()
            fam, _ = bench.family_table(step)
            print(f"  families {v}: " + ", ".join(f"{k} {x['ms']:.2f} ms / {x['launches']}" for k, x in fam.items()), flush=True)


if __name__ == "__main__":
    main()
