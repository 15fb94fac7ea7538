#!/usr/bin/env python3
"""Per-shape time breakdown of one UNet forward (HIP-event records of the executor): where do the milliseconds go?"""
import argparse
import collections
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import _lib, synthetic, unet
from invertible_cd_amd.unet_config import SD15, SDXL

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="sd15")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--xattn", type=int, default=2, help="UNet option xattn_fusion: 0 two launches, 1 fused wherever eligible, 2 (default) where faster")
ap.add_argument("--residual", type=int, default=2, help="UNet option residual: 0 fp16 stream, 1 fp32 twin, 2 error carry (default)")
ap.add_argument("--xattn-tile", type=int, default=0, help="2: force the 128x128 tile of the fused kernel, 4: the 256x128 tile")
a = ap.parse_args()
cfg = SD15 if a.arch == "sd15" else SDXL
res = 64 if a.arch == "sd15" else 128
sd = synthetic.synthetic_state_dict(cfg, seed=0, device="cuda", dtype=torch.float16)
m = unet.UNet2DConditionModel(cfg, sd)
m.set_option("xattn_fusion", 1 if a.xattn_tile else a.xattn)
m.set_option("residual", a.residual)
if a.xattn_tile:
    m.set_option("xattn_tile", a.xattn_tile)
del sd
inp = synthetic.synthetic_inputs(cfg, a.batch, res, res, device="cuda")
kw = dict(encoder_hidden_states=inp["context"].half(), timestep_cond=torch.randn(a.batch, 512, device="cuda").half())
if a.arch == "sdxl":
    kw["added_cond_kwargs"] = {"text_embeds": inp["text_embeds"].half(), "time_ids": inp["time_ids"]}
x = inp["latents"].half()
m(x, 999, **kw); torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(a.reps):
    m(x, 999, **kw)
torch.cuda.synchronize()
recs = _lib.profile_dump()
_lib.profile_enable(False)
agg = collections.OrderedDict()
for fam, M, N, K, aux, ms, fl in recs:
    k = (fam, M, N, K, aux)
    e = agg.setdefault(k, [0, 0.0, 0.0])
    e[0] += 1; e[1] += ms; e[2] += fl
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    m(x, 999, **kw)
torch.cuda.synchronize()
print(f"wall per forward (no events): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
tot = sum(e[1] for e in agg.values())
print(f"{a.arch} B={a.batch}: {tot / a.reps:.2f} ms of kernels per forward")
print(f"{'family':13s} {'M':>8s} {'N':>6s} {'K':>6s} {'aux':>5s} {'n/fwd':>6s} {'ms/fwd':>8s} {'us/call':>9s} {'TF/s':>7s} {'%':>6s}")
for (fam, M, N, K, aux), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = fl / (ms * 1e-3) / 1e12 if fl else 0
    print(f"{fam:13s} {M:8d} {N:6d} {K:6d} {aux:5d} {n / a.reps:6.1f} {ms / a.reps:8.3f} {ms / n * 1e3:9.1f} {tf:7.1f} {100 * ms / tot:6.2f}")
