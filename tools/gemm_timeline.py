#!/usr/bin/env python3
"""Where does one GEMM launch spend its time?  Per-block s_memrealtime stamps (icd_gemm_desc.debug_timeline) of a 256-wide-tile
launch: dispatch offset of each block, prologue (first k-tile landed), main loop, epilogue.

    python tools/gemm_timeline.py dense 8192 10240 1280 [--geglu] [--cfg 0]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from invertible_cd_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("kind")
ap.add_argument("M", type=int); ap.add_argument("N", type=int); ap.add_argument("K", type=int)
ap.add_argument("--geglu", action="store_true")
ap.add_argument("--cfg", type=int, default=0)
ap.add_argument("--gm", type=int, default=0)
a = ap.parse_args()
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).half()
M, N, K = a.M, a.N, a.K
x, w, b = rnd(M, K), rnd(N, K) * K ** -0.5, torch.zeros(N, device="cuda")
out = torch.empty((M, N // 2 if a.geglu else N), device="cuda", dtype=torch.float16)
res = None if a.geglu else rnd(M, N)
d = _lib.GemmDesc()
d.a0, d.w, d.out, d.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr()
d.resid = res.data_ptr() if res is not None else None
d.M, d.N, d.K, d.Nw, d.lda, d.ldw, d.ldo, d.ldr = M, N, K, N, K, K, out.stride(0), N
d.mode, d.batch, d.zdiv, d.alpha = 0, 1, 1, 1.0
d.flags = (((a.cfg + 1) << 24) if a.cfg < 4 else {4: 0x80000 | 0x200000, 5: 0x40000 | 0x200000, 6: 5 << 24, 7: (5 << 24) | 0x10000000, 8: 7 << 24}[a.cfg]) | (1 if a.geglu else 0)
# cfg 4: 256x128 three-stage tile, 5: 128x128 (two blocks per CU), 6: the ping-pong 256x256 tile (gemm_pp.hip), 7: its first schedule, 8: the ping-pong 192x256 tile
d.tune_group_m = a.gm
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    _lib.check(lib.icd_gemm(C.byref(d), st))
torch.cuda.synchronize()
tiles = {0: (256, 256), 1: (256, 320), 2: (192, 256), 3: (128, 320), 4: (256, 128), 5: (128, 128), 6: (256, 256), 7: (256, 256), 8: (192, 256)}[a.cfg]
nblk = ((M + tiles[0] - 1) // tiles[0]) * ((N + tiles[1] - 1) // tiles[1])
buf = torch.zeros((nblk, 8), dtype=torch.int64, device="cuda")
d.debug_timeline = buf.data_ptr()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.icd_gemm(C.byref(d), st))
e1.record()
torch.cuda.synchronize()
d.debug_timeline = None
t = buf.cpu().numpy().astype(np.float64) / 100.0          # 100 MHz -> us
t0 = t[:, 0].min()
start, pro, main, epi = t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
end = t[:, 3] - t0
print(f"{a.kind} {M}x{N}x{K} cfg{a.cfg} gm={a.gm} geglu={a.geglu}: {nblk} blocks, event time {e0.elapsed_time(e1) * 1e3:.1f} us, "
      f"first start -> last end {end.max():.1f} us, {2.0 * M * N * K / end.max() / 1e6:.0f} TFLOP/s over that span")
q = lambda v: f"min {v.min():7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}"
print("  prologue  us: " + q(pro))
print("  main loop us: " + q(main) + f"   ({np.median(main) / ((K + 63) // 64):.3f} us per k-tile)")
print("  epilogue  us: " + q(epi))
raw = buf.cpu().numpy()
if raw[:, 5].max() > 0:
    clk = (raw[:, 5] - raw[:, 4]).astype(np.float64) / np.maximum(main, 1e-3) / 1e3        # ticks per us -> GHz
    print(f"  shader clock during the main loop (s_memtime ticks / s_memrealtime): p50 {np.median(clk):.3f} GHz  min {clk.min():.3f}  max {clk.max():.3f};"
          f"  main loop = {np.median(raw[:, 5] - raw[:, 4]) / ((K + 63) // 64):.0f} shader cycles per k-tile")
order = np.argsort(start)
per_round = 512 if a.cfg == 5 else 256
rounds = (nblk + per_round - 1) // per_round
for r in range(min(rounds, 6)):
    sel = order[r * per_round:(r + 1) * per_round]
    print(f"  blocks {r * per_round:5d}..{min(nblk, (r + 1) * per_round) - 1:5d} (by start time): start {start[sel].min():7.2f} .. {start[sel].max():7.2f}   "
          f"end {end[sel].min():7.2f} .. {end[sel].max():7.2f}")
busy = (pro + main + epi).sum() / (per_round * end.max())
print(f"  CU occupancy by resident blocks over the span: {busy * 100:.1f} %   main-loop share of resident time: {main.sum() / (pro + main + epi).sum() * 100:.1f} %")
