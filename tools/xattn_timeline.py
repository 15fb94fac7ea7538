#!/usr/bin/env python3
"""Per-block timeline (prologue / main loop / epilogue, s_memrealtime stamps) of the fused query-projection + cross-attention
launch on the 256 x 256 host tile, SDXL 1024-token layer at B = 8.

    python tools/xattn_timeline.py
"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from invertible_cd_amd import ops, _lib
lib = _lib.load()
B, n_tok, C_, nk = 8, 1024, 1280, 77
M = B * n_tok
h = torch.randn(M, C_, device="cuda").half()
w = (torch.randn(C_, C_, device="cuda") * C_ ** -0.5).half(); b = torch.randn(C_, device="cuda"); s = w.float().sum(1).contiguous()
k = torch.randn(B * nk, C_, device="cuda").half(); ld = 80
vt = torch.zeros(B, C_, ld, device="cuda", dtype=torch.float16); vt[:, :, :nk] = torch.randn(B, C_, nk, device="cuda").half()
st = torch.empty(M, 2, device="cuda")
f = lambda tl=None: ops.query_cross_attention(h, w, k, vt, B, n_tok, nk, 0.125, bias=b, ln_stats=st, ln_colsum=s, ln_compute=True, timeline=tl)
for _ in range(10): f()
torch.cuda.synchronize()
nblk = (M // 256) * (C_ // 256)
buf = torch.zeros((nblk, 8), dtype=torch.int64, device="cuda")
f(buf); torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64) / 100.0
t0 = t[:, 0].min()
q = lambda v: f"min {v.min():7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}"
print("start    ", q(t[:, 0] - t0)); print("prologue ", q(t[:, 1] - t[:, 0])); print("main loop", q(t[:, 2] - t[:, 1])); print("epilogue ", q(t[:, 3] - t[:, 2])); print("end      ", q(t[:, 3] - t0))
print("clock GHz", q((t[:, 5] - t[:, 4]) / 100.0 / np.maximum(t[:, 2] - t[:, 1], 1e-3) / 1e3 * 100))
