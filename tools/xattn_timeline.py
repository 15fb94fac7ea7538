#!/usr/bin/env python3
"""Per-block timeline of the fused query projection + cross-attention launch (s_memrealtime stamps, icd_gemm_desc.debug_timeline): dispatch
offset, prologue, main loop, softmax / P.V epilogue - for the 256 x 256 and the per-sample 192 x 256 host tiles.

    python tools/xattn_timeline.py [B]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from invertible_cd_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_tok, C, nk = 1024, 1280, 77
M = B * n_tok
h = torch.randn(M, C, device="cuda").half()
w = (torch.randn(C, C, device="cuda") * C ** -0.5).half(); b = torch.randn(C, device="cuda"); s = w.float().sum(1).contiguous()
k = torch.randn(B * nk, C, device="cuda").half(); ld = (nk + 7) // 8 * 8
vt = torch.zeros(B, C, ld, device="cuda", dtype=torch.float16); vt[:, :, :nk] = torch.randn(B, C, nk, device="cuda").half()
st = torch.empty(M, 2, device="cuda")
for tile, rows in ((5, 256), (6, 192)):
    nblk = B * ((n_tok + rows - 1) // rows) * (C // 256)
    run = lambda tl=None: ops.query_cross_attention(h, w, k, vt, B, n_tok, nk, 0.125, bias=b, ln_stats=st, ln_colsum=s, ln_compute=True,
                                                    xattn_tile=tile, timeline=tl)
    for _ in range(20):
        run()
    buf = torch.zeros((nblk, 8), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    run(buf)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    start, main, epi, end = t[:, 0] - t0, t[:, 2] - t[:, 0], t[:, 3] - t[:, 2], t[:, 3] - t0
    q = lambda v: f"min {v.min():6.2f}  p50 {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}"
    full = main > 0
    print(f"B={B} {rows} x 256 host: {nblk} blocks, first start -> last end {end.max():.1f} us")
    print("  start offset        us: " + q(start))
    print("  prologue + main loop us: " + q(main[full]))
    print("  softmax / P.V epilogue us: " + q(epi[full]))
