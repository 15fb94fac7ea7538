#!/usr/bin/env python3
"""The north-star kernel against what it replaces: LN2 -> to_q -> cross-attention as ONE launch (256 x 256 host tile, the 192 x 256 one
laid out per sample - round 5 - and the 128-wide host) vs projection + attention, SDXL shapes.

    python tools/xattn_bench.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import ops, _lib
def timeit(fn, iters=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
lib = _lib.load()
for (B, n_tok, C, nk) in [(8, 1024, 1280, 77), (16, 1024, 1280, 77), (4, 1024, 1280, 77), (8, 4096, 640, 77)]:
    H = C // 64; M = B * n_tok
    h = torch.randn(M, C, device="cuda").half()
    w = (torch.randn(C, C, device="cuda") * C ** -0.5).half(); b = torch.randn(C, device="cuda"); s = w.float().sum(1).contiguous()
    k = torch.randn(B * nk, C, device="cuda").half(); ld = (nk + 7) // 8 * 8
    vt = torch.zeros(B, C, ld, device="cuda", dtype=torch.float16); vt[:, :, :nk] = torch.randn(B, C, nk, device="cuda").half()
    st = torch.empty(M, 2, device="cuda")
    q = torch.empty(M, C, device="cuda", dtype=torch.float16)
    def two():
        ops.gemm(h, w, bias=b, ln_stats=st, ln_colsum=s, out=q, ln_compute=True)
        ops.attention_fused(q, k, vt, B, H, n_tok, nk, 64, 0.125)
    t_two = timeit(two)
    t_gemm = timeit(lambda: ops.gemm(h, w, bias=b, ln_stats=st, ln_colsum=s, out=q, ln_compute=True))
    t_fused = timeit(lambda: ops.query_cross_attention(h, w, k, vt, B, n_tok, nk, 0.125, bias=b, ln_stats=st, ln_colsum=s, ln_compute=True,
                                                       xattn_tile=5))
    t_f192 = timeit(lambda: ops.query_cross_attention(h, w, k, vt, B, n_tok, nk, 0.125, bias=b, ln_stats=st, ln_colsum=s, ln_compute=True,
                                                      xattn_tile=6))
    t_f128 = timeit(lambda: ops.query_cross_attention(h, w, k, vt, B, n_tok, nk, 0.125, bias=b, ln_stats=st, ln_colsum=s, ln_compute=True,
                                                      xattn_tile=2))
    fl = 2.0 * M * C * C + 4.0 * M * nk * C
    print(f"B={B} n={n_tok} C={C}: projection {t_gemm:6.1f} + attention = {t_two:6.1f} us | fused (256x256 host) {t_fused:6.1f} us = {fl / t_fused / 1e6:5.0f} TFLOP/s"
          f" = {fl / t_fused / 1e6 / 2516.6 * 100:4.1f} % of MFMA peak | fused (192x256 host) {t_f192:6.1f} us = {fl / t_f192 / 1e6:5.0f} TFLOP/s"
          f" = {fl / t_f192 / 1e6 / 2516.6 * 100:4.1f} % | fused (128x128 host) {t_f128:6.1f} us", flush=True)
