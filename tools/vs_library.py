#!/usr/bin/env python3
"""icd_gemm against the vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS) on the dense shapes of the two UNets,
same box, same process.  Plain GEMM only (no bias / residual / GEGLU epilogue on either side), fp16 in, fp16 out.

    python tools/vs_library.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import ops

SHAPES = [(8192, 1280, 1280), (8192, 1280, 5120), (8192, 5120, 1280), (8192, 2560, 1280), (32768, 640, 640), (32768, 640, 2560),
          (32768, 2560, 640), (131072, 320, 320), (131072, 320, 1280), (131072, 1280, 320), (2048, 1280, 1280), (8192, 1280, 11520),
          (32768, 640, 5760), (131072, 320, 2880)]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


print(f"{'M':>7s} {'N':>6s} {'K':>6s} | {'icd_gemm':>18s} | {'torch.matmul':>18s} | ratio")
for M, N, K in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    wt = w.t()                                       # library computes a @ w^T with the weight stored [N, K] like ours
    t_icd = timeit(lambda: ops.gemm(a, w, out=out))
    t_lib = timeit(lambda: torch.matmul(a, wt, out=out))
    fl = 2.0 * M * N * K
    print(f"{M:7d} {N:6d} {K:6d} | {t_icd * 1e6:8.1f} us {fl / t_icd / 1e12:6.0f} TF | {t_lib * 1e6:8.1f} us {fl / t_lib / 1e12:6.0f} TF | {t_lib / t_icd:5.2f}x")
