#!/usr/bin/env python3
"""Plain and bias+residual GEMMs (us per launch) for ONE build of the package given by path - run it once per build in the same
job to A/B library versions on one box (profiles/r02_epilogue_ab.txt was made with builds of five commits under ab/<sha>/).

    python tools/epilogue_ab.py <dir containing invertible_cd_amd/> [lib]      # "lib": also time torch.matmul
"""
import sys, os
sys.path.insert(0, os.path.abspath(sys.argv[1]))
import torch
from invertible_cd_amd import ops
SHAPES = [(8192, 1280, 1280), (8192, 1280, 5120), (8192, 5120, 1280), (8192, 2560, 1280), (32768, 640, 640), (32768, 640, 2560),
          (32768, 2560, 640), (131072, 320, 320), (131072, 320, 1280), (131072, 1280, 320), (8192, 1280, 11520)]
def timeit(fn, iters=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
lib = len(sys.argv) > 2
row = []
for M, N, K in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    t0 = timeit(lambda: ops.gemm(a, w, out=out))
    t1 = timeit(lambda: ops.gemm(a, w, bias=b, resid=r, out=out))
    s = f"{M:7d}{N:6d}{K:6d} plain {t0:7.1f}  bias+res {t1:7.1f}"
    if lib:
        wt = w.t()
        s += f"  lib {timeit(lambda: torch.matmul(a, wt, out=out)):7.1f}"
    print(s, flush=True)
