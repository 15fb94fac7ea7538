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


def timeit(fn, iters=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def plan_of(M, N, K, ws):
    import ctypes as C
    from invertible_cd_amd import _lib
    d = _lib.GemmDesc()
    buf = torch.empty(8, device="cuda")
    d.a0 = d.w = d.out = buf.data_ptr()
    d.M, d.N, d.K, d.Nw, d.lda, d.ldw, d.ldo = M, N, K, N, K, K, N
    d.mode, d.batch, d.zdiv, d.alpha = 0, 1, 1, 1.0
    n = _lib.load().icd_gemm_workspace_bytes(M, N, K)
    if n > 0:
        d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), n
    info = _lib.GemmPlanInfo()
    _lib.check(_lib.load().icd_gemm_plan(C.byref(d), C.byref(info)), "icd_gemm_plan")
    return f"{info.tile_m}x{info.tile_n}" + (f" split-K {info.ksplit}" if info.ksplit > 1 else "")


# Round 6: the two sides are timed ROUND-ROBIN (ROUNDS x [icd_gemm x 20, torch.matmul x 20]) and the minimum per side is reported, as
# tools/gemm_bench.py does: rounds 1 - 5 timed icd_gemm first and the library second in ONE pass per shape, right after the idle gap in which the
# operands are generated - whatever runs first after an idle period runs at ramping clocks (10 - 20 % slow on this chip), a bias against the
# first side of every row.
ROUNDS = int(os.environ.get("ROUNDS", "5"))
print(f"{'M':>7s} {'N':>6s} {'K':>6s} | {'icd_gemm (min of ' + str(ROUNDS) + ')':>22s} | {'torch.matmul':>18s} | ratio | planner tile")
ws = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)
for M, N, K in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    wt = w.t()                                       # library computes a @ w^T with the weight stored [N, K] like ours
    f_icd, f_lib = (lambda: ops.gemm(a, w, out=out)), (lambda: torch.matmul(a, wt, out=out))
    for _ in range(10):
        f_icd(); f_lib()
    t_icd, t_lib = [], []
    for _ in range(ROUNDS):
        t_icd.append(timeit(f_icd)); t_lib.append(timeit(f_lib))
    t_icd, t_lib = min(t_icd), min(t_lib)
    fl = 2.0 * M * N * K
    print(f"{M:7d} {N:6d} {K:6d} | {t_icd * 1e6:12.1f} us {fl / t_icd / 1e12:6.0f} TF | {t_lib * 1e6:8.1f} us {fl / t_lib / 1e12:6.0f} TF | {t_lib / t_icd:5.2f}x | {plan_of(M, N, K, ws)}")
