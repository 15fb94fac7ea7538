#!/usr/bin/env python3
"""LayerNorm statistics launch + GEMM vs the GEMM alone (statistics given) vs ICD_GEMM_LN_COMPUTE (the GEMM computes them itself),
on the LayerNorm-consuming shapes of both UNets.  profiles/r02_ln_inline_bench.txt.

    python tools/ln_inline_bench.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import ops
def timeit(fn, iters=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, C, N, geglu) in [(8192, 1280, 2560, False), (8192, 1280, 1280, False), (8192, 1280, 10240, True), (32768, 640, 1280, False), (32768, 640, 640, False),
                         (32768, 640, 5120, True), (131072, 320, 640, False), (131072, 320, 320, False), (131072, 320, 2560, True)]:
    x = torch.randn(M, C, device="cuda").half()
    w = (torch.randn(N, C, device="cuda") * C ** -0.5).half()
    b = torch.randn(N, device="cuda"); s = w.float().sum(1).contiguous()
    st = torch.empty(M, 2, device="cuda")
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.float16)
    def sep():
        st2 = ops.layernorm_stats(x)
        ops.gemm(x, w, bias=b, geglu=geglu, ln_stats=st2, ln_colsum=s, out=out)
    t_sep = timeit(sep)
    t_gemm = timeit(lambda: ops.gemm(x, w, bias=b, geglu=geglu, ln_stats=st, ln_colsum=s, out=out))
    t_inl = timeit(lambda: ops.gemm(x, w, bias=b, geglu=geglu, ln_stats=st, ln_colsum=s, out=out, ln_compute=True))
    print(f"{M:7d} {C:5d} {N:6d} geglu={int(geglu)}  stats+gemm {t_sep:7.1f}  gemm alone {t_gemm:7.1f}  inline {t_inl:7.1f} us", flush=True)
