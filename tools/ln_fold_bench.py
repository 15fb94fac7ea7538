#!/usr/bin/env python3
"""LayerNorm folded into the consuming GEMM vs LayerNorm kernel + plain GEMM, same process, same clocks.

    python tools/ln_fold_bench.py [M C]      (default: SDXL 32x32 level at B = 8: M = 8192, C = 1280)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import ops

M, C = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 1280)
B = 8
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).half()
x = rnd(M, C)
gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")


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
    return e0.elapsed_time(e1) / iters * 1e3


cases = [("to_qk  N=2C", 2 * C, dict()), ("to_q   N=C", C, dict()), ("ff1    N=8C geglu", 8 * C, dict(geglu=True))]
print(f"M={M} C={C}: us per call, best of 3 rounds")
for name, N, kw in cases:
    w = rnd(N, C) * C ** -0.5
    bias = torch.zeros(N, device="cuda")
    s = w.float().sum(1).contiguous()
    res = []
    for _ in range(3):
        t_ln = timeit(lambda: ops.layernorm(x, gamma, beta))
        t_st = timeit(lambda: ops.layernorm_stats(x))
        ln = ops.layernorm(x, gamma, beta)
        st = ops.layernorm_stats(x)
        t_plain = timeit(lambda: ops.gemm(ln, w, bias=bias, **kw))
        t_fold = timeit(lambda: ops.gemm(x, w, bias=bias, ln_stats=st, ln_colsum=s, **kw))
        res.append((t_ln, t_st, t_plain, t_fold))
    t_ln, t_st, t_plain, t_fold = [min(r[i] for r in res) for i in range(4)]
    print(f"  {name:20s} layernorm {t_ln:6.1f}  stats {t_st:6.1f}  gemm {t_plain:7.1f}  gemm+ln-epilogue {t_fold:7.1f}   "
          f"unfused {t_ln + t_plain:7.1f} -> fused {t_st + t_fold:7.1f}")
w = rnd(C, C) * C ** -0.5
s = w.float().sum(1).contiguous()
st = ops.layernorm_stats(x)
ln = ops.layernorm(x, gamma, beta)
n_tok = M // B
t_plain = min(timeit(lambda: ops.project_vt(ln, w, B, n_tok, n_tok)) for _ in range(3))
t_fold = min(timeit(lambda: ops.project_vt(x, w, B, n_tok, n_tok, ln_stats=st, ln_colsum=s)) for _ in range(3))
print(f"  to_v   N=C transposed                              gemm {t_plain:7.1f}  gemm+ln-epilogue {t_fold:7.1f}")
