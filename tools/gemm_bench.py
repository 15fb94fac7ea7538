#!/usr/bin/env python3
"""GEMM / conv / attention microbenchmark through the C ABI (for A/B tuning and rocprofv3 --pmc runs).

    python tools/gemm_bench.py conv 32 64 64 320 320        # B H W Cin Cout (3x3, stride 1)
    python tools/gemm_bench.py dense 131072 320 320          # M N K
    python tools/gemm_bench.py geglu 131072 2560 320
    python tools/gemm_bench.py attn 32 8 4096 4096 40        # B H Nq Nk d
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import ops

kind = sys.argv[1]
a = [int(x) for x in sys.argv[2:]]
iters = int(os.environ.get("ITERS", "20"))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).half()
if kind == "conv":
    B, H, W, Ci, Co = a
    x, w, b = rnd(B * H * W, Ci), rnd(Co, 9 * Ci) * (9 * Ci) ** -0.5, torch.zeros(Co, device=dev)
    import ctypes as C
    from invertible_cd_amd import _lib
    def fn():
        out = torch.empty((B * H * W, Co), device=dev, dtype=torch.float16)
        d = _lib.GemmDesc()
        d.a0, d.w, d.out, d.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr()
        d.M, d.N, d.K, d.Nw, d.ldw, d.ldo = B * H * W, Co, 9 * Ci, Co, 9 * Ci, Co
        d.rows_per_sample, d.mode, d.C0, d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride = H * W, 1, Ci, H, W, H, W, 3, 1
        d.batch, d.zdiv, d.alpha, d.flags = 1, 1, 1.0, dbg
        _lib.check(_lib.load().icd_gemm(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out
    flops = 2.0 * B * H * W * Co * 9 * Ci
elif kind in ("dense", "geglu"):
    M, N, K = a
    x, w, b = rnd(M, K), rnd(N, K) * K ** -0.5, torch.zeros(N, device=dev)
    res = None if kind == "geglu" or os.environ.get("PLAIN") else rnd(M, N)      # PLAIN=1: bias only, no residual operand
    import ctypes as C
    from invertible_cd_amd import _lib
    skws = torch.empty(8 * M * N, device=dev, dtype=torch.float32) if M * N <= 8 << 20 else None
    def fn():
        out = torch.empty((M, N // 2 if kind == "geglu" else N), device=dev, dtype=torch.float16)
        d = _lib.GemmDesc()
        d.a0, d.w, d.out, d.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), b.data_ptr()
        d.resid = res.data_ptr() if res is not None else None
        d.M, d.N, d.K, d.Nw, d.lda, d.ldw, d.ldo, d.ldr = M, N, K, N, K, K, out.stride(0), N
        if os.environ.get("LDA0"):                # knock-out: every row of A is row 0 (A served by the L2 - "what if this operand never
            d.lda = 0                             # came from HBM"); results are meaningless, the timing is the point
        d.mode, d.batch, d.zdiv, d.alpha, d.flags = 0, 1, 1, 1.0, dbg | (1 if kind == "geglu" else 0)
        if skws is not None and kind == "dense":
            d.splitk_ws, d.splitk_ws_bytes = skws.data_ptr(), skws.numel() * 4
        _lib.check(_lib.load().icd_gemm(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out
    flops = 2.0 * M * N * K
elif kind == "ln":
    rows, Cc = a
    x, gam, bet = rnd(rows, Cc), torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    fn = lambda: ops.layernorm(x, gam, bet)
    flops = 4.0 * rows * Cc * 1e3          # reported "TFLOP/s" column = GB/s (4 B per element)
elif kind == "gn":
    B, HW, Cc = a
    x, gam, bet = rnd(B * HW, Cc), torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    fn = lambda: ops.groupnorm(x, B, HW, gam, bet, 1e-5, True)
    flops = 6.0 * B * HW * Cc * 1e3        # GB/s at 6 B per element
else:
    B, H, Nq, Nk, d = a
    q, k, v = rnd(B * Nq, H * d), rnd(B * Nk, H * d), rnd(B * Nk, H * d)
    ldv = (Nk + 7) // 8 * 8
    vt = torch.zeros(B, H * d, ldv, device=dev, dtype=torch.float16)
    vt[:, :, :Nk] = v.reshape(B, Nk, H * d).transpose(1, 2)
    import ctypes as C
    from invertible_cd_amd import _lib
    out_a = torch.empty_like(q)
    def fn():
        # DBGFLAGS here = icd_attention_fused_ex flags (2: prescaled query, 4: VALU scale form, bits 8..11: instantiation variant)
        _lib.check(_lib.load().icd_attention_fused_ex(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out_a.data_ptr(), B, H, Nq, Nk, d, q.stride(0),
                                                      k.stride(0), vt.stride(1), out_a.stride(0), vt.stride(0), d ** -0.5, dbg,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out_a
    flops = 4.0 * B * H * Nq * Nk * d
# DBGFLAGS may be a comma-separated list: every entry is timed in this process (same box).  The entries are visited round-robin
# ROUNDS times (default 5) and the minimum per entry is reported: the first thing timed after an idle period runs at ramping clocks
# (10-15 % slower, seen in round 3), which a single pass in list order turns into a bias against the first entry.
flag_list = os.environ.get("DBGFLAGS", "0").split(",")
rounds = int(os.environ.get("ROUNDS", "5"))
best = {}
for _ in range(20):
    dbg = int(flag_list[0], 0)
    fn()
torch.cuda.synchronize()
for r in range(rounds):
    for dbg_s in flag_list:
        dbg = int(dbg_s, 0)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        best.setdefault(dbg_s, []).append(ms)
for dbg_s in flag_list:
    v = sorted(best[dbg_s])
    ms = v[0]
    print(f"{kind} {a} flags={dbg_s}: {ms * 1e3:.1f} us/call (min of {rounds}; median {v[len(v) // 2] * 1e3:.1f})  {flops / ms / 1e9:.1f} TFLOP/s")
