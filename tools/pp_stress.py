#!/usr/bin/env python3
"""Race screen of the ping-pong GEMM tiles (gemm_pp.hip / gemm_pp320.hip): their results are bit-identical to the lockstep tiles' by
construction (same k order per accumulator, same epilogue), so ANY difference is a staging-pipeline bug (a fragment read before its DMA landed,
a slot restaged under a reader).  Such races come and go with timing: every shape is run REPS times on two streams at once (the second stream
perturbs the first one's timing with other tiles' launches) and compared bit for bit with one lockstep reference.

    python tools/pp_stress.py [REPS]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device="cuda", generator=g).half()
CASES = [("256x256", 5 << 24, 1 << 24, [(8192, 2560, 1280), (4096, 1280, 320), (16384, 1280, 640), (1000, 512, 192), (8192, 1280, 5120)]),
         ("192x256", 7 << 24, 3 << 24, [(8192, 1280, 1280), (8192, 1280, 5120), (3000, 512, 448), (16384, 2560, 320)]),
         ("256x320", 6 << 24, 2 << 24, [(32768, 640, 640), (8192, 2560, 1280), (16384, 320, 320), (5000, 1280, 2560), (32768, 640, 2560)])]
side = torch.cuda.Stream()
bad = 0
for name, flag, lockstep, shapes in CASES:
    for M, N, K in shapes:
        a, w, b, r = rnd(M, K), rnd(N, K) * K ** -0.5, torch.randn(N, device="cuda", generator=g), rnd(M, N)
        ref = ops.gemm(a, w, bias=b, resid=r, debug_flags=lockstep)
        a2, w2 = rnd(4096, 1280), rnd(1280, 1280) * 1280 ** -0.5
        torch.cuda.synchronize()
        n_bad = 0
        for i in range(reps):
            with torch.cuda.stream(side):                          # timing noise: another tile family on another stream
                for _ in range(2):
                    ops.gemm(a2, w2, debug_flags=(1 + i % 3) << 24)
            out = ops.gemm(a, w, bias=b, resid=r, debug_flags=flag)
            if not torch.equal(out, ref):
                n_bad += 1
        torch.cuda.synchronize()
        bad += n_bad
        print(f"{name} {M} x {N} x {K}: {reps - n_bad}/{reps} launches bit-identical to the lockstep tile")
# conv: the ping-pong conv tiles against the lockstep ones
import torch.nn.functional as F
for name, flag, lockstep, (B, H, W, Ci, Co) in [("256x256 conv", 5 << 24, 1 << 24, (4, 32, 32, 640, 1280)), ("256x320 conv", 6 << 24, 2 << 24, (8, 64, 64, 320, 320)),
                                                ("192x256 conv", 7 << 24, 3 << 24, (2, 32, 32, 1280, 1280))]:
    x, w = rnd(B * H * W, Ci), rnd(Co, 9 * Ci) * (9 * Ci) ** -0.5
    ref = ops.conv3x3(x, B, H, W, w, None, debug_flags=lockstep)
    n_bad = 0
    for i in range(reps):
        out = ops.conv3x3(x, B, H, W, w, None, debug_flags=flag)
        n_bad += int(not torch.equal(out, ref))
    bad += n_bad
    print(f"{name} B={B} {H}x{W} {Ci}->{Co}: {reps - n_bad}/{reps} launches bit-identical to the lockstep tile")
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} mismatching launches)")
sys.exit(1 if bad else 0)
