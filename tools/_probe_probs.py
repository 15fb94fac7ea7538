import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invertible_cd_amd import ops
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, H, Nq, Nk, d) in [(8, 8, 1024, 1024, 80), (8, 8, 256, 256, 160), (8, 8, 4096, 77, 40), (8, 8, 1024, 77, 80)]:
    q = torch.randn(B * Nq, H * d, device="cuda").half(); k = torch.randn(B * Nk, H * d, device="cuda").half()
    qc = torch.randint(0, 255, (B * Nq, H * d), device="cuda", dtype=torch.uint8); kc = torch.randint(0, 255, (B * Nk, H * d), device="cuda", dtype=torch.uint8)
    ld = (Nk + 7) // 8 * 8
    acc = torch.zeros(B * H, Nq, ld, device="cuda", dtype=torch.float16)[:, :, :Nk]
    t0 = timeit(lambda: ops.attention_probs(q, k, B, H, Nq, Nk, d, d ** -0.5))
    t1 = timeit(lambda: ops.attention_probs(q, k, B, H, Nq, Nk, d, d ** -0.5, q_carry=qc, k_carry=kc))
    t2 = timeit(lambda: ops.attention_probs(q, k, B, H, Nq, Nk, d, d ** -0.5, q_carry=qc, k_carry=kc, acc=acc))
    mb = B * H * Nq * ld * 2 / 1e6
    print(f"B={B} H={H} {Nq}x{Nk} d={d}: P = {mb:.0f} MB; fp16 operands {t0:.1f} us, split {t1:.1f} us, split + store accumulate {t2:.1f} us")
