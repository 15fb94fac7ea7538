import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from invertible_cd_amd import ops
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6
g = torch.Generator(device="cuda").manual_seed(0)
for B, H, W, C in ((32, 64, 64, 320), (8, 128, 128, 320), (8, 64, 64, 320), (1, 512, 512, 128)):
    x = torch.randn(B * H * W, C, device="cuda", generator=g).half()
    w = (torch.randn(4, 9 * C, device="cuda", generator=g) * (9 * C) ** -0.5).half()
    b = torch.zeros(4, device="cuda")
    print(f"conv_out B={B} {H}x{W} C={C}: {t(lambda: ops.conv_out(x, B, H, W, w, b)):.1f} us")
