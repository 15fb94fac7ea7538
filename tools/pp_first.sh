set -u
mkdir -p gpurun_out/pp8
python - <<'PY' > gpurun_out/pp8/persist_check.txt 2>&1
import torch, sys
sys.path.insert(0, '.')
from invertible_cd_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in [(8192, 2560, 1280), (4096, 1280, 320), (1000, 512, 192), (131072, 1280, 320)]:
    a = torch.randn(M, K, device="cuda", generator=g).half(); w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g)
    x = ops.gemm(a, w, bias=b, debug_flags=5 << 24)
    for _ in range(3):
        y = ops.gemm(a, w, bias=b, debug_flags=(5 << 24) | 0x40000000)
        assert torch.equal(x, y), (M, N, K)
    print("persistent == per-tile launch", M, N, K)
PY
cat gpurun_out/pp8/persist_check.txt
for s in "dense 8192 1280 1280" "dense 8192 1280 5120" "dense 8192 5120 1280" "geglu 8192 10240 1280" "geglu 131072 2560 320" "dense 131072 1280 320" "dense 32768 2560 640" "dense 8192 8192 8192"; do
  PLAIN=1 DBGFLAGS=0x5000000,0x45000000 ROUNDS=5 python tools/gemm_bench.py $s 2>&1 | tail -2
done > gpurun_out/pp8/bench.txt 2>&1
cat gpurun_out/pp8/bench.txt
