set -u
mkdir -p gpurun_out/pp6
B="--steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-ref-batching --no-sdxl --no-edit --no-live-traffic"
for r in 1 2; do
for arch in sd15 sdxl; do
for t in 0 0x20000000; do
python bench.py --arch $arch $B --gemm-tune $t 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
kf = d.get('kernel_families', {})
print('$arch gemm_tune=$t round $r: value', d['value'], 'one_batch', d.get('value_one_batch_at_a_time'), {k: round(v['ms'], 2) for k, v in kf.items()})
"
done; done; done > gpurun_out/pp6/bench_ab.txt 2>&1
cat gpurun_out/pp6/bench_ab.txt
python tools/vs_library.py > gpurun_out/pp6/vs_library.txt 2>&1
cat gpurun_out/pp6/vs_library.txt
