#!/bin/bash
# Round-6 evidence in one GPU call: tools/profile_round.sh r06 (bench default + per-architecture rocprofv3 stats + PMC traffic / MFMA passes), the parity run,
# the precision-probe gaps, the per-shape tables and the in-flight sweep.  Writes gpurun_out/r06/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06
mkdir -p $OUT
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1
python tools/precision_gap.py 2>/dev/null | grep -v amdgpu > $OUT/r06_precision_gap.txt
python tools/shape_profile.py --arch sd15 --batch 32 2>/dev/null | grep -v amdgpu > $OUT/r06_shapes_sd15_b32.txt
python tools/shape_profile.py --arch sdxl --batch 8 2>/dev/null | grep -v amdgpu > $OUT/r06_shapes_sdxl_b8.txt
B="--steps 8 --warmup 3 --no-cpu-baseline --no-vae --no-ref-batching --no-sdxl --no-edit --no-live-traffic --no-profile"
for arch in sd15 sdxl; do for n in 1 2 3; do
python bench.py --arch $arch $B --in-flight $n 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$arch in_flight=$n value', d['value'], 'ms_per_step', d['ms_per_step'])"
done; done > $OUT/r06_in_flight.txt 2>&1
python -m pytest tests -m gpu -q -s --durations=15 2>&1 | grep -v '^$' > $OUT/r06_parity.txt
tail -30 $OUT/r06_parity.txt
cat $OUT/r06_in_flight.txt $OUT/r06_precision_gap.txt
# the whole round's GEMM work against the lockstep tiles of rounds 1 - 5, two batches in flight, alternating, same box
B="--steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-ref-batching --no-sdxl --no-edit --no-live-traffic"
for r in 1 2; do for arch in sd15 sdxl; do for t in 0 0x20000000; do
python bench.py --arch $arch $B --gemm-tune $t 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); kf = d.get('kernel_families', {})
print('$arch gemm_tune=$t round $r: value', d['value'], 'one_batch', d.get('value_one_batch_at_a_time'), {k: round(v['ms'], 2) for k, v in kf.items()})"
done; done; done > $OUT/r06_pp_bench_ab.txt 2>&1
python tools/vs_library.py 2>/dev/null | grep -v amdgpu > $OUT/r06_vs_library.txt
cat $OUT/r06_pp_bench_ab.txt $OUT/r06_vs_library.txt
