#!/bin/bash
# Evidence for profiles/: the bench line, the rocprofv3 kernel-trace summary of the same command, the two PMC passes behind
# roofline.traffic, the -s output of the parity tests.  Run on the GPU box:  bash tools/profile_round.sh r02_x
# (writes gpurun_out/<tag>/...; copy the summaries you want judged into profiles/).  The rocprofv3 passes run with ONE batch in flight
# (--in-flight 1): per-kernel durations and counters are those of the roofline pass of the bench line, undisturbed by a second stream.
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
[ -n "${SKIP_DEFAULT:-}" ] || timeout -k 5 1500 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
for arch in ${ARCHS:-sd15 sdxl}; do
  b=32; [ $arch = sdxl ] && b=8
  python bench.py --arch $arch --steps 5 --warmup 2 --no-cpu-baseline --no-sdxl --no-edit > $OUT/${TAG}_bench_${arch}_b${b}.json 2> $OUT/bench_$arch.err
  timeout -k 5 900 rocprofv3 --kernel-trace -d $OUT/prof_$arch -o $arch -- python bench.py --arch $arch --steps 3 --warmup 2 --in-flight 1 --no-cpu-baseline --no-vae --no-ref-batching --no-sdxl --no-edit \
      > $OUT/${TAG}_bench_${arch}_b${b}_under_rocprof.json 2> $OUT/rocprof_$arch.err
  db=$(find $OUT/prof_$arch -name "*.db" | head -1)
  python tools/rocpd_stats.py "$db" > $OUT/${TAG}_bench_${arch}_b${b}_kernel_stats.txt 2>> $OUT/rocprof_$arch.err
  # (counters on this library's kernels only - torch's weight-building kernels run unserialised - and every pass under a timeout: an
  #  unrestricted SDXL pass once sat for an hour)
  KR='--kernel-include-regex gemm|attn|gn_|splitk|layernorm|conv_out|carry|split2|pack|sinusoid|x0_step|activation'
  timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv $KR -d $OUT/pmcF_$arch -o p -- python bench.py --arch $arch --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-vae --no-ref-batching --no-profile --no-sdxl --no-edit > /dev/null 2> $OUT/pmcF_$arch.err
  timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv $KR -d $OUT/pmcW_$arch -o p -- python bench.py --arch $arch --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-vae --no-ref-batching --no-profile --no-sdxl --no-edit > /dev/null 2> $OUT/pmcW_$arch.err
  python tools/hbm_traffic.py $OUT/pmcF_$arch $OUT/pmcW_$arch --arch $arch --batch $b > $OUT/${TAG}_hbm_traffic_${arch}_b${b}.json 2>> $OUT/pmcF_$arch.err
  timeout -k 5 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv $KR -d $OUT/pmcM_$arch -o p -- python bench.py --arch $arch --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-vae --no-ref-batching --no-profile --no-sdxl --no-edit > /dev/null 2> $OUT/pmcM_$arch.err
  python tools/mfma_util.py $OUT/pmcM_$arch --arch $arch --batch $b > $OUT/${TAG}_mfma_util_${arch}_b${b}.json 2>> $OUT/pmcM_$arch.err
  rm -rf $OUT/pmcF_$arch $OUT/pmcW_$arch $OUT/pmcM_$arch
  find $OUT/prof_$arch -name "*.db" -delete
done
tail -c 600 $OUT/${TAG}_bench_default.json
