set -u
mkdir -p gpurun_out/pp1
python -m pytest tests/test_ops_gpu.py -q -x -k "big_tile or ping_pong or conv" 2>&1 | tail -4 > gpurun_out/pp1/test.txt
cat gpurun_out/pp1/test.txt
python tools/tune_gemm.py --arch sdxl --batch 8 > gpurun_out/pp1/tune_sdxl.txt 2>&1
python tools/tune_gemm.py --arch sd15 --batch 32 > gpurun_out/pp1/tune_sd15.txt 2>&1
tail -3 gpurun_out/pp1/tune_sd15.txt
