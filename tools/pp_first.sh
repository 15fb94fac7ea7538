set -u
mkdir -p gpurun_out/pp3
python -m pytest tests/test_ops_gpu.py -q -x 2>&1 | tail -6 > gpurun_out/pp3/test.txt
cat gpurun_out/pp3/test.txt
python tools/tune_gemm.py --arch sd15 --batch 32 --min-us 30 > gpurun_out/pp3/tune_sd15.txt 2>&1
python tools/tune_gemm.py --arch sdxl --batch 8 --min-us 30 > gpurun_out/pp3/tune_sdxl.txt 2>&1
tail -2 gpurun_out/pp3/tune_sd15.txt gpurun_out/pp3/tune_sdxl.txt
python -m pytest tests/test_unet_gpu.py tests/test_loading_gpu.py -q -x -s 2>&1 | grep -v '^$' | tail -40 > gpurun_out/pp3/test2.txt
tail -30 gpurun_out/pp3/test2.txt
