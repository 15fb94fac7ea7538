set -u
mkdir -p gpurun_out/pp7
python -m pytest tests/test_vae_gpu.py tests/test_clip_gpu.py tests/test_pipeline_gpu.py tests/test_entry_gpu.py -q -x -s 2>&1 | grep -v '^$' | tail -40 > gpurun_out/pp7/vae.txt
cat gpurun_out/pp7/vae.txt
