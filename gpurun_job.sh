python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ref-batching 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SD15', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])"
python bench.py --arch sdxl --steps 2 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SDXL', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])"
