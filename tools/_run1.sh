python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
F="--no-cpu-baseline --size 96 --timesteps 24 --batch 8 --steps 6 --warmup 2"
for m in 0 1 0 1; do WDG_WGRAD_STREAM=$m python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T24 wgrad_stream=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 8 --warmup 3"
for m in 0 1 0 1; do WDG_WGRAD_STREAM=$m python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline wgrad_stream=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
