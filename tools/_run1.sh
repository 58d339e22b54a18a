python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 8 --warmup 3"
for m in 1 1 1; do python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), round(d['ms_per_step'],2))"; done
