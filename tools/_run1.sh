python -m pytest tests/test_ops_gpu.py -x -q -k "philox or copy or permute or channels" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -2
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-serial-pass --steps 8 --warmup 3"
python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), round(d['ms_per_step'],2)); print({k:(v.get('ms') or v.get('seconds')) for k,v in d['other_configs'].items()})"
