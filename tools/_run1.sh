python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm" 2>&1 | tail -2
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 8 --warmup 3"
for m in 0 1 0 1; do python bench.py $F --tune halo_ln=$m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline halo_ln=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
