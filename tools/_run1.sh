python -m pytest tests/test_ops_gpu.py -x -q -k "layernorm or conv_cases or conv_fwd" 2>&1 | tail -2
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 8 --warmup 3"
for i in 1 2; do python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), round(d['ms_per_step'],2), 'serial', round(d['roofline']['single_stream_ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['roofline']['all_conv_kernels'].items() if 'halo' in k})"; done
