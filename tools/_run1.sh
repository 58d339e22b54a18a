python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -2
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 8 --warmup 3"
python bench.py $F > gpurun_out/r03d.json 2>/dev/null
python bench.py $F --tune convlstm1_mfma=2 > gpurun_out/r03d_fwdvalu.json 2>/dev/null
python - <<'PY'
import json
for k in ('r03d','r03d_fwdvalu'):
    d=json.loads(open(f'gpurun_out/{k}.json').read().strip().splitlines()[-1])
    print(k, d['value'], d['ms_per_step'])
PY
