python -m pytest tests/test_ops_gpu.py -x -q -k "conv" 2>&1 | tail -2
F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 5 --warmup 2 --per-layer"
python bench.py $F --tune igemm_ks=1 --tune igemm_pipe=3 > gpurun_out/r03c_p3.json 2> gpurun_out/r03c_p3_layers.txt
python bench.py $F --tune igemm_ks=1 --tune igemm_pipe=5 > gpurun_out/r03c_p5.json 2> gpurun_out/r03c_p5_layers.txt
python - <<'PY'
import json
for k in (3,5):
    d=json.loads(open(f'gpurun_out/r03c_p{k}.json').read().strip().splitlines()[-1])
    print(k, d['value'], d['ms_per_step'])
PY
