python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['frac'], d['step_frac_of_mfma_f32_peak'])"
