F="--no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --steps 8 --warmup 3"
for m in 1 2 1 2; do WDG_OVERLAP_BRANCHES=$m python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline overlap_branches=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
