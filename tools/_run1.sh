F="--no-cpu-baseline --no-serial-pass --size 96 --timesteps 24 --batch 8 --steps 6 --warmup 2"
for m in 0 1 0 1; do WDG_OVERLAP_GEN=0 WDG_LSTM_STEP_GEMM=$m python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T24 no-gen-overlap lstm_step_gemm=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
