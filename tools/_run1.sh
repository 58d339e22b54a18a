bash tools/profile_round.sh r03i > gpurun_out/r03i_profile.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs > gpurun_out/r03i_bench_check.json 2>/dev/null
