set -u
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05zzz_gpu_tests.log
cat gpurun_out/r05zzz_gpu_tests.log
bash tools/profile_round.sh r05zzz
cp gpurun_out/pmc_traffic.json gpurun_out/r05zzz_pmc_traffic.json
cp gpurun_out/rocprof_dominant.json gpurun_out/r05zzz_rocprof_dominant.json
