set -u
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05zz_gpu_tests.log
cat gpurun_out/r05zz_gpu_tests.log
bash tools/profile_round.sh r05zz
cp gpurun_out/pmc_traffic.json gpurun_out/r05zz_pmc_traffic.json
cp gpurun_out/rocprof_dominant.json gpurun_out/r05zz_rocprof_dominant.json
