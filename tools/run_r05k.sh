set -u
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05k_gpu_tests.log
cat gpurun_out/r05k_gpu_tests.log
bash tools/profile_round.sh r05k
