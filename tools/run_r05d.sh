set -u
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05d_tests.log
cat gpurun_out/r05d_tests.log
bash tools/ab_step.sh "chain1:WDG_CHAIN_LN_BWD=1" "chain0:WDG_CHAIN_LN_BWD=0" "chain1:WDG_CHAIN_LN_BWD=1" "chain0:WDG_CHAIN_LN_BWD=0" > gpurun_out/r05d_ab_step.txt 2>&1
cat gpurun_out/r05d_ab_step.txt
bash tools/pmc_ops.sh r05d upconv_bwd,upconv_fwd "x32:wgrad_xcd=32" "x0:wgrad_xcd=0"
