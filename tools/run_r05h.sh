set -u
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm_backward or dense_head" 2>&1 | tail -8 > gpurun_out/r05h_tests_new.log
cat gpurun_out/r05h_tests_new.log
python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05h_tests_model.log
cat gpurun_out/r05h_tests_model.log
bash tools/ab_step.sh "new:" "lnb1:WDG_TUNING=dgrad_lnbwd=1" "nochain:WDG_CHAIN_LN_BWD=0" "new:" "lnb1:WDG_TUNING=dgrad_lnbwd=1" "dma0:WDG_TUNING=igemm_dma=0" > gpurun_out/r05h_ab_step.txt 2>&1
cat gpurun_out/r05h_ab_step.txt
