set -u
python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "train_step or conv_fwd_dgrad_wgrad or upsample_conv_transpose_backward" 2>&1 | tail -4 > gpurun_out/r05i_tests.log
cat gpurun_out/r05i_tests.log
bash tools/ab_step.sh "new:" "noasm:WDG_ASSEMBLE_INPUT=0" "red1:WDG_TUNING=wgrad_reduce4=0" "new:" "noasm:WDG_ASSEMBLE_INPUT=0" "red1:WDG_TUNING=wgrad_reduce4=0" > gpurun_out/r05i_ab_step.txt 2>&1
cat gpurun_out/r05i_ab_step.txt
