set -u
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "fused_single_step_convlstm or test_discriminator or train_step" 2>&1 | tail -4 > gpurun_out/r05j_tests.log
cat gpurun_out/r05j_tests.log
AB_STEPS=12 bash tools/ab_step.sh "new:" "mixcopy:WDG_MIX_IN_PLACE=0" "new:" "mixcopy:WDG_MIX_IN_PLACE=0" "new:" "mixcopy:WDG_MIX_IN_PLACE=0" > gpurun_out/r05j_ab_step.txt 2>&1
cat gpurun_out/r05j_ab_step.txt
