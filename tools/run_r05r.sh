set -u
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "layernorm or test_discriminator or train_step_default" 2>&1 | tail -3
AB_STEPS=12 bash tools/ab_step.sh "new:" "wpr0:WDG_TUNE=reduce_wpr=0" "new:" "wpr0:WDG_TUNE=reduce_wpr=0" > gpurun_out/r05r_ab.txt 2>&1; cat gpurun_out/r05r_ab.txt
