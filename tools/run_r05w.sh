set -u
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "batchnorm or test_generator or train_step" 2>&1 | tail -2
AB_STEPS=12 bash tools/ab_step.sh "new:" "bn1024:WDG_TUNE=bn_bwd_blocks=1024" "hoist:WDG_HOIST_REAL=1" "new:" "bn1024:WDG_TUNE=bn_bwd_blocks=1024" "hoist:WDG_HOIST_REAL=1" > gpurun_out/r05w_ab.txt 2>&1; cat gpurun_out/r05w_ab.txt
