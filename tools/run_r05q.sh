set -u
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "misc or colsum or train_step or test_generator" 2>&1 | tail -3
AB_STEPS=12 bash tools/ab_step.sh "new:" "new:" > gpurun_out/r05q_ab.txt 2>&1; cat gpurun_out/r05q_ab.txt
