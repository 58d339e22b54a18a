set -u
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_engine_cpu.py -x -q -m gpu -k "batchnorm or test_generator or train_step or lazy" 2>&1 | tail -3
python3 tools/perf_ops.py --reps 7 --cases bn_bwd_c16_reduce,bn_bwd_c16_apply,bn_bwd_c128_reduce,bn_bwd_c128_apply,misc_meansq,misc_colsum512,misc_colsum2,ln_bwd_c16 2>&1 | tail -10
AB_STEPS=12 bash tools/ab_step.sh "new:" "new:" > gpurun_out/r05t_ab.txt 2>&1; cat gpurun_out/r05t_ab.txt
