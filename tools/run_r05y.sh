set -u
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "single_step_convlstm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
AB_STEPS=20 bash tools/ab_step.sh "new:" "old:WDG_DHIGH_DIRECT=0" "new:" "old:WDG_DHIGH_DIRECT=0" "new:" "old:WDG_DHIGH_DIRECT=0" > gpurun_out/r05am_ab.txt 2>&1; cut -c1-60 gpurun_out/r05am_ab.txt
python tools/perf_ops.py --cases lstm_b_bwd_x > gpurun_out/r05am_perf.txt 2>&1; tail -3 gpurun_out/r05am_perf.txt
