set -u
WDG_TUNING=igemm_dma=31 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05f_tests_dma.log
cat gpurun_out/r05f_tests_dma.log
CASES=d1_fwd_ln,d1_fwd,d1_dgrad,d2_fwd_ln,d2_dgrad,d3_fwd_ln,d3_dgrad,d4_fwd_ln,d4_dgrad,g0_fwd,g2_fwd,g2_dgrad,g4_fwd,g4_dgrad,g5_fwd,g5_dgrad,upconv_gemm
python3 tools/perf_ops.py --reps 9 --cases $CASES --variant "base:igemm_dma=0" --variant "dma:igemm_dma=31" > gpurun_out/r05f_perf_dma.txt 2>&1
cat gpurun_out/r05f_perf_dma.txt
