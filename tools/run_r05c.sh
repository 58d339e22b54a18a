set -u
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05c_tests.log
cat gpurun_out/r05c_tests.log
bash tools/pmc_ops.sh r05c d1_fwd_ln,d1_wgrad,g0_fwd,g0_wgrad,d2_fwd_ln "class:tap_class_order=1" "rowmajor:tap_class_order=0"
for b in 1 2 4 8 3 7 15; do
  echo "== skeleton WDG_KLOOP_EXP=$b"
  WDG_LIB=$PWD/gpurun_variants/libwdgan_exp$b.so python3 tools/perf_ops.py --reps 5 --cases d1_fwd_ln,d1_dgrad,d1_wgrad,d2_fwd_ln,d2_dgrad,d2_wgrad,g0_fwd 2>&1 | tail -8
done > gpurun_out/r05c_skeletons.txt 2>&1
cat gpurun_out/r05c_skeletons.txt
