set -u
CASES=g0_fwd,g2_fwd,g4_fwd,g4_dgrad,g5_fwd,d1_fwd_ln,d2_fwd_ln,d2_dgrad
for rep in 1 2; do
echo "== shipped"; python3 tools/perf_ops.py --reps 7 --cases $CASES 2>&1 | tail -9
echo "== setprio"; WDG_LIB=$PWD/gpurun_variants/libwdgan_expprio.so python3 tools/perf_ops.py --reps 7 --cases $CASES 2>&1 | tail -9
done > gpurun_out/r05l_setprio.txt 2>&1
cat gpurun_out/r05l_setprio.txt
