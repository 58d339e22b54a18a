set -u
./gpurun_variants/lds_dma_oob_probe > gpurun_out/r05e_lds_dma_oob.txt 2>&1; head -8 gpurun_out/r05e_lds_dma_oob.txt; tail -18 gpurun_out/r05e_lds_dma_oob.txt | head -6
CASES=d1_fwd_ln,d1_dgrad,d2_fwd_ln,d2_dgrad,d3_fwd_ln,d3_dgrad,d4_fwd_ln,d4_dgrad,g5_fwd,g2_dgrad
for rep in 1 2; do
echo "== shipped"; python3 tools/perf_ops.py --reps 7 --cases $CASES 2>&1 | tail -11
echo "== early loads"; WDG_LIB=$PWD/gpurun_variants/libwdgan_expearly.so python3 tools/perf_ops.py --reps 7 --cases $CASES 2>&1 | tail -11
done > gpurun_out/r05e_early_loads.txt 2>&1
cat gpurun_out/r05e_early_loads.txt
