set -u
C=g4_wgrad,g2_wgrad,g0_wgrad,d1_wgrad,d2_wgrad,g5_wgrad
python tools/perf_ops.py --reps 7 --cases $C 2>&1 | tail -7 > gpurun_out/r05ar_base.txt
WDG_LIB=gpurun_variants/libwdgan_exp8.so python tools/perf_ops.py --reps 7 --cases $C 2>&1 | tail -7 > gpurun_out/r05ar_nobarrier.txt
paste gpurun_out/r05ar_base.txt gpurun_out/r05ar_nobarrier.txt | cut -c1-200
