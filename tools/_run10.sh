set -u
O=gpurun_out
C=d3_fwd,d3_dgrad,d3_wgrad,d4_fwd,d4_dgrad,d4_wgrad
python tools/perf_ops.py --reps 7 --replan --cases $C --variant "base:tile64=2,small_m=256,force_fwd_split=0,force_dgrad_split=0,force_wgrad_split=0" --variant "t128x64:tile64=0,small_m=256" --variant "t128x128:tile64=0,small_m=0" > $O/r04j_small_tiles.txt 2>&1
python tools/perf_ops.py --reps 7 --replan --cases $C --variant "base:tile64=2,small_m=256,force_fwd_split=0,force_dgrad_split=0,force_wgrad_split=0" --variant "s8:force_fwd_split=8,force_dgrad_split=8,force_wgrad_split=8" --variant "s16:force_fwd_split=16,force_dgrad_split=16,force_wgrad_split=16" --variant "s32:force_fwd_split=32,force_dgrad_split=32,force_wgrad_split=32" --variant "s96:force_fwd_split=96,force_dgrad_split=96,force_wgrad_split=96" > $O/r04j_small_splits.txt 2>&1
python tools/perf_ops.py --reps 7 --replan --cases $C --variant "base:tile64=2,small_m=256,force_fwd_split=0,force_dgrad_split=0,force_wgrad_split=0" --variant "t128x128_s32:tile64=0,small_m=0,force_fwd_split=32,force_dgrad_split=32" --variant "t128x128_s64:tile64=0,small_m=0,force_fwd_split=64,force_dgrad_split=64" --variant "t128x64_s48:tile64=0,small_m=256,force_fwd_split=48,force_dgrad_split=48" > $O/r04j_small_mix.txt 2>&1
cat $O/r04j_small_tiles.txt $O/r04j_small_splits.txt $O/r04j_small_mix.txt
