set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
B="python3 $ROOT/bench.py --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --no-serial-pass --no-child-legs --size 96 --timesteps 24 --batch 8"
$B --steps 5 --warmup 2 | tail -1 | cut -c1-400
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/t24_trace -o run -- $B --steps 2 --warmup 1 > $ROOT/$OUT/t24_trace.log 2>&1 )
S=$(find $OUT/t24_trace -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG:-r05}_t24_kernel_stats.csv; rm -rf $OUT/t24_trace
export WDG_OVERLAP_GEN=0 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/t24_trace -o run -- $B --steps 2 --warmup 1 > $ROOT/$OUT/t24_trace.log 2>&1 )
S=$(find $OUT/t24_trace -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG:-r05}_t24_kernel_stats_serial.csv; rm -rf $OUT/t24_trace
head -40 $OUT/${TAG:-r05}_t24_kernel_stats_serial.csv | cut -c1-150
