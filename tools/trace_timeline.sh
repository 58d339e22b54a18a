#!/bin/bash
# kernel timeline of the default (multi-stream) headline step: rocprofv3 --kernel-trace -> gpurun_out/<tag>_kernel_trace.csv
# (start / end / queue per dispatch), summarised by tools/timeline_summary.py.   bash tools/trace_timeline.sh <tag> [bench args...]
set -u
TAG=${1:-r04}; shift || true
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --no-serial-pass --no-child-legs --steps 2 --warmup 2 $*"
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/tl_trace -o run -- $BENCH > $ROOT/$OUT/${TAG}_tl.log 2>&1 )
T=$(find $OUT/tl_trace -name '*kernel_trace.csv' | head -1)
cp "$T" $OUT/${TAG}_kernel_trace.csv; rm -rf $OUT/tl_trace
python3 tools/timeline_summary.py $OUT/${TAG}_kernel_trace.csv
