#!/bin/bash
# kernel + HIP runtime API trace of the train step (who keeps the GPU waiting): gpurun_out/<tag>_kernel_trace.csv, <tag>_hip_api.csv
#   bash tools/trace_host.sh <tag> [bench args...]
set -u
TAG=${1:-r04}; shift || true
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --no-serial-pass --no-child-legs --steps 2 --warmup 2 $*"
( cd /tmp && rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $ROOT/$OUT/th_trace -o run -- $BENCH > $ROOT/$OUT/${TAG}_th.log 2>&1 )
K=$(find $OUT/th_trace -name '*kernel_trace.csv' | head -1); A=$(find $OUT/th_trace -name '*hip_api_trace.csv' | head -1)
cp "$K" $OUT/${TAG}_kernel_trace.csv; cp "$A" $OUT/${TAG}_hip_api.csv; ls -la $OUT/${TAG}_hip_api.csv; rm -rf $OUT/th_trace
python3 tools/host_gaps.py $OUT/${TAG}_kernel_trace.csv $OUT/${TAG}_hip_api.csv
