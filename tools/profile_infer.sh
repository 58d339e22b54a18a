#!/bin/bash
# kernel stats of one 16-tile predict() group of the shipped generator at inference precision (eager launches):
#   bash tools/profile_infer.sh <tag> [bf16|fp16]
set -u
TAG=${1:-r04}; PREC=${2:-bf16}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/inf_trace -o run -- python3 $ROOT/tools/prof_infer.py $PREC 5 > $ROOT/$OUT/inf_trace.log 2>&1 )
S=$(find $OUT/inf_trace -name '*kernel_stats.csv' | head -1); cp "$S" $OUT/${TAG}_infer_${PREC}_kernel_stats.csv; rm -rf $OUT/inf_trace
tail -2 $OUT/inf_trace.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_infer_${PREC}_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total per forward (7 forwards) %.3f ms" % (tot/7e6))
for r in rows[:24]:
    print("%8.3f ms/fwd %5d x %8.1f us  %s" % (float(r['TotalDurationNs'])/7e6, int(r['Calls'])//7, float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
