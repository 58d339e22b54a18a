#!/bin/bash
# Runs on the GPU box (gpurun):  bash tools/pmc_ops.sh <tag> <cases> "<variant>" ["<variant>" ...]
# One tools/perf_ops.py process per variant ("name:key=val,...") under the four --pmc passes of tools/profile_round.sh;
# per variant a per-kernel summary gpurun_out/<tag>_<name>_pmc.csv (fabric fetch / write, L2 hit, waits, LDS conflicts, matrix-pipe
# busy) + the un-profiled A/B timing table of all variants in one process gpurun_out/<tag>_ab.txt.
set -u
TAG=$1; CASES=$2; shift 2
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS=()
for v in "$@"; do ARGS+=(--variant "$v"); done
python3 tools/perf_ops.py --reps 9 --cases "$CASES" --replan "${ARGS[@]}" > $OUT/${TAG}_ab.txt 2>&1
cat $OUT/${TAG}_ab.txt
for v in "$@"; do
  name=${v%%:*}
  P="python3 $ROOT/tools/perf_ops.py --reps 2 --cases $CASES --replan --variant $v"
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/po_fetch -o run -- $P > $ROOT/$OUT/${TAG}_po1.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $ROOT/$OUT/po_write -o run -- $P > $ROOT/$OUT/${TAG}_po2.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $ROOT/$OUT/po_sq -o run -- $P > $ROOT/$OUT/${TAG}_po3.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/po_mfma -o run -- $P > $ROOT/$OUT/${TAG}_po4.log 2>&1 )
  python3 tools/pmc_summary.py $OUT/po_fetch $OUT/po_write $OUT/po_sq $OUT/po_mfma $OUT/${TAG}_${name}_pmc.csv --no-traffic > /dev/null 2>&1
  echo "== $v"; grep -v "at::native\|philox\|elementwise" $OUT/${TAG}_${name}_pmc.csv | head -12
  rm -rf $OUT/po_fetch $OUT/po_write $OUT/po_sq $OUT/po_mfma
done
