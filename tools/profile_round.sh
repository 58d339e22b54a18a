#!/bin/bash
# Runs on the GPU box (gpurun):  bash tools/profile_round.sh <tag> [quick]
#   1. default bench line                      -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats        -> gpurun_out/<tag>_kernel_stats.csv  (bench.py --steps 1 --warmup 1; default multi-stream schedule)
#                                                 gpurun_out/<tag>_kernel_stats_serial.csv  (the same on one stream: per-kernel durations)
#   3. four --pmc passes (fabric fetch, fabric write + L2 hit, SQ wait / LDS, MFMA busy) + tools/pmc_summary  -> gpurun_out/<tag>_pmc_summary.csv + pmc_traffic.json   (skipped with "quick")
# The summaries are then copied into profiles/ by hand (gpurun_out/ is scratch).
set -u
TAG=${1:-r02}
QUICK=${2:-}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --no-serial-pass --no-child-legs"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_trace -o run -- $BENCH --steps 1 --warmup 1 > $ROOT/$OUT/${TAG}_trace.log 2>&1 )
STATS=$(find $OUT/${TAG}_trace -name '*kernel_stats.csv' | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/${TAG}_kernel_stats.csv && head -25 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_trace
# the same on ONE stream (every overlap off): per-kernel durations that measure the kernel, not the sharing of the chip between
# streams — what bench.py's roofline figures are taken from and must agree with.  The PMC passes below run in this mode too.
export WDG_OVERLAP_GEN=0 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_trace -o run -- $BENCH --steps 1 --warmup 1 > $ROOT/$OUT/${TAG}_trace_serial.log 2>&1 )
STATS=$(find $OUT/${TAG}_trace -name '*kernel_stats.csv' | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/${TAG}_kernel_stats_serial.csv
TRACE=$(find $OUT/${TAG}_trace -name '*kernel_trace.csv' | head -1)
# the dominant kernel's average launch duration under the tracer (one-stream schedule) -> rocprof_dominant.json (bench.py quotes it
# beside its HIP-event figure when the kernel sources and the launch mix match).  Three figures of the SAME process, so that a
# reader can see where a difference between "the tracer" and "the HIP events" comes from:
#   avg_launch_us            all launches in the trace = warm-up step + timed step (what --stats averages)
#   avg_launch_us_timed_step the launches of the timed step only (the second half of the dispatches, by start time)
#   hip_events_us            bench.py's own HIP-event average over the timed step, read from the JSON line of this traced run
python3 - "$OUT/${TAG}_kernel_stats_serial.csv" "$OUT/rocprof_dominant.json" "$TRACE" "$OUT/${TAG}_trace_serial.log" <<'PY'
import csv, hashlib, json, sys
from pathlib import Path
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith("void wdg_igemm_kernel<128, 128")]
calls, ns = sum(int(r["Calls"]) for r in rows), sum(float(r["TotalDurationNs"]) for r in rows)
h = hashlib.sha256()
for f in sorted((Path("wind-downscaling-gan_amd") / "csrc").glob("*.h*")):
    h.update(f.name.encode()); h.update(f.read_bytes())
steps = 2      # bench.py --steps 1 --warmup 1: two train steps in the trace
out = {"kernel": "wdg_igemm_kernel<128,128>", "csrc_sha256": h.hexdigest(), "launches_per_step": calls / steps,
       "avg_launch_us": ns / calls * 1e-3, "source": sys.argv[1] + " (rocprofv3 --kernel-trace --stats, one-stream schedule)"}
try:
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[3]))
               if r["Kernel_Name"].startswith("void wdg_igemm_kernel<128, 128"))
    last = d[len(d) // 2:]
    out["avg_launch_us_warmup_step"] = sum(e - s for s, e in d[:len(d) // 2]) / (len(d) // 2) * 1e-3
    out["avg_launch_us_timed_step"] = sum(e - s for s, e in last) / len(last) * 1e-3
    for ln in open(sys.argv[4]):
        if ln.startswith("{"):
            out["hip_events_us"] = json.loads(ln)["roofline"]["avg_launch_ms"] * 1e3
except Exception as exc:
    out["per_step_error"] = repr(exc)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $OUT/${TAG}_trace
if [ -z "$QUICK" ]; then
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -o run -- $BENCH --steps 1 --warmup 0 > $ROOT/$OUT/${TAG}_pmc1.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $ROOT/$OUT/pmc_write -o run -- $BENCH --steps 1 --warmup 0 > $ROOT/$OUT/${TAG}_pmc2.log 2>&1 )
  ( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $ROOT/$OUT/pmc_sq -o run -- $BENCH --steps 1 --warmup 0 > $ROOT/$OUT/${TAG}_pmc3.log 2>&1 )
  # direct matrix-pipe utilisation: busy cycles of the MFMA pipe (summed over the 1024 SIMDs) against the kernel's active cycles
  ( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/pmc_mfma -o run -- $BENCH --steps 1 --warmup 0 > $ROOT/$OUT/${TAG}_pmc4.log 2>&1 )
  python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_mfma $OUT/${TAG}_pmc_summary.csv > $OUT/${TAG}_pmc_summary.log 2>&1
  head -30 $OUT/${TAG}_pmc_summary.csv
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_mfma
fi
