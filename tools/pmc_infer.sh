#!/bin/bash
# Counter evidence for the 16-bit inference path: one kernel trace + six --pmc passes (each its own run, no trace
# domains beside --pmc) over tools/prof_infer_group.py (one predict() group of 16 tiles x 24 h, eager launches).
#   bash tools/pmc_infer.sh <tag> [bf16|fp16] [tiles] [trace]     (trace: the kernel trace only)
# writes gpurun_out/<tag>_infer_group_<prec>_kernel_stats.csv and gpurun_out/<tag>_pmc_infer_<prec>.txt
set -u
TAG=${1:-r06}; PREC=${2:-bf16}; TILES=${3:-16}; MODE=${4:-all}
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/pmci_trace -o run -- python3 $ROOT/tools/prof_infer_group.py $PREC 5 eager $TILES > $ROOT/$OUT/pmci_trace.log 2>&1
S=$(find $ROOT/$OUT/pmci_trace -name '*kernel_stats.csv' | head -1); cp "$S" $ROOT/$OUT/${TAG}_infer_group_${PREC}_kernel_stats.csv; rm -rf $ROOT/$OUT/pmci_trace
tail -1 $ROOT/$OUT/pmci_trace.log
python3 - "$ROOT/$OUT/${TAG}_infer_group_${PREC}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
# forwards in the trace = launches of a kernel that runs once per forward (warm-ups + timed repetitions of tools/prof_infer_group.py)
nf = max([int(r['Calls']) for r in rows if 'wdg_upconv_fused_h16_kernel' in r['Name'] or 'wdg_conv_thin16_h16_kernel' in r['Name']] or [7])
print("kernel time per forward (%d forwards in the trace) %.3f ms" % (nf, tot / nf / 1e6))
for r in rows[:12]:
    print("%8.3f ms/fwd %5d x %8.1f us  %s" % (float(r['TotalDurationNs']) / nf / 1e6, int(r['Calls']) // nf, float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
[ "$MODE" = trace ] && exit 0
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/pmci_$i -o run -- python3 $ROOT/tools/prof_infer_group.py $PREC 2 eager $TILES > $ROOT/$OUT/pmci_$i.log 2>&1
  echo "pass $i rc=$? $(grep -ci error $ROOT/$OUT/pmci_$i.log) error lines"
done
cd $ROOT
python3 - "$TAG" "$PREC" <<'PY'
import csv, glob, collections, re, sys
tag, prec = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmci_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'^void ', '', re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')))[:110]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
out = f'gpurun_out/{tag}_pmc_infer_{prec}.txt'
with open(out, 'w') as o:
    o.write("# per-launch means; FETCH_SIZE/WRITE_SIZE in KB (gfx950: FETCH_SIZE counts wide reads at half their bytes)\n")
    for k in sorted(acc):
        if 'wdg' not in k:
            continue
        a = {c: acc[k][c] / n[k][c] for c in acc[k]}
        o.write(f"{k}   launches/pass={max(n[k].values())}\n")
        for c in sorted(a):
            o.write(f"   {c:36s} {a[c]:16.1f}\n")
        d = []
        if a.get('SQ_BUSY_CYCLES'):
            d.append(f"mfma_busy/sq_busy={a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / a['SQ_BUSY_CYCLES']:.3f}")
        if a.get('SQ_WAVE_CYCLES'):
            d.append(f"wait_inst_any/wave_cycles={a.get('SQ_WAIT_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}")
            d.append(f"active_inst_any/wave_cycles={a.get('SQ_ACTIVE_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}")
        if a.get('SQ_LDS_IDX_ACTIVE'):
            d.append(f"lds_conflict/lds_active={a.get('SQ_LDS_BANK_CONFLICT', 0) / a['SQ_LDS_IDX_ACTIVE']:.3f}")
        if a.get('TCC_REQ_sum'):
            d.append(f"l2_hit={a.get('TCC_HIT_sum', 0) / max(1.0, a.get('TCC_HIT_sum', 0) + a.get('TCC_MISS_sum', 0)):.3f}")
        o.write("   -> " + "  ".join(d) + "\n")
print(open(out).read()[:9000])
PY
rm -rf gpurun_out/pmci_[0-9]
