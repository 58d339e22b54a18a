#!/bin/bash
# Counter passes over one python command (each pass its own run, --pmc only).   bash tools/pmc_cmd.sh <tag> <kernel-substring> <script> [args...]
# writes gpurun_out/<tag>_pmc.txt: per-launch means of the kernels whose name contains the substring
set -u
TAG=$1; SUB=$2; SCRIPT=$(pwd)/$3; shift 3
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/pmcc_$i -o run -- python3 $SCRIPT "$@" > $ROOT/$OUT/pmcc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $ROOT
python3 - "$TAG" "$SUB" <<'PY'
import csv, glob, collections, re, sys
tag, sub = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmcc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'^void ', '', re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')))[:110]
        if sub in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
with open(f'gpurun_out/{tag}_pmc.txt', 'w') as o:
    for k in sorted(acc):
        a = {c: acc[k][c] / n[k][c] for c in acc[k]}
        o.write(f"{k}   launches/pass={max(n[k].values())}\n")
        for c in sorted(a):
            o.write(f"   {c:36s} {a[c]:16.1f}\n")
        d = []
        if a.get('SQ_BUSY_CYCLES'):
            d.append(f"mfma_busy/sq_busy={a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / a['SQ_BUSY_CYCLES']:.3f}")
        if a.get('SQ_WAVE_CYCLES'):
            d.append(f"wait_any/wave_cycles={a.get('SQ_WAIT_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}")
            d.append(f"wait_inst_any/wave_cycles={a.get('SQ_WAIT_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}")
            d.append(f"active_inst_any/wave_cycles={a.get('SQ_ACTIVE_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.3f}")
        o.write("   -> " + "  ".join(d) + "\n")
print(open(f'gpurun_out/{tag}_pmc.txt').read()[:6000])
PY
rm -rf gpurun_out/pmcc_[0-9]
