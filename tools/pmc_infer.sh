set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
P="python3 $ROOT/tools/prof_infer.py bf16 2"
cd /tmp
rocprofv3 --list-avail > $ROOT/$OUT/pmc_list_avail.txt 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/pmci_$i -o run -- $P > $ROOT/$OUT/pmci_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $ROOT
python3 - <<'PY'
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmci_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r'\(.*','',r['Kernel_Name'])[:60]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
with open('gpurun_out/pmci_summary.txt','w') as o:
    for k in acc:
        if 'wdg' not in k: continue
        o.write(k+'\n')
        for c in sorted(acc[k]): o.write(f"   {c:40s} {acc[k][c]/n[k][c]:16.1f}  (n={n[k][c]})\n")
print(open('gpurun_out/pmci_summary.txt').read()[:6000])
PY
rm -rf gpurun_out/pmci_[0-9]
