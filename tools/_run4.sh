set -u
O=gpurun_out
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1))'
: > $O/r04d_queues.txt
for Q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$Q $B 2>/dev/null | python -c "$J" plain_q$Q >> $O/r04d_queues.txt
  GPU_MAX_HW_QUEUES=$Q WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist_q$Q >> $O/r04d_queues.txt
done
GPU_MAX_HW_QUEUES=8 $B --size 96 --timesteps 24 --batch 8 2>/dev/null | python -c "$J" t24_q8 >> $O/r04d_queues.txt
$B --size 96 --timesteps 24 --batch 8 2>/dev/null | python -c "$J" t24_q4 >> $O/r04d_queues.txt
cat $O/r04d_queues.txt
