set -u
O=gpurun_out
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1))'
: > $O/r04e_queues.txt
for Q in 1 2 3 4; do
  GPU_MAX_HW_QUEUES=$Q $B 2>/dev/null | python -c "$J" plain_q$Q >> $O/r04e_queues.txt
done
for Q in 2 4 8; do
  GPU_MAX_HW_QUEUES=$Q WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 $B 2>/dev/null | python -c "$J" noinner_q$Q >> $O/r04e_queues.txt
  GPU_MAX_HW_QUEUES=$Q WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 WDG_OVERLAP_DISC=0 $B 2>/dev/null | python -c "$J" noinner_nodisc_q$Q >> $O/r04e_queues.txt
  GPU_MAX_HW_QUEUES=$Q WDG_DIST_ALWAYS=1 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 $B 2>/dev/null | python -c "$J" dist_noinner_q$Q >> $O/r04e_queues.txt
done
GPU_MAX_HW_QUEUES=2 WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist_q2 >> $O/r04e_queues.txt
GPU_MAX_HW_QUEUES=3 WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist_q3 >> $O/r04e_queues.txt
WDG_OVERLAP_GEN=0 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 $B 2>/dev/null | python -c "$J" serial_q4 >> $O/r04e_queues.txt
cat $O/r04e_queues.txt
