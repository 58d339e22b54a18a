set -u
O=gpurun_out
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1))'
: > $O/r04f_sched.txt
T="--size 96 --timesteps 24 --batch 8"
$B $T 2>/dev/null | python -c "$J" t24_default_q4 >> $O/r04f_sched.txt
WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 $B $T 2>/dev/null | python -c "$J" t24_noinner_q4 >> $O/r04f_sched.txt
GPU_MAX_HW_QUEUES=8 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 $B $T 2>/dev/null | python -c "$J" t24_noinner_q8 >> $O/r04f_sched.txt
GPU_MAX_HW_QUEUES=8 WDG_WGRAD_STREAM=1 WDG_OVERLAP_BRANCHES=0 $B $T 2>/dev/null | python -c "$J" t24_wgradonly_q8 >> $O/r04f_sched.txt
GPU_MAX_HW_QUEUES=8 WDG_DIST_ALWAYS=1 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0 $B $T 2>/dev/null | python -c "$J" t24_dist_noinner_q8 >> $O/r04f_sched.txt
E="GPU_MAX_HW_QUEUES=8 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0"
env $E WDG_GEN_PRIO=-1 $B 2>/dev/null | python -c "$J" gen_hi >> $O/r04f_sched.txt
env $E WDG_DISC_PRIO=-1 $B 2>/dev/null | python -c "$J" disc_hi >> $O/r04f_sched.txt
env $E WDG_GEN_PRIO=-1 WDG_DISC_PRIO=-1 $B 2>/dev/null | python -c "$J" both_hi >> $O/r04f_sched.txt
env $E $B 2>/dev/null | python -c "$J" base_noinner_q8 >> $O/r04f_sched.txt
env $E WDG_WGRAD_STREAM=1 $B 2>/dev/null | python -c "$J" wgradonly_q8 >> $O/r04f_sched.txt
env $E WDG_OVERLAP_BRANCHES=1 $B 2>/dev/null | python -c "$J" branchesonly_q8 >> $O/r04f_sched.txt
cat $O/r04f_sched.txt
