set -u
O=gpurun_out
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1), j.get("stream_placement"))'
: > $O/r04g_probe.txt
T="--size 96 --timesteps 24 --batch 8"
$B 2>/dev/null | python -c "$J" plain >> $O/r04g_probe.txt
WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist >> $O/r04g_probe.txt
WDG_DIST_ALWAYS=1 $B --no-sync-bn 2>/dev/null | python -c "$J" dist_nosyncbn >> $O/r04g_probe.txt
$B $T 2>/dev/null | python -c "$J" t24_plain >> $O/r04g_probe.txt
WDG_DIST_ALWAYS=1 $B $T 2>/dev/null | python -c "$J" t24_dist >> $O/r04g_probe.txt
GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | python -c "$J" plain_q8 >> $O/r04g_probe.txt
GPU_MAX_HW_QUEUES=8 WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist_q8 >> $O/r04g_probe.txt
GPU_MAX_HW_QUEUES=8 WDG_DIST_ALWAYS=1 $B $T 2>/dev/null | python -c "$J" t24_dist_q8 >> $O/r04g_probe.txt
GPU_MAX_HW_QUEUES=8 WDG_DIST_ALWAYS=1 WDG_CHAIN_GRAPHS=0 $B $T 2>/dev/null | python -c "$J" t24_dist_q8_nographs >> $O/r04g_probe.txt
cat $O/r04g_probe.txt
