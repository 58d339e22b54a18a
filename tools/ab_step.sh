#!/bin/bash
# A/B of the train step on ONE GPU box: every argument is a "label:ENV=val ENV=val ... [-- bench args]" variant, run one after
# the other in fresh processes (5 timed steps after 2 warm-up steps, no extra legs); prints "label ms_per_step".
#   bash tools/ab_step.sh "plain:" "dist:WDG_DIST_ALWAYS=1" "t24:-- --size 96 --timesteps 24 --batch 8"
B="python bench.py --gpus 1 --steps ${AB_STEPS:-5} --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), j.get("stream_placement"))'
for V in "$@"; do
  L="${V%%:*}"; R="${V#*:}"
  E="${R%%--*}"; A=""
  case "$R" in *--*) A="${R#*--}";; esac
  env $E $B $A 2>/dev/null | python -c "$J" "$L"
done
