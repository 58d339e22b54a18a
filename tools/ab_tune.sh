#!/bin/bash
# whole-step A/B of wdg_set_tuning keys: bash tools/ab_tune.sh <tag> "key=v" "key=v" ...   (alternating with the default, 2 repetitions)
TAG=$1; shift; OUT=gpurun_out/${TAG}_ab_tune.txt; : > $OUT
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-generator-leg --no-config-legs --no-serial-pass --no-child-legs"
for rep in 1 2; do
  echo "default: $($B 2>/dev/null | python3 -c "import sys,json; [print('%.3f ms' % json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")" >> $OUT
  for kv in "$@"; do
    echo "$kv: $($B --tune $kv 2>/dev/null | python3 -c "import sys,json; [print('%.3f ms' % json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")" >> $OUT
  done
done
cat $OUT
