#!/bin/bash
# A/B of the data-parallel stand-ins (engine/trainer.py DistSync, WDG_DP_PROXY=1) on one GPU: which part of the collectives' cost is the
# fifth busy queue (large gradient all-reduces on their own stream) and which the blocking small ones (SyncBN statistics, metrics).
#   bash tools/ab_dp_proxy.sh <tag>      -> gpurun_out/<tag>_ab_dp_proxy.txt
TAG=${1:-r06}; OUT=gpurun_out/${TAG}_ab_dp_proxy.txt; : > $OUT
B="python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-generator-leg --no-split-leg --no-config-legs --no-serial-pass --no-child-legs"
run() { echo "== $1" >> $OUT; shift; env "$@" MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) $B 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   ms_per_step %.3f' % j['ms_per_step'], (j.get('rccl') or {}).get('proxy'))" >> $OUT; }
for rep in 1 2; do
run "undistributed (no process group)" X=1
run "one-rank RCCL group, real (identity) collectives" WDG_DIST_ALWAYS=1
run "proxy: large on a fifth stream (16 workgroups, ring time at 150 GB/s) + small 10 us waits" WDG_DIST_ALWAYS=1 WDG_DP_PROXY=1
run "proxy: large only" WDG_DIST_ALWAYS=1 WDG_DP_PROXY=1 WDG_DP_PROXY_SMALL_US=0
run "proxy: small only (10 us)" WDG_DIST_ALWAYS=1 WDG_DP_PROXY=1 WDG_DP_PROXY_LARGE=0
run "proxy: small only (30 us: a slower 8-rank latency)" WDG_DIST_ALWAYS=1 WDG_DP_PROXY=1 WDG_DP_PROXY_LARGE=0 WDG_DP_PROXY_SMALL_US=30
run "proxy: large with 64 workgroups" WDG_DIST_ALWAYS=1 WDG_DP_PROXY=1 WDG_DP_PROXY_SMALL_US=0 WDG_DP_PROXY_BLOCKS=64
run "proxy: large at 50 GB/s per link (3x longer)" WDG_DIST_ALWAYS=1 WDG_DP_PROXY=1 WDG_DP_PROXY_SMALL_US=0 WDG_DP_PROXY_GBPS=50
done
cat $OUT
