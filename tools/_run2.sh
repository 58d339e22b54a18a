# round-4 second GPU call: A/B of the new kernel variants + the data-parallel schedule question
set -u
O=gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -4 > $O/r04b_tests.log
python tools/perf_ops.py --reps 9 --cases d1_fwd_ln,d1_fwd,g0_fwd,g2_fwd,d2_fwd_ln --variant "t2d_on:tile2d=1" --variant "t2d_pow2:tile2d=2" --variant "t2d_off:tile2d=0" > $O/r04b_perf_tile2d.txt 2>&1
python tools/perf_ops.py --reps 9 --cases halo16_fwd,halo16_dgrad,halo16_fwd_lncat,g11_fwd,g11_dgrad --variant "stage1:halo1_stage=1" --variant "stage0:halo1_stage=0" > $O/r04b_perf_halo1.txt 2>&1
python tools/perf_ops.py --reps 9 --cases d1_wgrad,d2_wgrad,d3_wgrad,d4_wgrad,g0_wgrad,g2_wgrad,g4_wgrad,g5_wgrad --variant "xcd0:wgrad_xcd=0" --variant "xcd16:wgrad_xcd=16" --variant "xcd64:wgrad_xcd=64" > $O/r04b_perf_wgrad_xcd.txt 2>&1
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1))'
$B 2>/dev/null | python -c "$J" plain > $O/r04b_dp.txt
WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist >> $O/r04b_dp.txt
WDG_DIST_ALWAYS=1 $B --no-sync-bn 2>/dev/null | python -c "$J" dist_nosyncbn >> $O/r04b_dp.txt
WDG_DIST_ALWAYS=1 WDG_OVERLAP_GEN=0 $B 2>/dev/null | python -c "$J" dist_overlapgen0 >> $O/r04b_dp.txt
WDG_DIST_ALWAYS=1 WDG_OVERLAP_DISC=0 $B 2>/dev/null | python -c "$J" dist_overlapdisc0 >> $O/r04b_dp.txt
WDG_OVERLAP_GEN=0 $B 2>/dev/null | python -c "$J" plain_overlapgen0 >> $O/r04b_dp.txt
cat $O/r04b_tests.log $O/r04b_perf_tile2d.txt $O/r04b_perf_halo1.txt $O/r04b_perf_wgrad_xcd.txt $O/r04b_dp.txt
