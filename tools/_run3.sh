set -u
O=gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -4 > $O/r04c_tests.log
python tools/perf_ops.py --reps 9 --cases d1_fwd_ln,d1_fwd,d2_fwd_ln,d2_fwd,d3_fwd_ln,d3_fwd,d4_fwd_ln,d4_fwd,upconv_fwd,upconv_gemm,upconv_bwd --variant "lnwave1:ln_wave=1" --variant "lnwave0:ln_wave=0" > $O/r04c_perf_lnwave.txt 2>&1
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1))'
$B 2>/dev/null | python -c "$J" plain > $O/r04c_dp.txt
WDG_DIST_INIT_ONLY=1 $B 2>/dev/null | python -c "$J" init_only >> $O/r04c_dp.txt
WDG_DIST_ALWAYS=1 WDG_DIST_NOOP=1 $B 2>/dev/null | python -c "$J" dist_noop >> $O/r04c_dp.txt
WDG_DIST_ALWAYS=1 WDG_DIST_NOOP=1 $B --no-sync-bn 2>/dev/null | python -c "$J" dist_noop_nosyncbn >> $O/r04c_dp.txt
WDG_DIST_ALWAYS=1 $B 2>/dev/null | python -c "$J" dist >> $O/r04c_dp.txt
cat $O/r04c_tests.log $O/r04c_perf_lnwave.txt $O/r04c_dp.txt
