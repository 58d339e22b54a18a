set -u
O=gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -4 > $O/r04h_tests.log
python tools/perf_ops.py --reps 9 --cases g0_fwd,g0_fwd_bn,g2_fwd,g2_fwd_bn,g4_fwd,g5_fwd,g5_fwd_bn,d1_fwd,d1_fwd_ln,d1_dgrad,d2_fwd_ln,g0_dgrad,g2_dgrad,g4_dgrad > $O/r04h_perf_epi.txt 2>&1
B="python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-serial-pass --no-generator-leg --no-config-legs --no-split-leg --no-child-legs"
J='import json,sys; j=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(j["ms_per_step"],2), round(j["host_enqueue_ms_per_step"],1))'
$B 2>/dev/null | python -c "$J" plain > $O/r04h_bench.txt
$B 2>/dev/null | python -c "$J" plain_again >> $O/r04h_bench.txt
cat $O/r04h_tests.log $O/r04h_perf_epi.txt $O/r04h_bench.txt
