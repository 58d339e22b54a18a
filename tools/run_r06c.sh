#!/bin/bash
# skeleton table of the patch kernel (experiment build), the new groups-per-launch test, and the 32-tile trace
export WDG_LIB=$PWD/gpurun_variants/libwdgan_v0.so
python3 tools/perf_patch.py 5 patch_dbg=1,2,4,8,16,32,3,12,15 > gpurun_out/r06c_patch_skeletons.txt 2>&1
unset WDG_LIB
python3 -m pytest tests/test_predict_tiling.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06c_tests.log
bash tools/pmc_infer.sh r06c bf16 32 trace > gpurun_out/r06c_trace32.txt 2>&1
python3 tools/prof_infer_group.py bf16 10 graph 16 > gpurun_out/r06c_graph.txt 2>&1
python3 tools/prof_infer_group.py bf16 10 graph 32 >> gpurun_out/r06c_graph.txt 2>&1
