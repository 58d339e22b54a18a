#!/bin/bash
python3 -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py tests/test_predict_tiling.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06d_tests.log
bash tools/pmc_infer.sh r06d bf16 16 trace > gpurun_out/r06d_trace16.txt 2>&1
export WDG_LIB=$PWD/gpurun_variants/libwdgan_v0.so
python3 tools/perf_patch.py 5 patch_dbg=1,2,4,8,3,12,15 > gpurun_out/r06d_patch_skeletons.txt 2>&1
