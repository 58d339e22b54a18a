#!/bin/bash
python3 -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py tests/test_predict_tiling.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06u_tests.log
python3 tools/perf_patch.py 5 patch_pair=0 > gpurun_out/r06u_perf_patch_pair.txt 2>&1
bash tools/pmc_infer.sh r06u bf16 32 trace > gpurun_out/r06u_trace32.txt 2>&1
