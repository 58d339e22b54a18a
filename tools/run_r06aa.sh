#!/bin/bash
python3 -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py tests/test_predict_tiling.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06aa_tests.log
bash tools/pmc_infer.sh r06aa bf16 64 trace > gpurun_out/r06aa_trace64.txt 2>&1
python3 tools/perf_patch.py 5 > gpurun_out/r06aa_perf_patch.txt 2>&1
