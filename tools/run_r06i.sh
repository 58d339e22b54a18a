#!/bin/bash
python3 -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py tests/test_predict_tiling.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06i_tests.log
python3 -m pytest "tests/test_model_gpu.py::test_generator_narrow_features_on_large_maps" -x -q -m gpu 2>&1 | grep -v "^E  *[-+0-9e., \[\]]*$" | tail -30 > gpurun_out/r06i_narrow.log
bash tools/pmc_infer.sh r06i bf16 32 trace > gpurun_out/r06i_trace32.txt 2>&1
