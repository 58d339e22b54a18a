#!/bin/bash
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06j_gpu_tests.log
bash tools/ab_dp_proxy.sh r06j > /dev/null 2>&1
