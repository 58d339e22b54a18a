#!/bin/bash
# A/B: weight stages by LDS-DMA (variant library) against the register-staged default; correctness of the variant first
WDG_LIB=$PWD/gpurun_variants/libwdgan_wdma.so python3 -m pytest tests/test_bf16_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06x_tests_wdma.log
for rep in 1 2; do
for N in 16 32 64; do
  echo "default $N: $(python3 tools/prof_infer_group.py bf16 10 graph $N 2>/dev/null | tail -1)" >> gpurun_out/r06x_ab.txt
  echo "wdma    $N: $(WDG_LIB=$PWD/gpurun_variants/libwdgan_wdma.so python3 tools/prof_infer_group.py bf16 10 graph $N 2>/dev/null | tail -1)" >> gpurun_out/r06x_ab.txt
done
done
echo "== default" > gpurun_out/r06x_perf_patch.txt; python3 tools/perf_patch.py 5 >> gpurun_out/r06x_perf_patch.txt 2>&1
echo "== wdma" >> gpurun_out/r06x_perf_patch.txt; WDG_LIB=$PWD/gpurun_variants/libwdgan_wdma.so python3 tools/perf_patch.py 5 >> gpurun_out/r06x_perf_patch.txt 2>&1
