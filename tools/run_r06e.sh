#!/bin/bash
# lock-step hypothesis: the second workgroup of every CU starts late (patch_stagger, 0.1 us units) — new (direct weight fragments) and old (LDS-staged weights) K loops
echo "== new kernel" > gpurun_out/r06e_stagger.txt
python3 tools/perf_patch.py 5 patch_stagger=50,100,150,200,300 >> gpurun_out/r06e_stagger.txt 2>&1
echo "== old kernel (LDS-staged weights, table-driven loop)" >> gpurun_out/r06e_stagger.txt
WDG_LIB=$PWD/gpurun_variants/libwdgan_old.so python3 tools/perf_patch.py 5 patch_stagger=50,100,150,200,300 >> gpurun_out/r06e_stagger.txt 2>&1
