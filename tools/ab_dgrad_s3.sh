#!/bin/bash
# Evidence for dgrad_patch_s3.hip: both routes of wdg_conv_dgrad_lnbwd on the discriminator's 7x7 s3 32 -> 64 layer (32 images of 256 x 256),
# the patch kernel's skeletons and occupancy sweep (measurement builds in gpurun_variants/, made by the commands in the header of the
# output), its in-kernel phase clocks.   bash tools/ab_dgrad_s3.sh <tag>
TAG=${1:-r06}; OUT=gpurun_out/${TAG}_dgrad_s3.txt
{
echo "# every python process starts with 60 untimed launches (clocks: a cold first case measures 60 us more)
# routes (HIP events, 20 launches each; patch_s3 = dgrad_patch_s3.hip, igemm = wdg_igemm_kernel<256,32> epilogue 5 / 6)"
python3 tools/bench_dgrad_s3.py 32 20 2>/dev/null | grep route
echo "# occupancy: LDS padded by k KB (1 + 256 * (k + 1)): 0 = three workgroups per CU, 7 = two (the default pads to 54 KB), 70 = one"
WDG_S3_ROUTES=257,2049,18177,257,2049 python3 tools/bench_dgrad_s3.py 32 20 2>/dev/null | grep route
echo "# skeletons at two workgroups per CU (-DWDG_S3_SKELETONS; bits: 1 no weight refills, 2 no dy fragment reads, 4 no epilogue, 8 no DMA of the norm's input)"
WDG_LIB=$PWD/gpurun_variants/libwdgan_s3skel.so WDG_S3_ROUTES=1,3,5,9,17,25,27,31 python3 tools/bench_dgrad_s3.py 32 20 2>/dev/null | grep route
echo "# skeletons at three workgroups per CU"
WDG_LIB=$PWD/gpurun_variants/libwdgan_s3skel.so WDG_S3_ROUTES=257,259,265,273,281,283,287 python3 tools/bench_dgrad_s3.py 32 20 2>/dev/null | grep route
echo "# in-kernel phase clocks (-DWDG_S3_PROF=1, shader clock per wave; the build's own time is distorted by its counters' atomics)"
WDG_S3_PROF=1 WDG_LIB=$PWD/gpurun_variants/libwdgan_s3prof.so WDG_S3_ROUTES=1,257,18177 python3 tools/bench_dgrad_s3.py 32 5 2>/dev/null | grep -A1 "phase clocks" | grep -v "^--"
} > $OUT 2>&1
cat $OUT
