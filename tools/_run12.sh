O=gpurun_out/r04l_step16_skeletons.txt
: > $O
export WDG_LIB=$PWD/gpurun_variants/libwdgan_exp.so
for D in 0 1 2 3 4 16 32 64 80 83 87 119; do
  python tools/perf_step16.py bf16 patch_dbg=$D 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
