#!/bin/bash
# builds libwdgan variants of conv_patch_h16.hip (compile-time knobs) into gpurun_variants/ for A/B runs on the GPU box
set -e
cd "$(dirname "$0")/../wind-downscaling-gan_amd/csrc"
OUT=../../gpurun_variants; mkdir -p $OUT
OBJS=$(ls *.o | grep -v conv_patch_h16.o)
i=0
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $FLAGS -c conv_patch_h16.hip -o /tmp/patch_v$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libwdgan_v$i.so $OBJS /tmp/patch_v$i.o
  echo "v$i: $FLAGS"
  i=$((i+1))
done
