#!/bin/bash
# Measurement builds of libwdgan.so with the K-loop skeleton switches of conv_igemm.hip (-DWDG_KLOOP_EXP=<bits>: timing only, results
# are wrong by construction) into gpurun_variants/libwdgan_exp<bits>.so; load with WDG_LIB=... (engine/native.py).
#   bash tools/build_exp.sh 1 2 4 8 15
set -e
cd "$(dirname "$0")/../wind-downscaling-gan_amd/csrc"
OUT=../../gpurun_variants; mkdir -p $OUT
OBJS=$(ls *.o | grep -v '^conv_igemm.o$')
for BITS in "$@"; do
  case "$BITS" in
    early) FLAGS="-DWDG_EARLY_LOADS=1";;     # operand requests pinned in front of the MFMAs on the narrow tiles
    prio) FLAGS="-DWDG_MFMA_PRIO=1";;        # s_setprio(1)/(0) around the MFMA clusters
    *) FLAGS="-DWDG_KLOOP_EXP=$BITS";;
  esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $FLAGS -c conv_igemm.hip -o /tmp/igemm_exp$BITS.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libwdgan_exp$BITS.so $OBJS /tmp/igemm_exp$BITS.o
  echo "exp$BITS built"
done
