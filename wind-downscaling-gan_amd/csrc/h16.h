// h16.h — the two 16-bit operand formats shared by the inference-precision kernels (template parameter FMT):
// 0 = bf16 (v_mfma_f32_16x16x32_bf16), 1 = IEEE fp16 (v_mfma_f32_16x16x32_f16).  Both round to nearest even from the
// fp32 activations while staging and accumulate in fp32.
#pragma once
#include "common.h"

template <int FMT> struct WdgH16;
template <> struct WdgH16<0> { typedef __bf16 T; };
template <> struct WdgH16<1> { typedef _Float16 T; };
template <int FMT> using wdg_h16 = typename WdgH16<FMT>::T;
template <int FMT> using wdg_h16x8 = wdg_h16<FMT> __attribute__((ext_vector_type(8)));
template <int FMT>
__device__ __forceinline__ f32x4 wdg_mfma16(const wdg_h16x8<FMT>& a, const wdg_h16x8<FMT>& b, const f32x4& c) {
    if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// two float4 (8 consecutive channels) -> one 16-byte operand slot
template <int FMT>
__device__ __forceinline__ wdg_h16x8<FMT> wdg_pack_h16(const f32x4& a, const f32x4& b) {
    wdg_h16x8<FMT> v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (wdg_h16<FMT>)a[j];
        v[4 + j] = (wdg_h16<FMT>)b[j];
    }
    return v;
}
