// common.h — shared helpers for libwdgan (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/wdgan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void wdg_set_error(const char* fmt, ...);

#define WDG_CHECK_ARG(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            wdg_set_error("%s: %s", __func__, msg);                \
            return WDG_ERR_ARG;                                    \
        }                                                          \
    } while (0)

#define WDG_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            wdg_set_error("%s: %s -> %s", __func__, #call, hipGetErrorString(e_)); \
            return WDG_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define WDG_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            wdg_set_error("%s: launch -> %s", __func__, hipGetErrorString(e_));    \
            return WDG_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

// exact unsigned division n / d for n < 2^31 by a multiply-high and a shift (d >= 1)
struct wdg_fastdiv {
    unsigned m, s;   // m == 0 marks d == 1
};
static inline wdg_fastdiv wdg_fastdiv_make(unsigned d) {
    wdg_fastdiv f = {0u, 0u};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;                       // s = ceil(log2 d) >= 1
    f.m = (unsigned)(((1ull << (31 + s)) / d) + 1);    // < 2^32 because d > 2^(s-1)
    f.s = s - 1;
    return f;
}

static inline int wdg_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t wdg_ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ unsigned wdg_fastdiv_do(unsigned n, wdg_fastdiv f) {
    return f.m ? __umulhi(n, f.m) >> f.s : n;
}

__device__ __forceinline__ float wdg_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// tanh(x) = 1 - 2 / (1 + exp(2x)) on the hardware exp2 / rcp: 4 instructions (libm's tanhf is ~40 with two branches), absolute
// error ~1e-7 (relative error grows below |x| ~ 1e-3, where the result is added to or multiplied with O(1) quantities in every
// LSTM cell that uses it).  Saturates cleanly: exp -> inf gives 1, exp -> 0 gives -1.
__device__ __forceinline__ float wdg_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + e), 1.f);
}

// wave-level sum (64 lanes)
__device__ __forceinline__ float wdg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wdg_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane of the row: four v_add_f32 with DPP
// operands (quad swaps, half-row mirror, row mirror) instead of four ds_bpermute round trips
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define WDG_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))
__device__ __forceinline__ float wdg_row16_sum(float v) {
    v = WDG_DPP_ADD(v, 0xB1);    // quad_perm [1,0,3,2]
    v = WDG_DPP_ADD(v, 0x4E);    // quad_perm [2,3,0,1]
    v = WDG_DPP_ADD(v, 0x141);   // row_half_mirror
    v = WDG_DPP_ADD(v, 0x140);   // row_mirror
    return v;
}
// wave sum through the row sums: four scalar reads of the row results
__device__ __forceinline__ float wdg_wave_sum_fast(float v) {
    v = wdg_row16_sum(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
}
#endif

// ---- buffer (SRD) loads: 32-bit byte offsets relative to a wave-uniform base, hardware range check.
// The descriptor spans 2 GiB; an offset of WDG_SRD_OOB is out of range and the load returns zeros
// (cdna_hip_programming.md T8/T20: build the descriptor only from kernel arguments / blockIdx values).
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef __amdgpu_buffer_rsrc_t wdg_srd;
#define WDG_SRD_BYTES 0x7FFFFFFF
#define WDG_SRD_OOB 0x80000000u
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wdg_srd wdg_make_srd(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, WDG_SRD_BYTES, 0x00020000);
}
__device__ __forceinline__ f32x4 wdg_buffer_load_f32x4(wdg_srd srd, unsigned byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, (int)byte_off, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
#endif

#if defined(__HIPCC__)
// Bijective XCD-aware remap (cdna_hip_programming.md T1): hardware deals consecutive workgroups round-robin
// over the 8 XCDs; give XCD x the contiguous tile range [x*q + min(x,r), ...) so that neighbouring output
// tiles, which share input rows (kh > stride) and the same filter panel, hit the same 4 MiB L2.
__device__ __forceinline__ int wdg_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

#endif

// ---- LayerNorm-backward parameter gradients of the fused launches (conv_igemm.hip EPI 5, the split-K second stage, the dense head):
// blocks add their partial sums into one of WDG_LNB_REP replica slabs [3][C] (dgamma, dbeta, dbias); wdg_lnb_finish (norms.hip) sums
// the replicas into the gradient vectors and clears them
#define WDG_LNB_REP 64
#if defined(__HIPCC__)
int wdg_lnb_finish(float* par, int rep, int C, float* dgamma, float* dbeta, float* dbias, hipStream_t stream);
#endif

