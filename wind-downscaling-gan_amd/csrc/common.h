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

static inline int wdg_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t wdg_ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wdg_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

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
