// sn_adam.hip — spectral-normalisation power iteration (tfa.layers.SpectralNormalization as used at
// /root/reference/src/downscaling/gan/models.py:33,39,49,55,95,103,114,123,134) and the TF-form Adam
// update (/root/reference/src/downscaling/gan/train.py:34-35,57-58), plus library bookkeeping.
//
// The SN update must be bit-identical on every data-parallel rank (weights are replicated and the
// update is a function of the weights only), so every reduction below runs in a fixed order:
// no float atomics.
#include "common.h"
#include <algorithm>
#include <stdarg.h>

static thread_local char g_err[512] = "";
void wdg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* wdg_last_error(void) { return g_err; }
extern "C" const char* wdg_version(void) { return "wdgan 0.1 gfx950"; }

// ---- SN step 1: vraw[r] = <u, W[r,:]>, per-block partial sum of squares ---------------------------
__global__ void __launch_bounds__(256) wdg_sn_rowdot_kernel(const float* __restrict__ w,
                                                            const float* __restrict__ u, int rows, int cols,
                                                            float* vraw, float* part1) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sq = 0.f;
    // each wave owns rows blockIdx.x*32 + wave*8 .. +8
    for (int i = 0; i < 8; ++i) {
        const int r = blockIdx.x * 32 + wave * 8 + i;
        if (r >= rows) break;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += u[c] * w[(size_t)r * cols + c];
        s = wdg_wave_sum(s);
        if (lane == 0) {
            vraw[r] = s;
            sq += s * s;
        }
    }
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part1[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- SN step 2: per 64-row chunk, part2[chunk][c] = sum_r v[r] W[r][c] ----------------------------
__global__ void __launch_bounds__(256) wdg_sn_colpart_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ vraw,
                                                             const float* __restrict__ part1, int nb1, int rows,
                                                             int cols, float* part2) {
    __shared__ float vs[64];
    __shared__ float s_scale;
    if (threadIdx.x == 0) {
        float n2 = 0.f;
        for (int i = 0; i < nb1; ++i) n2 += part1[i];  // fixed order
        s_scale = 1.f / sqrtf(fmaxf(n2, 1e-12f));      // tf.math.l2_normalize
    }
    __syncthreads();
    const int r0 = blockIdx.x * 64;
    if (threadIdx.x < 64) vs[threadIdx.x] = (r0 + threadIdx.x < rows) ? vraw[r0 + threadIdx.x] * s_scale : 0.f;
    __syncthreads();
    const int nr = min(64, rows - r0);
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += vs[r] * w[(size_t)(r0 + r) * cols + c];
        part2[(size_t)blockIdx.x * cols + c] = s;
    }
}

// ---- SN step 3 (one block): u_raw, its norm, sigma; writes u and 1/sigma -------------------------
__global__ void __launch_bounds__(1024) wdg_sn_finish_kernel(const float* __restrict__ part2, int nchunks,
                                                             int cols, float* u, float* inv_sigma) {
    __shared__ float red[1024];
    __shared__ float s_norm2;
    // each thread owns columns t, t+1024, ...
    float local = 0.f;
    for (int c = threadIdx.x; c < cols; c += 1024) {
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part2[(size_t)k * cols + c];
        u[c] = s;  // u_raw, normalised below
        local += s * s;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) s_norm2 = red[0];
    __syncthreads();
    const float n2 = s_norm2;
    const float sc = 1.f / sqrtf(fmaxf(n2, 1e-12f));
    for (int c = threadIdx.x; c < cols; c += 1024) u[c] = u[c] * sc;
    // sigma = <u_raw, u_new> = n2 * sc
    if (threadIdx.x == 0) inv_sigma[0] = 1.f / (n2 * sc);
}

__global__ void __launch_bounds__(256) wdg_scale_inplace_kernel(float* w, int64_t n, const float* __restrict__ s) {
    const float k = s[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) w[i] *= k;
}

extern "C" size_t wdg_sn_scratch_floats(int rows, int cols) {
    const size_t nb1 = (rows + 31) / 32, nchunks = (rows + 63) / 64;
    return (size_t)rows + nb1 + nchunks * (size_t)cols + 8;
}

extern "C" int wdg_sn_power_iter(float* w, float* u, int rows, int cols, float* scratch, wdg_stream stream) {
    WDG_CHECK_ARG(w && u && scratch && rows > 0 && cols > 0, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb1 = (rows + 31) / 32, nchunks = (rows + 63) / 64;
    float* vraw = scratch;
    float* part1 = vraw + rows;
    float* part2 = part1 + nb1;
    float* inv_sigma = part2 + (size_t)nchunks * cols;
    hipLaunchKernelGGL(wdg_sn_rowdot_kernel, dim3(nb1), dim3(256), 0, st, w, u, rows, cols, vraw, part1);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_sn_colpart_kernel, dim3(nchunks), dim3(256), 0, st, w, vraw, part1, nb1, rows, cols,
                       part2);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_sn_finish_kernel, dim3(1), dim3(1024), 0, st, part2, nchunks, cols, u, inv_sigma);
    WDG_LAUNCH_CHECK();
    const int64_t n = (int64_t)rows * cols;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(wdg_scale_inplace_kernel, dim3(blocks), dim3(256), 0, st, w, n, inv_sigma);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- Adam, TF form --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wdg_adam_tf_kernel(float* p, const float* __restrict__ g, float* m,
                                                          float* v, int64_t n, float lr_t, float b1, float b2,
                                                          float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);
        const float vi = v[i] + (1.f - b2) * (gi * gi - v[i]);
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int wdg_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                           float beta2, float eps, float grad_scale, wdg_stream stream) {
    WDG_CHECK_ARG(p && g && m && v && n >= 0, "bad argument");
    if (n == 0) return WDG_OK;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192));
    hipLaunchKernelGGL(wdg_adam_tf_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t,
                       beta1, beta2, eps, grad_scale);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
