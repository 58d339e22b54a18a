// norms.hip — BatchNormalization / LayerNormalization forward + backward (HBM-bound kernels).
// Reference call sites: /root/reference/src/downscaling/gan/models.py:34,40,50,56,69 (BatchNorm,
// axis -1, eps 1e-3, momentum .99) and :97,105,116,125,136 (LayerNorm, axis -1, eps 1e-3).
// The LeakyReLU(0.2) that precedes every norm (conv -> bias -> LReLU -> norm) has its derivative
// fused into the norm-backward kernels (sign(y) == sign(pre-activation)).
#include "common.h"
#include <algorithm>

// ---- block-level per-channel reduction helper -------------------------------------------------
// Thread t owns channel group c4 = t % c4n for pixel row t / c4n.  Sums NV float4 values over the
// rows of the block through LDS, then adds them to out[v*C + 4*c4 + j] with one atomic per value.
template <int NV, typename OutT>
__device__ __forceinline__ void wdg_block_colreduce(const float (&v)[NV][4], int c4, int c4n, int rows,
                                                   bool active, OutT* out, int C, float* lds /*[NV*256*4]*/) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[(k * 256 + t) * 4 + j] = active ? v[k][j] : 0.f;
    __syncthreads();
    int live = rows;
    if ((rows & (rows - 1)) == 0) {
        // tree over the pixel rows with all threads (rows is a power of two whenever C/4 is): log2(rows) steps instead
        // of a serial loop by c4n threads (for 16 channels that loop was 4 threads x 64 rows)
        for (int sft = rows >> 1; sft > 0; sft >>= 1) {
            if (active && t < sft * c4n) {
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) lds[(k * 256 + t) * 4 + j] += lds[(k * 256 + t + sft * c4n) * 4 + j];
            }
            __syncthreads();
        }
        live = 1;
    }
    if (t < c4n) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                OutT s = 0;
                for (int r = 0; r < live; ++r) s += (OutT)lds[(k * 256 + r * c4n + t) * 4 + j];
                atomicAdd(&out[(size_t)k * C + 4 * t + j], s);
            }
        }
    }
    __syncthreads();
}

struct ColGeom {
    int c4n, rows;  // channel groups, pixel rows per pass (rows * c4n <= 256)
};
static inline ColGeom col_geom(int C) {
    ColGeom g;
    g.c4n = C / 4;
    g.rows = 256 / g.c4n;
    if (g.rows < 1) g.rows = 1;
    return g;
}
static inline int col_blocks(int64_t P, int rows) {
    int64_t per_block = (int64_t)rows * 64;  // ~64 pixels per thread row
    int64_t b = (P + per_block - 1) / per_block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(b, 4096));
}

// ---- BatchNorm ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wdg_bn_stats_kernel(const float* __restrict__ x, int64_t P, int C,
                                                           int ldx, double* stats, int c4n, int rows) {
    __shared__ float lds[2 * 256 * 4];
    const int t = threadIdx.x;
    const int c4 = t % c4n, prow = t / c4n;
    const bool active = prow < rows;
    float v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (active) {
        // eight pixels per trip, loads first (pure stream: keep several 16-byte loads in flight per lane)
        const int64_t stride = (int64_t)gridDim.x * rows;
        for (int64_t p = (int64_t)blockIdx.x * rows + prow; p < P; p += 8 * stride) {
            f32x4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                a[u] = p + u * stride < P ? *reinterpret_cast<const f32x4*>(x + (p + u * stride) * ldx + 4 * c4)
                                          : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[0][j] += a[u][j];
                    v[1][j] += a[u][j] * a[u][j];
                }
        }
    }
    wdg_block_colreduce<2, double>(v, c4, c4n, rows, active, stats, C, lds);
}

extern "C" int wdg_bn_stats(const float* x, int64_t P, int C, int ldx, double* stats, wdg_stream stream) {
    WDG_CHECK_ARG(x && stats && C % 4 == 0 && C <= 1024 && ldx % 4 == 0, "bad argument");
    ColGeom g = col_geom(C);
    // few blocks: every block ends with 2*C fp64 atomics on the same addresses, and 2048-8192 blocks made that
    // serialised tail longer than the streaming pass (profiles/r01ah); fp32 partials stay short enough (<= ~512 values)
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((P + g.rows * 32 - 1) / (g.rows * 32), 512));
    hipLaunchKernelGGL(wdg_bn_stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, P, C, ldx,
                       stats, g.c4n, g.rows);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// one wave per channel: the lanes stride over the replica slabs [replicas][2][C] the producers spread their atomics
// over (fixed order per lane, then a wave butterfly), lane 0 finalises
__global__ void __launch_bounds__(256) wdg_bn_finalize_train_kernel(const double* stats, int replicas, double count, const float* gamma,
                                                                    const float* beta, float* mmean, float* mvar, float momentum,
                                                                    float eps, float* ss, float* saved, int C) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int r = lane; r < replicas; r += 64) {
        s1 += stats[(size_t)r * 2 * C + c];
        s2 += stats[(size_t)r * 2 * C + C + c];
    }
    s1 = wdg_wave_sum_d(s1);
    s2 = wdg_wave_sum_d(s2);
    if (lane) return;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float scale = (float)((double)gamma[c] * invstd);
    ss[c] = scale;
    ss[C + c] = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
    saved[c] = (float)mean;
    saved[C + c] = (float)invstd;
    mmean[c] = mmean[c] * momentum + (float)mean * (1.f - momentum);
    // TF 2.4 runs BatchNormalization on 5-D input through the fused op (ndims in (4, 5)), whose batch_variance output —
    // the value Keras folds into moving_variance — carries Bessel's correction N / (N - 1); the normalisation itself
    // uses the biased variance above.
    const double var_unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    mvar[c] = mvar[c] * momentum + (float)var_unbiased * (1.f - momentum);
}

extern "C" int wdg_bn_finalize_train(const double* stats, int replicas, double count, const float* gamma, const float* beta,
                                     float* moving_mean, float* moving_var, float momentum, float eps,
                                     float* scale_shift, float* saved_mean_invstd, int C, wdg_stream stream) {
    WDG_CHECK_ARG(stats && gamma && beta && moving_mean && moving_var && scale_shift && saved_mean_invstd && replicas >= 1,
                  "null argument");
    hipLaunchKernelGGL(wdg_bn_finalize_train_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       stats, replicas, count, gamma, beta, moving_mean, moving_var, momentum, eps, scale_shift,
                       saved_mean_invstd, C);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// stats[0][:] = sum_r stats[r][:] (in place): the data-parallel exchange of the batch statistics then moves 2*C values
__global__ void __launch_bounds__(256) wdg_bn_collapse_kernel(double* stats, int replicas, int C2) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C2) return;
    double s = 0;
    for (int r = lane; r < replicas; r += 64) s += stats[(size_t)r * C2 + c];
    s = wdg_wave_sum_d(s);
    if (lane == 0) stats[c] = s;
}

extern "C" int wdg_bn_collapse(double* stats, int replicas, int C, wdg_stream stream) {
    WDG_CHECK_ARG(stats && replicas >= 1 && C > 0, "bad argument");
    hipLaunchKernelGGL(wdg_bn_collapse_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, (hipStream_t)stream, stats, replicas, 2 * C);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

__global__ void wdg_bn_finalize_infer_kernel(const float* gamma, const float* beta, const float* mmean,
                                             const float* mvar, float eps, float* ss, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double invstd = 1.0 / sqrt((double)mvar[c] + (double)eps);
    ss[c] = (float)((double)gamma[c] * invstd);
    ss[C + c] = (float)((double)beta[c] - (double)mmean[c] * (double)gamma[c] * invstd);
}

extern "C" int wdg_bn_finalize_infer(const float* gamma, const float* beta, const float* moving_mean,
                                     const float* moving_var, float eps, float* scale_shift, int C,
                                     wdg_stream stream) {
    WDG_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale_shift, "null argument");
    hipLaunchKernelGGL(wdg_bn_finalize_infer_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       gamma, beta, moving_mean, moving_var, eps, scale_shift, C);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

__global__ void __launch_bounds__(256) wdg_bn_apply_kernel(const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ ss, float* z, int ldz,
                                                           int64_t P, int C, wdg_fastdiv div_c4n) {
    const int c4n = C / 4;
    const int64_t total = P * c4n;
    const bool small = total < (1LL << 31);          // index split by multiply-high instead of a 64-bit division per float4
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = small ? (int64_t)wdg_fastdiv_do((unsigned)idx, div_c4n) : idx / c4n;
        const int c = 4 * (int)(idx - p * c4n);
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + p * ldx + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + C + c);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = a[j] * sc[j] + sh[j];
        *reinterpret_cast<f32x4*>(z + p * ldz + c) = r;
    }
}

extern "C" int wdg_bn_apply(const float* x, int ldx, const float* scale_shift, float* z, int ldz, int64_t P,
                            int C, wdg_stream stream) {
    WDG_CHECK_ARG(x && scale_shift && z && C % 4 == 0 && ldx % 4 == 0 && ldz % 4 == 0, "bad argument");
    const int64_t total = P * (C / 4);
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 16384));
    hipLaunchKernelGGL(wdg_bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       scale_shift, z, ldz, P, C, wdg_fastdiv_make((unsigned)(C / 4)));
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

static int g_bn_bwd_blocks = 256;         // wdg_set_tuning("bn_bwd_blocks", n): grid cap of the BatchNorm backward passes (reduce: n, apply: 2 n).  Swept in round 5 (profiles/r05u_perf_bn_bwd_blocks.txt): 64 -> 182 us, 128 -> 103, 256 -> 86, 512 -> 102, 1024 -> 140 on the 16 @ 256^2 tensor - beyond one block per CU the per-block tail (column reduce + atomics on 2 C addresses) outweighs the extra waves
void wdg_bn_set_bwd_blocks(int v) { g_bn_bwd_blocks = v > 0 ? v : 256; }
constexpr int BNB_U = 4;      // pixels per trip and thread of the BatchNorm backward passes (2 x BNB_U 16-byte loads in flight)
__global__ void __launch_bounds__(256) wdg_bn_bwd_reduce_kernel(const float* __restrict__ dz, int lddz,
                                                                const float* __restrict__ y, int ldy,
                                                                const float* __restrict__ saved, int64_t P,
                                                                int C, double* red, int c4n, int rows) {
    __shared__ float lds[2 * 256 * 4];
    const int t = threadIdx.x;
    const int c4 = t % c4n, prow = t / c4n;
    const bool active = prow < rows;
    float v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (active) {
        const f32x4 mean = *reinterpret_cast<const f32x4*>(saved + 4 * c4);
        const f32x4 inv = *reinterpret_cast<const f32x4*>(saved + C + 4 * c4);
        const int64_t stride = (int64_t)gridDim.x * rows;
        for (int64_t p = (int64_t)blockIdx.x * rows + prow; p < P; p += BNB_U * stride) {
            // four pixels per trip, the eight loads first and UNCONDITIONAL (tail pixels re-read the last one and are masked out of
            // the sums): `if (q < P) load` compiles to an exec-masked block per load with a full wait at its join — one round trip
            // after the other, and with two waves per SIMD (the fp64 atomics keep the grid small) nothing hides them: 2.1 TB/s
            f32x4 g[BNB_U], a[BNB_U];
            float m[BNB_U];
#pragma unroll
            for (int u = 0; u < BNB_U; ++u) {
                const int64_t q = p + u * stride;
                const int64_t qq = q < P ? q : P - 1;
                m[u] = q < P ? 1.f : 0.f;
                g[u] = *reinterpret_cast<const f32x4*>(dz + qq * lddz + 4 * c4);
                a[u] = *reinterpret_cast<const f32x4*>(y + qq * ldy + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < BNB_U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (a[u][j] - mean[j]) * inv[j];
                    const float gm = g[u][j] * m[u];
                    v[0][j] += gm;
                    v[1][j] += gm * xh;
                }
        }
    }
    wdg_block_colreduce<2, double>(v, c4, c4n, rows, active, red, C, lds);
}

extern "C" int wdg_bn_bwd_reduce(const float* dz, int lddz, const float* y, int ldy,
                                 const float* saved_mean_invstd, int64_t P, int C, double* red,
                                 wdg_stream stream) {
    WDG_CHECK_ARG(dz && y && saved_mean_invstd && red && C % 4 == 0 && C <= 1024 && lddz % 4 == 0 && ldy % 4 == 0,
                  "bad argument");
    ColGeom g = col_geom(C);
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((P + g.rows * 32 - 1) / (g.rows * 32), g_bn_bwd_blocks));   // see wdg_bn_stats
    hipLaunchKernelGGL(wdg_bn_bwd_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, lddz, y,
                       ldy, saved_mean_invstd, P, C, red, g.c4n, g.rows);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

__global__ void __launch_bounds__(256) wdg_bn_bwd_apply_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ y, int ldy,
    const float* __restrict__ saved, const float* __restrict__ gamma, const double* __restrict__ red_mean,
    const double* __restrict__ red_param, double count, float act_slope, float* dpre, int lddpre, float* dgamma,
    float* dbeta, float* dbias, int64_t P, int C, int c4n, int rows) {
    __shared__ float lds[256 * 4];
    const int t = threadIdx.x;
    const int c4 = t % c4n, prow = t / c4n;
    const bool active = prow < rows;
    float v[1][4] = {{0, 0, 0, 0}};
    if (active) {
        const f32x4 mean = *reinterpret_cast<const f32x4*>(saved + 4 * c4);
        const f32x4 inv = *reinterpret_cast<const f32x4*>(saved + C + 4 * c4);
        f32x4 k0, mdz, mdzx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            k0[j] = gamma[4 * c4 + j] * inv[j];
            mdz[j] = (float)(red_mean[4 * c4 + j] / count);
            mdzx[j] = (float)(red_mean[C + 4 * c4 + j] / count);
        }
        // four pixels per trip with the loads first (dpre may alias dz: each thread only touches its own slots)
        const int64_t stride = (int64_t)gridDim.x * rows;
        for (int64_t p = (int64_t)blockIdx.x * rows + prow; p < P; p += BNB_U * stride) {
            f32x4 g[BNB_U], a[BNB_U];
#pragma unroll
            for (int u = 0; u < BNB_U; ++u) {
                const int64_t q = p + u * stride;
                const int64_t qq = q < P ? q : P - 1;                  // (unconditional loads: see wdg_bn_bwd_reduce_kernel)
                g[u] = *reinterpret_cast<const f32x4*>(dz + qq * lddz + 4 * c4);
                a[u] = *reinterpret_cast<const f32x4*>(y + qq * ldy + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < BNB_U; ++u) {
                const int64_t q = p + u * stride;
                if (q >= P) continue;
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (a[u][j] - mean[j]) * inv[j];
                    float d = k0[j] * (g[u][j] - mdz[j] - xh * mdzx[j]);
                    if (act_slope >= 0.f) d *= (a[u][j] > 0.f ? 1.f : act_slope);
                    r[j] = d;
                    v[0][j] += d;
                }
                *reinterpret_cast<f32x4*>(dpre + q * lddpre + 4 * c4) = r;
            }
        }
    }
    if (dbias) wdg_block_colreduce<1, float>(v, c4, c4n, rows, active, dbias, C, lds);
    if (blockIdx.x == 0 && t < C && red_param) {
        if (dgamma) dgamma[t] += (float)red_param[C + t];
        if (dbeta) dbeta[t] += (float)red_param[t];
    }
    if (blockIdx.x == 0 && red_param && C > 256) {
        for (int c = 256 + t; c < C; c += 256) {
            if (dgamma) dgamma[c] += (float)red_param[C + c];
            if (dbeta) dbeta[c] += (float)red_param[c];
        }
    }
}

extern "C" int wdg_bn_bwd_apply(const float* dz, int lddz, const float* y, int ldy,
                                const float* saved_mean_invstd, const float* gamma, const double* red_mean,
                                const double* red_param, double count, float act_slope, float* dpre,
                                int lddpre, float* dgamma, float* dbeta, float* dbias, int64_t P, int C,
                                wdg_stream stream) {
    WDG_CHECK_ARG(dz && y && saved_mean_invstd && gamma && red_mean && dpre, "null argument");
    WDG_CHECK_ARG(C % 4 == 0 && C <= 1024 && lddz % 4 == 0 && ldy % 4 == 0 && lddpre % 4 == 0, "bad sizes");
    ColGeom g = col_geom(C);
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((P + g.rows * 16 - 1) / (g.rows * 16), 2 * g_bn_bwd_blocks));
    hipLaunchKernelGGL(wdg_bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, lddz, y,
                       ldy, saved_mean_invstd, gamma, red_mean, red_param, count, act_slope, dpre, lddpre,
                       dgamma, dbeta, dbias, P, C, g.c4n, g.rows);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- LayerNorm ---------------------------------------------------------------------------------
// L lanes (power of two <= 64) cooperate on one pixel; each lane owns channel groups sub, sub+L, ...
template <int MAXCH>  // max float4 chunks per lane
__global__ void __launch_bounds__(256) wdg_ln_fwd_kernel(const float* __restrict__ y, int ldy,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float* z,
                                                         int ldz, float* mean_rstd, int64_t P, int C, int L) {
    const int t = threadIdx.x;
    const int sub = t % L;
    const int ppb = 256 / L;
    const int c4n = C / 4;
    const float invC = 1.f / (float)C;
    for (int64_t p = (int64_t)blockIdx.x * ppb + t / L; p < P; p += (int64_t)gridDim.x * ppb) {
        f32x4 a[MAXCH];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int c4 = sub + k * L;
            a[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c4 < c4n) a[k] = *reinterpret_cast<const f32x4*>(y + p * ldy + 4 * c4);
            s += a[k][0] + a[k][1] + a[k][2] + a[k][3];
        }
        for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int c4 = sub + k * L;
            if (c4 < c4n) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = a[k][j] - mean;
                    q += d * d;
                }
            }
        }
        for (int o = L >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.f / sqrtf(q * invC + eps);
#pragma unroll
        for (int k = 0; k < MAXCH; ++k) {
            const int c4 = sub + k * L;
            if (c4 < c4n) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = (a[k][j] - mean) * rstd * g[j] + b[j];
                *reinterpret_cast<f32x4*>(z + p * ldz + 4 * c4) = r;
            }
        }
        if (mean_rstd && sub == 0) {
            mean_rstd[2 * p] = mean;
            mean_rstd[2 * p + 1] = rstd;
        }
    }
}

static inline int ln_lanes(int C) {
    int c4n = C / 4, L = 1;
    while (L < c4n && L < 64) L <<= 1;
    return L;
}

extern "C" int wdg_ln_fwd(const float* y, int ldy, const float* gamma, const float* beta, float eps, float* z,
                          int ldz, float* mean_rstd, int64_t P, int C, wdg_stream stream) {
    WDG_CHECK_ARG(y && gamma && beta && z, "null argument");
    WDG_CHECK_ARG(C % 4 == 0 && C <= 1024 && ldy % 4 == 0 && ldz % 4 == 0, "bad sizes");
    const int L = ln_lanes(C);
    const int ppb = 256 / L;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((P + ppb - 1) / ppb, 16384));
    const int chunks = (C / 4 + L - 1) / L;
    if (chunks <= 1)
        hipLaunchKernelGGL(wdg_ln_fwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, ldy, gamma,
                           beta, eps, z, ldz, mean_rstd, P, C, L);
    else
        hipLaunchKernelGGL(wdg_ln_fwd_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, ldy, gamma,
                           beta, eps, z, ldz, mean_rstd, P, C, L);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

template <int MAXCH>
__global__ void __launch_bounds__(256) wdg_ln_bwd_kernel(const float* __restrict__ dz, int lddz,
                                                         const float* __restrict__ y, int ldy,
                                                         const float* __restrict__ mean_rstd,
                                                         const float* __restrict__ gamma, float act_slope,
                                                         float* dpre, int lddpre, float* dgamma, float* dbeta,
                                                         float* dbias, int64_t P, int C, int L) {
    __shared__ float lds[3 * MAXCH * 256 * 4];
    const int t = threadIdx.x;
    const int sub = t % L;
    const int ppb = 256 / L;
    const int c4n = C / 4;
    const float invC = 1.f / (float)C;
    float ag[MAXCH][4], ab[MAXCH][4], abias[MAXCH][4];
#pragma unroll
    for (int k = 0; k < MAXCH; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) ag[k][j] = ab[k][j] = abias[k][j] = 0.f;
    f32x4 gm[MAXCH];
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int c4 = sub + k * L;
        gm[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c4 < c4n) gm[k] = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
    }
    // U pixels per trip with all their loads issued first: the kernel is a pure stream (two reads, one write per
    // element) and one pixel per trip left it latency-bound at < 2 TB/s.  dpre may alias dz (in-place), which is safe
    // because a thread only ever touches its own (pixel, channel-group) slots — but it is also why the compiler cannot
    // batch the loads by itself.
    constexpr int U = MAXCH == 1 ? 4 : 1;
    const int64_t stride = (int64_t)gridDim.x * ppb;
    for (int64_t p0 = (int64_t)blockIdx.x * ppb + t / L; p0 < P; p0 += U * stride) {
        f32x4 a[U][MAXCH], g[U][MAXCH];
        float mean[U], rstd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + u * stride;
            const bool on = p < P;
            mean[u] = on ? mean_rstd[2 * p] : 0.f;
            rstd[u] = on ? mean_rstd[2 * p + 1] : 0.f;
#pragma unroll
            for (int k = 0; k < MAXCH; ++k) {
                const int c4 = sub + k * L;
                a[u][k] = g[u][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (on && c4 < c4n) {
                    a[u][k] = *reinterpret_cast<const f32x4*>(y + p * ldy + 4 * c4);
                    g[u][k] = *reinterpret_cast<const f32x4*>(dz + p * lddz + 4 * c4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + u * stride;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < MAXCH; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (a[u][k][j] - mean[u]) * rstd[u];
                    const float gg = g[u][k][j] * gm[k][j];
                    s1 += gg;
                    s2 += gg * xh;
                    ag[k][j] += g[u][k][j] * xh;
                    ab[k][j] += g[u][k][j];
                }
            for (int o = L >> 1; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            s1 *= invC;
            s2 *= invC;
            if (p < P) {
#pragma unroll
                for (int k = 0; k < MAXCH; ++k) {
                    const int c4 = sub + k * L;
                    if (c4 < c4n) {
                        f32x4 r;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xh = (a[u][k][j] - mean[u]) * rstd[u];
                            float d = rstd[u] * (g[u][k][j] * gm[k][j] - s1 - xh * s2);
                            if (act_slope >= 0.f) d *= (a[u][k][j] > 0.f ? 1.f : act_slope);
                            r[j] = d;
                            abias[k][j] += d;
                        }
                        *reinterpret_cast<f32x4*>(dpre + p * lddpre + 4 * c4) = r;
                    }
                }
            }
        }
    }
    // reduce the three per-channel accumulators over the ppb pixel slots of the block: lanes of a wave that share a
    // channel group differ in the lane bits >= log2(L) -> xor-shuffles, then the four waves through LDS; one atomic
    // per (block, channel).  (The former serial LDS loop by L threads plus one atomic per block from 2048 blocks on
    // the same 3*C addresses cost more than the streaming pass itself.)
    if (!dgamma && !dbeta && !dbias) return;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0 = ag[k][j], v1 = ab[k][j], v2 = abias[k][j];
            for (int o = L; o < 64; o <<= 1) {
                v0 += __shfl_xor(v0, o, 64);
                v1 += __shfl_xor(v1, o, 64);
                v2 += __shfl_xor(v2, o, 64);
            }
            ag[k][j] = v0; ab[k][j] = v1; abias[k][j] = v2;
        }
    const int lane = t & 63, wave = t >> 6;
    // L <= 64 divides 64, so `sub` = lane % L inside a wave
    if (lane < L) {
#pragma unroll
        for (int k = 0; k < MAXCH; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lds[(((wave * 3 + 0) * MAXCH + k) * 64 + lane) * 4 + j] = ag[k][j];
                lds[(((wave * 3 + 1) * MAXCH + k) * 64 + lane) * 4 + j] = ab[k][j];
                lds[(((wave * 3 + 2) * MAXCH + k) * 64 + lane) * 4 + j] = abias[k][j];
            }
    }
    __syncthreads();
    // thread -> (which, k, sub, j)
    for (int idx = t; idx < 3 * MAXCH * L * 4; idx += 256) {
        const int j = idx & 3;
        const int sub2 = (idx >> 2) % L;
        const int k = ((idx >> 2) / L) % MAXCH;
        const int which = (idx >> 2) / (L * MAXCH);
        const int c4 = sub2 + k * L;
        if (c4 >= c4n) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += lds[(((w * 3 + which) * MAXCH + k) * 64 + sub2) * 4 + j];
        float* dst = which == 0 ? dgamma : which == 1 ? dbeta : dbias;
        if (dst) atomicAdd(&dst[4 * c4 + j], v);
    }
}

extern "C" int wdg_ln_bwd(const float* dz, int lddz, const float* y, int ldy, const float* mean_rstd,
                          const float* gamma, float act_slope, float* dpre, int lddpre, float* dgamma,
                          float* dbeta, float* dbias, int64_t P, int C, wdg_stream stream) {
    WDG_CHECK_ARG(dz && y && mean_rstd && gamma && dpre, "null argument");
    WDG_CHECK_ARG(C % 4 == 0 && C <= 1024 && lddz % 4 == 0 && ldy % 4 == 0 && lddpre % 4 == 0, "bad sizes");
    const int L = ln_lanes(C);
    const int ppb = 256 / L;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((P + ppb * 8 - 1) / (ppb * 8), 1024));
    const int chunks = (C / 4 + L - 1) / L;
    if (chunks <= 1)
        hipLaunchKernelGGL(wdg_ln_bwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, lddz, y,
                           ldy, mean_rstd, gamma, act_slope, dpre, lddpre, dgamma, dbeta, dbias, P, C, L);
    else
        hipLaunchKernelGGL(wdg_ln_bwd_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, lddz, y,
                           ldy, mean_rstd, gamma, act_slope, dpre, lddpre, dgamma, dbeta, dbias, P, C, L);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// Replica slabs [rep][3][C] of the LayerNorm-backward epilogue (EPI 5) -> dgamma / dbeta / dbias (accumulated; any may be NULL),
// and the slabs cleared for the next launch.  One thread per (quantity, channel), replicas summed in order.
__global__ void __launch_bounds__(256) wdg_ln_param_finish_kernel(float* par, int rep, int C, float* dgamma, float* dbeta, float* dbias) {
    // one wave per (quantity, channel): the lanes take the replicas (fixed assignment, then a wave sum) — a thread per value
    // walking the 64 replicas serially took 17 us for 48 values
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= 3 * C) return;
    const int which = idx / C, c = idx - which * C;
    float v = 0.f;
    for (int r = lane; r < rep; r += 64) {
        v += par[(size_t)r * 3 * C + idx];
        par[(size_t)r * 3 * C + idx] = 0.f;
    }
    v = wdg_wave_sum_fast(v);
    float* dst = which == 0 ? dgamma : which == 1 ? dbeta : dbias;
    if (dst && lane == 0) dst[c] += v;
}

int wdg_lnb_finish(float* par, int rep, int C, float* dgamma, float* dbeta, float* dbias, hipStream_t stream) {
    hipLaunchKernelGGL(wdg_ln_param_finish_kernel, dim3((3 * C + 3) / 4), dim3(256), 0, stream, par, rep, C, dgamma, dbeta, dbias);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- Dense(1) + GlobalAveragePooling head backward (models.py:137-140) chained with the backward of the LayerNormalization
// (+ LeakyReLU) that produced the head's input (models.py:125,136): dz[row][k] = dscore[b] / T * w[k] never leaves the registers —
// one wave per pixel of the final map forms its C values of dz, runs the norm's backward on them and writes dpre; the dense
// layer's own dw / db as in wdg_dense_gap_bwd.  x [rows][K] is the norm's OUTPUT (the head's input), y [rows * npix][C] its input.
__global__ void __launch_bounds__(256) wdg_dense_gap_bwd_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ dscore, float* dx, float* dw, float* db,
                                                                   int B, int T, int K, const float* __restrict__ y,
                                                                   const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                   int C, float act_slope, float* par, int rep) {
    __shared__ float red[4 * 3 * 1024];
    const float invT = 1.f / (float)T;
    const int npix = K / C, rows = B * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4n = C >> 2;
    const float invC = 1.f / (float)C;
    f32x4 pg[4], pb[4], pd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pg[j] = pb[j] = pd[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int m = blockIdx.x * 4 + wave; m < rows * npix; m += gridDim.x * 4) {
        const int r = m / npix, q = m - r * npix;
        const float ds = dscore[r % B] * invT;           // row = t * B + b
        const float mean = mean_rstd[2 * (long long)m], rstd = mean_rstd[2 * (long long)m + 1];
        f32x4 v[4], yy[4], g[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c4 = lane + 64 * j;
            v[j] = yy[j] = g[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c4 < c4n) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + (long long)q * C + 4 * c4);
                yy[j] = *reinterpret_cast<const f32x4*>(y + (long long)m * C + 4 * c4);
                g[j] = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[j][k] = ds * w4[k];
                    const float xh = (yy[j][k] - mean) * rstd, gg = v[j][k] * g[j][k];
                    s1 += gg;
                    s2 += gg * xh;
                }
            }
        }
        s1 = wdg_wave_sum(s1) * invC;
        s2 = wdg_wave_sum(s2) * invC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c4 = lane + 64 * j;
            if (c4 >= c4n) continue;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (yy[j][k] - mean) * rstd;
                float d = rstd * (v[j][k] * g[j][k] - s1 - xh * s2);
                if (act_slope >= 0.f) d *= (yy[j][k] > 0.f ? 1.f : act_slope);
                pg[j][k] = fmaf(v[j][k], xh, pg[j][k]);
                pb[j][k] += v[j][k];
                pd[j][k] += d;
                o[k] = d;
            }
            *reinterpret_cast<f32x4*>(dx + (long long)m * C + 4 * c4) = o;
        }
    }
    // the dense layer's own gradients (as wdg_dense_gap_bwd_kernel)
    if (dw) {
        for (int k = blockIdx.x * 256 + threadIdx.x; k < K; k += gridDim.x * 256) {
            float s = 0.f;
            for (int r = 0; r < rows; ++r) s += x[(int64_t)r * K + k] * dscore[r % B];
            dw[k] += s * invT;
        }
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dscore[b];
        db[0] += s;
    }
    if (!par) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < c4n) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                red[(wave * 3 + 0) * 1024 + 4 * c4 + k] = pg[j][k];
                red[(wave * 3 + 1) * 1024 + 4 * c4 + k] = pb[j][k];
                red[(wave * 3 + 2) * 1024 + 4 * c4 + k] = pd[j][k];
            }
        }
    }
    __syncthreads();
    float* slab = par + (size_t)(blockIdx.x % (unsigned)rep) * 3 * C;
    for (int idx = threadIdx.x; idx < 3 * C; idx += 256) {
        const int which = idx / C, n = idx - which * C;
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) t += red[(wv * 3 + which) * 1024 + n];
        atomicAdd(slab + idx, t);
    }
}

extern "C" int wdg_dense_gap_bwd_ln(const float* x, const float* w, const float* dscore, float* dx, float* dw, float* db, int B, int T,
                                    int K, const float* y, const float* mean_rstd, const float* gamma, int C, float act_slope,
                                    float* dgamma, float* dbeta, float* dbias, float* par_ws, wdg_stream stream) {
    WDG_CHECK_ARG(x && w && dscore && dx && y && mean_rstd && gamma, "null argument");
    WDG_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 1024 && K % C == 0, "the head's input must be whole pixels of C <= 1024 channels, C % 4 == 0");
    WDG_CHECK_ARG(((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)gamma & 15) == 0, "16-byte alignment");
    const bool want_par = dgamma || dbeta || dbias;
    WDG_CHECK_ARG(!want_par || par_ws, "parameter gradients need the scratch");
    const int64_t rows_px = (int64_t)B * T * (K / C);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((rows_px + 3) / 4, 1024));
    hipLaunchKernelGGL(wdg_dense_gap_bwd_ln_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, dscore, dx, dw, db, B, T, K, y,
                       mean_rstd, gamma, C, act_slope, want_par ? par_ws : nullptr, WDG_LNB_REP);
    WDG_LAUNCH_CHECK();
    if (!want_par) return WDG_OK;
    return wdg_lnb_finish(par_ws, WDG_LNB_REP, C, dgamma, dbeta, dbias, (hipStream_t)stream);
}
