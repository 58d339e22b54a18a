// metrics.hip — the generator evaluation metrics `get_network` compiles into the GAN
// (/root/reference/src/downscaling/gan/metrics.py:32-45 wind-speed-weighted RMSE, :66-73 extreme-weighted RMSE,
// :79-88 wind-speed RMSE, :94-105 angular cosine distance / opposite cosine similarity, :121-137 log-spectral
// distance, :155-187 spatially convolved KS statistic; wired in at api.py:77-81, evaluated by ganbase.py:71 on every
// train step) as fused reductions: ONE pass over (real, generated) winds produces the per-sample sums of all the
// pointwise metrics; the log-spectral distance reduces the two spectra (rocFFT output) in one pass; the spatial KS
// statistic — in the reference 100 CDF evaluations per patch position through tfp.Empirical, by far the most expensive
// thing a train step triggers — becomes a quantisation pass plus one histogram scan per patch position.
// All HBM-bound: 16 B per pixel (pointwise), 16 B per spectral bin (LSD), 2 B per pixel + LDS work (KS).
#include "common.h"
#include <algorithm>

// ---- pointwise metrics --------------------------------------------------------------------------------------------
// out[b][0] = sum tau*((u^-beta u)^2 + (v^-beta v)^2)      wind_speed_weighted_rmse  (NaN terms -> 0)
// out[b][1] = sum (|w| - |w^|)^2                           wind_speed_rmse           (NaN terms -> 0)
// out[b][2] = sum acos(clip(cos, -1, 1)) / pi              angular_cosine_distance
// out[b][3] = sum 0.5 * (1 - cos)                          opposite_cosine_similarity
// out[b][4] = sum u^2 + v^2                                extreme_weighted_rmse: weight normaliser
// out[b][5] = sum u^2 (u-u^)^2 + v^2 (v-v^)^2              extreme_weighted_rmse: numerator (NaN terms -> 0)
// cos follows tf.keras.losses.cosine_similarity: both vectors l2-normalised with x * rsqrt(max(sum x^2, 1e-12)).
__global__ void __launch_bounds__(256) wdg_metrics_pointwise_kernel(const float* __restrict__ real,
                                                                    const float* __restrict__ fake, int64_t P,
                                                                    double* out) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int b = blockIdx.y;
    const f32x2* r2 = reinterpret_cast<const f32x2*>(real) + (int64_t)b * P;
    const f32x2* f2 = reinterpret_cast<const f32x2*>(fake) + (int64_t)b * P;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const f32x2 r = r2[p], f = f2[p];
        const float u = r[0], v = r[1], uh = f[0], vh = f[1];
        const float est = sqrtf(uh * uh + vh * vh), rea = sqrtf(u * u + v * v);
        const float beta = (4.f + rea) / (4.f + est);
        const float tau = est >= rea ? 0.425f : 1.f - 0.425f;
        const float du = uh - beta * u, dv = vh - beta * v;
        const float wsw = tau * (du * du + dv * dv);
        const float wsr = (rea - est) * (rea - est);
        const float nr = rsqrtf(fmaxf(u * u + v * v, 1e-12f)), nf = rsqrtf(fmaxf(uh * uh + vh * vh, 1e-12f));
        const float c = (u * nr) * (uh * nf) + (v * nr) * (vh * nf);
        const float cb = fminf(fmaxf(c, -1.f), 1.f);
        const float eu = u * u * (u - uh) * (u - uh), ev = v * v * (v - vh) * (v - vh);
        acc[0] += wsw != wsw ? 0.0 : (double)wsw;
        acc[1] += wsr != wsr ? 0.0 : (double)wsr;
        acc[2] += (double)(acosf(cb) * 0.31830988618379067f);
        acc[3] += (double)(0.5f * (1.f - c));
        acc[4] += (double)(u * u) + (double)(v * v);
        acc[5] += (eu != eu ? 0.0 : (double)eu) + (ev != ev ? 0.0 : (double)ev);
    }
    __shared__ double red[4][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double s = wdg_wave_sum_d(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) atomicAdd(&out[b * 6 + threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int wdg_metrics_pointwise(const float* real, const float* fake, int64_t pixels_per_sample, int B, double* out6,
                                     wdg_stream stream) {
    WDG_CHECK_ARG(real && fake && out6 && B > 0 && pixels_per_sample > 0, "bad argument");
    WDG_CHECK_ARG(((uintptr_t)real & 7) == 0 && ((uintptr_t)fake & 7) == 0, "real / fake must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    WDG_HIP(hipMemsetAsync(out6, 0, sizeof(double) * 6 * B, st));
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((pixels_per_sample + 2047) / 2048, 512));
    hipLaunchKernelGGL(wdg_metrics_pointwise_kernel, dim3(bx, B), dim3(256), 0, st, real, fake, pixels_per_sample, out6);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- log-spectral distance: out[b] = sum_bins (10 log10((|R|^2 + eps) / (|F|^2 + eps)))^2 ----------------------------
// spectra: interleaved complex64 [B][n_per_sample] (the rfft2d outputs).  log10(x) = divide_no_nan(log x, log 10).
__global__ void __launch_bounds__(256) wdg_lsd_reduce_kernel(const float* __restrict__ sr, const float* __restrict__ sf,
                                                             int64_t n, float eps, double* out) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int b = blockIdx.y;
    const f32x2* r2 = reinterpret_cast<const f32x2*>(sr) + (int64_t)b * n;
    const f32x2* f2 = reinterpret_cast<const f32x2*>(sf) + (int64_t)b * n;
    double acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const f32x2 r = r2[i], f = f2[i];
        const float pr = r[0] * r[0] + r[1] * r[1] + eps, pf = f[0] * f[0] + f[1] * f[1] + eps;
        const float ratio = pf == 0.f ? 0.f : pr / pf;                       // tf.math.divide_no_nan
        const float l = 10.f * (logf(ratio) / 2.302585092994046f);
        acc += (double)(l * l);
    }
    __shared__ double red[4];
    const double s = wdg_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[b], red[0] + red[1] + red[2] + red[3]);
}

extern "C" int wdg_lsd_reduce(const float* spec_real, const float* spec_fake, int64_t bins_per_sample, int B, float eps,
                              double* out, wdg_stream stream) {
    WDG_CHECK_ARG(spec_real && spec_fake && out && B > 0 && bins_per_sample > 0, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    WDG_HIP(hipMemsetAsync(out, 0, sizeof(double) * B, st));
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((bins_per_sample + 2047) / 2048, 512));
    hipLaunchKernelGGL(wdg_lsd_reduce_kernel, dim3(bx, B), dim3(256), 0, st, spec_real, spec_fake, bins_per_sample, eps, out);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- spatially convolved KS statistic ---------------------------------------------------------------------------------
// metrics.py:155-187: for every (time, channel) image and every patch position (stride 1, VALID), the KS statistic of
// the patch's values in the real vs the generated field, the sup taken over the 100 points linspace(-30, 30, 100) of
// |ECDF_real(p) - ECDF_fake(p)| with ECDF(p) = mean(sample <= p); the result is the mean over (time x channel) and batch,
// an image of patch positions.
// Pass 1 maps every value to its bin k = #{points < x} (0..100; NaN -> 100: it is <= no point).  ECDF(p_k) then is
// #{bin <= k} / n, so per patch position one signed histogram (real - fake) and a running sum give the statistic.
#define WDG_KS_POINTS 100
__global__ void __launch_bounds__(256) wdg_ks_bins_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ pts,
                                                          unsigned char* bins) {
    __shared__ float p[WDG_KS_POINTS];
    if (threadIdx.x < WDG_KS_POINTS) p[threadIdx.x] = pts[threadIdx.x];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        int lo = 0, hi = WDG_KS_POINTS;          // smallest k with v <= p[k]; WDG_KS_POINTS if none (also NaN)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (v <= p[mid]) hi = mid; else lo = mid + 1;
        }
        bins[i] = (unsigned char)lo;
    }
}

// one block = 16 x 16 patch positions of one (image, time, channel); bins are [B][T][H][W][C] bytes
__global__ void __launch_bounds__(256) wdg_ks_patch_kernel(const unsigned char* __restrict__ br, const unsigned char* __restrict__ bf,
                                                           int T, int H, int W, int C, int ps, double scale, double* out) {
    extern __shared__ short lds[];
    const int tw = 16 + ps - 1;
    short* hist = lds;                                            // [101][256]
    unsigned char* tr = reinterpret_cast<unsigned char*>(lds + (WDG_KS_POINTS + 1) * 256);   // [tw][tw]
    unsigned char* tf = tr + tw * tw;
    const int Hp = H - ps + 1, Wp = W - ps + 1;
    const int img = blockIdx.z;                                   // (b*T + t)*C + c
    const int c = img % C, bt = img / C;
    const int y0 = blockIdx.y * 16, x0 = blockIdx.x * 16;
    const unsigned char* sr = br + (int64_t)bt * H * W * C + c;
    const unsigned char* sf = bf + (int64_t)bt * H * W * C + c;
    for (int i = threadIdx.x; i < tw * tw; i += 256) {
        const int yy = y0 + i / tw, xx = x0 + i % tw;
        const bool ok = yy < H && xx < W;
        tr[i] = ok ? sr[((int64_t)yy * W + xx) * C] : (unsigned char)WDG_KS_POINTS;
        tf[i] = ok ? sf[((int64_t)yy * W + xx) * C] : (unsigned char)WDG_KS_POINTS;
    }
    for (int k = 0; k <= WDG_KS_POINTS; ++k) hist[k * 256 + threadIdx.x] = 0;
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int py = y0 + ly, px = x0 + lx;
    if (py < Hp && px < Wp) {
        for (int a = 0; a < ps; ++a)
            for (int b = 0; b < ps; ++b) {
                hist[tr[(ly + a) * tw + lx + b] * 256 + threadIdx.x] += 1;
                hist[tf[(ly + a) * tw + lx + b] * 256 + threadIdx.x] -= 1;
            }
        int cum = 0, m = 0;
        for (int k = 0; k < WDG_KS_POINTS; ++k) {
            cum += hist[k * 256 + threadIdx.x];
            m = max(m, abs(cum));
        }
        atomicAdd(&out[(int64_t)py * Wp + px], (double)m * scale);
    }
}

extern "C" size_t wdg_spatial_ks_scratch_bytes(int B, int T, int H, int W, int C) {
    return (size_t)2 * B * T * H * W * C;
}

extern "C" int wdg_spatial_ks(const float* real, const float* fake, int B, int T, int H, int W, int C, int patch,
                              const float* points100, void* scratch, double* out, wdg_stream stream) {
    WDG_CHECK_ARG(real && fake && points100 && scratch && out, "null argument");
    WDG_CHECK_ARG(patch >= 1 && patch <= H && patch <= W && patch <= 48, "patch size out of range (1..min(H, W, 48))");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)B * T * H * W * C;
    unsigned char* br = (unsigned char*)scratch;
    unsigned char* bf = br + n;
    const int qb = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(wdg_ks_bins_kernel, dim3(qb), dim3(256), 0, st, real, n, points100, br);
    hipLaunchKernelGGL(wdg_ks_bins_kernel, dim3(qb), dim3(256), 0, st, fake, n, points100, bf);
    WDG_LAUNCH_CHECK();
    const int Hp = H - patch + 1, Wp = W - patch + 1;
    WDG_HIP(hipMemsetAsync(out, 0, sizeof(double) * Hp * Wp, st));
    const int tw = 16 + patch - 1;
    const size_t lds = (size_t)(WDG_KS_POINTS + 1) * 256 * sizeof(short) + 2 * (size_t)tw * tw;
    static bool attr_set = false;
    if (!attr_set) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_ks_patch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_set = true;
    }
    const double scale = 1.0 / ((double)patch * patch) / ((double)B * T * C);
    hipLaunchKernelGGL(wdg_ks_patch_kernel, dim3((Wp + 15) / 16, (Hp + 15) / 16, B * T * C), dim3(256), lds, st, br, bf, T, H, W, C,
                       patch, scale, out);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
