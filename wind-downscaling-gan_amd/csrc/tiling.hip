// tiling.hip — the tile bookkeeping of the inference driver (/root/reference/src/downscaling/api.py:96-151, `predict`) on the
// device: (1) gather the latitude-flipped 96 x 96 x 24 h tiles out of the field and, in the same pass, the NaN-aware
// sums the normalisation needs (api.py:117-129: np.nanmean / np.nanstd over axes (0, 1, 2) of the stacked tiles, i.e. one
// statistic per (column inside the tile, channel)); (2) normalise in place; (3) add a group's 2-pixel-cropped predictions
// and counts into the output grid (api.py:139-150: the uniform mean over the tiles covering a pixel).  The reference does
// these with numpy / pandas on the host; a chain of torch slice operations does them in ~20 ms per 1200 x 1200 x 24 h field;
// these three kernels are HBM passes over the tiles.
#include "common.h"
#include <algorithm>

// keys[n] = {sx, row0, k, 0}: tile n covers columns sx .. sx+S-1, rows row0, row0-1, ... (latitude flipped) and timesteps
// k*T .. k*T+T-1.  field [Ttot][LAT][LON][C]; tiles [N][T][S][S][C].  stats [S*C][3] fp64: sum, sum of squares, count of the
// non-NaN elements of column j, channel c over all tiles, timesteps and rows.
// Every block keeps its sums in registers and adds them to one of R replica rows of `stats` at its end (atomics on S*C
// addresses would serialise).  The variance is the TWO-PASS centred form np.nanstd uses — mean first, then the sum of squared
// deviations from it (a one-pass sumsq / n - mean^2 cancels on a near-constant column and clamps a tiny positive variance to
// zero, which turns a finite normalised value into inf / NaN): finish1 folds the replicas into the fp64 mean and the count
// (kept in replica row 0, sum-of-squares slots of every row zeroed), the centred kernel re-reads the gathered tiles and
// accumulates (v - mean)^2 into the zeroed slots, finish2 folds them: std = sqrt(sum / n) (population variance), fp32.
__global__ void __launch_bounds__(256) wdg_tiles_stats_finish1_kernel(double* stats, int R, int SC, float* mean_std) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SC) return;
    double s = 0, c = 0;
    for (int r = 0; r < R; ++r) {
        double* p = stats + ((long long)r * SC + e) * 3;
        s += p[0]; c += p[2];
        p[1] = 0.0;
    }
    const double m = s / c;
    stats[(long long)e * 3] = m;
    stats[(long long)e * 3 + 2] = c;
    mean_std[2 * e] = (float)m;
}

__global__ void __launch_bounds__(256) wdg_tiles_centred_kernel(const float* __restrict__ tiles, double* stats, int R, int SC,
                                                                long long lines) {
    double q0[2] = {0, 0}, m0[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < SC) m0[u] = stats[(long long)e * 3];
    }
    for (long long line = blockIdx.x; line < lines; line += gridDim.x) {
        const float* src = tiles + line * SC;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = threadIdx.x + 256 * u;
            if (e < SC) {
                const float v = src[e];
                if (v == v) { const double d = (double)v - m0[u]; q0[u] += d * d; }
            }
        }
    }
    double* row = stats + (long long)(blockIdx.x % R) * SC * 3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < SC) atomicAdd(&row[3 * e + 1], q0[u]);
    }
}

__global__ void __launch_bounds__(256) wdg_tiles_stats_finish2_kernel(const double* __restrict__ stats, int R, int SC,
                                                                      float* mean_std) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SC) return;
    double q = 0;
    for (int r = 0; r < R; ++r) q += stats[((long long)r * SC + e) * 3 + 1];
    mean_std[2 * e + 1] = (float)sqrt(q / stats[(long long)e * 3 + 2]);
}

__global__ void __launch_bounds__(256) wdg_tiles_gather_rep_kernel(const float* __restrict__ field, int LAT, int LON, int C,
                                                                   const int4* __restrict__ keys, int N, int T, int S,
                                                                   float* __restrict__ tiles, double* stats, int R, long long lines) {
    // R persistent-ish blocks per replica: block b walks lines b, b + gridDim.x, ...; its sums stay in registers (every thread owns
    // up to ceil(S*C / 256) fixed (column, channel) slots) and reach its replica row once at the end
    const int SC = S * C;
    double s0[2] = {0, 0}, q0[2] = {0, 0}, c0[2] = {0, 0};   // S*C <= 512
    for (long long line = blockIdx.x; line < lines; line += gridDim.x) {
        const int i = (int)(line % S);
        const long long r = line / S;
        const int t = (int)(r % T);
        const int n = (int)(r / T);
        const int4 key = keys[n];
        const float* src = field + (((long long)(key.z * T + t) * LAT + (key.y - i)) * LON + key.x) * C;
        float* dst = tiles + line * SC;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = threadIdx.x + 256 * u;
            if (e < SC) {
                const float v = src[e];
                dst[e] = v;
                if (v == v) { s0[u] += (double)v; q0[u] += (double)v * (double)v; c0[u] += 1.0; }
            }
        }
    }
    double* row = stats + (long long)(blockIdx.x % R) * SC * 3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < SC) {
            atomicAdd(&row[3 * e], s0[u]);
            atomicAdd(&row[3 * e + 1], q0[u]);
            atomicAdd(&row[3 * e + 2], c0[u]);
        }
    }
}

// tiles[...] = (tiles[...] - mean[j, c]) / std[j, c]   (fp32, as the driver's broadcast expression)
__global__ void __launch_bounds__(256) wdg_tiles_normalise_kernel(float* tiles, const float* __restrict__ mean_std, int SC,
                                                                  long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int e = (int)(idx % SC);
        tiles[idx] = (tiles[idx] - mean_std[2 * e]) / mean_std[2 * e + 1];
    }
}

extern "C" int wdg_tiles_gather_normalise(const float* field, int LAT, int LON, int C, const int32_t* keys4, int N, int T, int S,
                                          float* tiles, double* stats_scratch, int replicas, float* mean_std, wdg_stream stream) {
    WDG_CHECK_ARG(field && keys4 && tiles && stats_scratch && mean_std && N > 0 && T > 0 && S > 0 && C > 0, "bad argument");
    WDG_CHECK_ARG(S * C <= 512 && replicas >= 1, "tile rows of more than 512 elements are not supported");
    hipStream_t st = (hipStream_t)stream;
    const int SC = S * C;
    WDG_HIP(hipMemsetAsync(stats_scratch, 0, (size_t)replicas * SC * 3 * sizeof(double), st));
    const long long lines = (long long)N * T * S;
    const int blocks = (int)std::min<long long>(lines, 4096);
    hipLaunchKernelGGL(wdg_tiles_gather_rep_kernel, dim3(blocks), dim3(256), 0, st, field, LAT, LON, C,
                       reinterpret_cast<const int4*>(keys4), N, T, S, tiles, stats_scratch, replicas, lines);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_tiles_stats_finish1_kernel, dim3((SC + 255) / 256), dim3(256), 0, st, stats_scratch, replicas, SC, mean_std);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_tiles_centred_kernel, dim3(blocks), dim3(256), 0, st, tiles, stats_scratch, replicas, SC, lines);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_tiles_stats_finish2_kernel, dim3((SC + 255) / 256), dim3(256), 0, st, stats_scratch, replicas, SC, mean_std);
    WDG_LAUNCH_CHECK();
    const long long total = lines * SC;
    hipLaunchKernelGGL(wdg_tiles_normalise_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0, st,
                       tiles, mean_std, SC, total);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// pred [B][T][S][S][ldp] (first 2 channels used); the first n_real tiles of the group are added, cropped by `crop` pixels
// on every side, into acc [NT][LAT][LON][2] (fp64) and cnt [NT][LAT][LON] (int32).  Tiles of one group overlap: atomics.
__global__ void __launch_bounds__(256) wdg_tiles_blend_kernel(const float* __restrict__ pred, int ldp, const int4* __restrict__ keys,
                                                              int n_real, int T, int S, int crop, int LAT, int LON, double* acc,
                                                              int* cnt) {
    const int inner = S - 2 * crop;
    const long long total = (long long)n_real * T * inner * inner;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int jj = (int)(idx % inner) + crop;
        long long r = idx / inner;
        const int i = (int)(r % inner) + crop;
        r /= inner;
        const int t = (int)(r % T);
        const int n = (int)(r / T);
        const int4 key = keys[n];
        const float* p = pred + ((((long long)n * T + t) * S + i) * S + jj) * ldp;
        const long long o = ((long long)(key.z * T + t) * LAT + (key.y - i)) * LON + key.x + jj;
        atomicAdd(&acc[2 * o], (double)p[0]);
        atomicAdd(&acc[2 * o + 1], (double)p[1]);
        atomicAdd(&cnt[o], 1);
    }
}

extern "C" int wdg_tiles_blend(const float* pred, int ldp, const int32_t* keys4, int n_real, int T, int S, int crop, int LAT, int LON,
                               double* acc, int32_t* cnt, wdg_stream stream) {
    WDG_CHECK_ARG(pred && keys4 && acc && cnt && n_real >= 0 && ldp >= 2 && S > 2 * crop, "bad argument");
    if (n_real == 0) return WDG_OK;
    const long long total = (long long)n_real * T * (S - 2 * crop) * (S - 2 * crop);
    hipLaunchKernelGGL(wdg_tiles_blend_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, pred, ldp, reinterpret_cast<const int4*>(keys4), n_real, T, S, crop, LAT, LON, acc, cnt);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
