// upconv4.hip — UpSampling2D(2, 'bilinear') + Conv2DTranspose(5x5, stride 1, 'same') as FOUR 4x4 convolutions
// on the low-resolution grid (/root/reference/src/downscaling/gan/models.py:62-64, the generator's largest
// layer: 30.8 % of its MACs).
//
// Bilinear x2 with half-pixel centres is linear with fixed taps (0.25 / 0.75 on two neighbouring low-res
// samples), so the 5 high-res taps that an output row 2i+ph sees collapse onto 4 low-res rows
// (i-2+ph .. i+1+ph); the same holds for columns.  For each output phase (ph, pw) the layer therefore IS a
// 4x4 convolution of the low-res tensor with the composite kernel
//     Wc[ph,pw][dh][dw] = sum_{a,b} beta_ph(a, dh) * beta_pw(b, dw) * W[a][b]        (a, b = 5x5 taps)
// — 16 instead of 25 taps per output pixel (-36 % MACs), no interpolation arithmetic in the staging, and the
// four phases of a low-res tile share ONE staged halo.  Exactness at the image border: the interpolation's
// edge clamp is reproduced by clamping the low-res read coordinates; the 5x5 layer's zero padding of the
// UPSAMPLED image (high-res taps that fall outside it contribute nothing) changes the composite kernel only
// for the first / last low-res row and column, so each of the 3x3 (row class, column class) combinations has
// its own composite kernel: the interior one serves the main kernel, the other eight a small gather kernel
// over the one-pixel border ring.  Re-association changes rounding at the 1e-7 level only.
#include "common.h"
#include <algorithm>

// composite weights: [variant = rc*3+cc][phase = ph*2+pw][tap = th*4+tw][o (16, zero padded)][Cp]
//   rc / cc: 0 = first row / column, 1 = interior, 2 = last;   low-res tap dh = th + ph - 2 (dw likewise)
struct WdgUp4 {
    const float* X;      // low-res [n_img][H][W][ldA]
    const float* Wc;     // composite weights
    const float* bias;   // [N] or null
    float* Out;          // high-res [n_img][2H][2W][ldO]
    long long imgStrideA, imgStrideO;
    int n_img, H, W, ldA, ldO;
    int C4, Cp, N;       // channel groups of 4, padded channels, outputs (<= 16)
    int act;
    float slope;
    int tiles_h, tiles_w;
    int nW, nH;          // border segments of 16 pixels along a row / a column
};

// weight of low-res sample i+d in upsampled row 2i+ph+a; rc selects the zero-padding drops of the first / last row
__device__ __forceinline__ float up4_coef(int ph, int rc, int a, int d) {
    const int j = ph + a;                       // upsampled row relative to 2i
    if ((rc == 0 && j < 0) || (rc == 2 && j >= 2)) return 0.f;
    const int m = j >= 0 ? j >> 1 : -((1 - j) >> 1);   // floor(j / 2)
    if ((j - 2 * m) == 0) return d == m - 1 ? 0.25f : d == m ? 0.75f : 0.f;
    return d == m ? 0.75f : d == m + 1 ? 0.25f : 0.f;
}

// w: the transposed layer's kernel as stored (HWIO [5][5][N][C]: I = layer outputs, O = layer inputs).
// y[oh][ow][o] = sum_{th,tw} U[oh + 2 - th][ow + 2 - tw][c] * w[th][tw][o][c]  ->  a = 2 - th, b = 2 - tw.
__global__ void __launch_bounds__(256) wdg_upconv4_pack_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                               int N, int C, int Cp) {
    const long long total = 9LL * 4 * 16 * 16 * Cp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % Cp);
        long long r = idx / Cp;
        const int o = (int)(r % 16); r /= 16;
        const int tap = (int)(r % 16); r /= 16;
        const int phase = (int)(r % 4);
        const int v = (int)(r / 4);
        float s = 0.f;
        if (o < N && c < C) {
            const int rc = v / 3, cc = v % 3, ph = phase >> 1, pw = phase & 1;
            const int dh = (tap >> 2) + ph - 2, dw = (tap & 3) + pw - 2;
            for (int a = -2; a <= 2; ++a) {
                const float ch = up4_coef(ph, rc, a, dh);
                if (ch == 0.f) continue;
                for (int b = -2; b <= 2; ++b) {
                    const float cw = up4_coef(pw, cc, b, dw);
                    if (cw != 0.f) s += ch * cw * w[(((2 - a) * 5 + (2 - b)) * N + o) * (long long)C + c];
                }
            }
        }
        out[idx] = s;
    }
}

// ---- main kernel: interior low-res pixels [1, H-2] x [1, W-2]; block = 8 x 16 low-res tile, wave = output phase ----
constexpr int U4_TH = 8, U4_TW = 16, U4_HH = U4_TH + 4, U4_HW = U4_TW + 4, U4_NPIX = U4_HH * U4_HW;   // 12 x 20 = 240
static_assert(U4_NPIX % 16 == 0, "halo pixel count must be a multiple of 16 (conflict-free fragment reads)");

__global__ void __launch_bounds__(256) wdg_upconv4_kernel(const WdgUp4 p) {
    __shared__ __attribute__((aligned(16))) f32x4 lds[2][4 * U4_NPIX];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ph = wave >> 1, pw = wave & 1;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h;
    const int img = bid / p.tiles_h;
    const int i0 = 1 + ty * U4_TH, j0 = 1 + tx * U4_TW;
    const float* Ximg = p.X + (long long)img * p.imgStrideA;

    // staging slots of this thread: NL float4 per 16-channel chunk, coordinates clamped (= the bilinear edge clamp)
    constexpr int NL = (4 * U4_NPIX + 255) / 256;
    int soff[NL];     // global element offset of the slot's pixel
    int sdst[NL];     // LDS slot, or -1
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = t + 256 * i;
        const int kg = idx / U4_NPIX, pix = idx - kg * U4_NPIX;
        const int hy = pix / U4_HW, hx = pix - hy * U4_HW;
        const int gy = min(max(i0 - 2 + hy, 0), p.H - 1), gx = min(max(j0 - 2 + hx, 0), p.W - 1);
        soff[i] = (gy * p.W + gx) * p.ldA + 4 * kg;
        sdst[i] = idx < 4 * U4_NPIX ? idx : -1;
    }
    f32x4 rs[NL];
    auto load_chunk = [&](int ck) {
        const int kgs = min(4, p.C4 - 4 * ck);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int kg = (t + 256 * i) / U4_NPIX;
            rs[i] = (sdst[i] >= 0 && kg < kgs) ? *reinterpret_cast<const f32x4*>(Ximg + soff[i] + 16 * ck)
                                                : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x4 acc[U4_TH];
#pragma unroll
    for (int r = 0; r < U4_TH; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // interior composite kernel (variant 4), this wave's phase: [tap][o = li][Cp]
    const float* Wph = p.Wc + ((long long)(4 * 4 + wave) * 16 * 16 + li) * p.Cp + 4 * lg;
    const int abase = lg * U4_NPIX + li + (ph * U4_HW + pw);   // + (th*HW + tw) + r*HW per tap / row

    const int nchunk = (p.C4 + 3) >> 2;
    // Weight fragments come straight from global memory (L1/L2 resident), four taps per batch, double buffered.  Order
    // matters for the in-order vmcnt queue: the first batch of a chunk is issued BEFORE the next chunk's halo prefetch,
    // so waiting for it never waits for the prefetch; later batches are issued behind the prefetch, which has landed by
    // the time they are needed (4 taps = 128 MFMAs later).
    f32x4 bw[2][4];
    auto load_w = [&](int ck, int g, f32x4 (&dst)[4]) {
        const bool kvalid = lg < min(4, p.C4 - 4 * ck);
        const float* Wk = Wph + 16 * ck + (long long)(4 * g) * 16 * p.Cp;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            dst[u] = kvalid ? *reinterpret_cast<const f32x4*>(Wk + (long long)u * 16 * p.Cp) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    load_chunk(0);
    for (int ck = 0; ck < nchunk; ++ck) {
        f32x4* st = lds[ck & 1];
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (sdst[i] >= 0) st[sdst[i]] = rs[i];
        load_w(ck, 0, bw[0]);
        __syncthreads();   // one barrier per chunk: the other stage is only rewritten after the next barrier
        if (ck + 1 < nchunk) load_chunk(ck + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) load_w(ck, g + 1, bw[(g + 1) & 1]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tap = 4 * g + u;
                const int ao = abase + (tap >> 2) * U4_HW + (tap & 3);
                f32x4 af[U4_TH];
#pragma unroll
                for (int r = 0; r < U4_TH; ++r) af[r] = st[ao + r * U4_HW];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < U4_TH; ++r)
                        // A = weights, B = pixels: transposed accumulator, reg q of lane (li, lg) = output channel 4*lg + q of
                        // low-res column j0 + li -> one 16-byte store per row instead of four 4-byte ones
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[g & 1][u][j], af[r][j], acc[r], 0, 0, 0);
            }
        }
    }

    const int j = j0 + li;
    if (4 * lg < ((p.N + 3) & ~3) && j <= p.W - 2) {
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = 4 * lg + q < p.N ? p.bias[4 * lg + q] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < U4_TH; ++r) {
            const int i = i0 + r;
            if (i > p.H - 2) continue;
            f32x4 v = acc[r] + bv;
            if (p.act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = wdg_lrelu(v[q], p.slope);
            }
            *reinterpret_cast<f32x4*>(p.Out + (long long)img * p.imgStrideO +
                                      ((long long)(2 * i + ph) * (2 * p.W) + (2 * j + pw)) * p.ldO + 4 * lg) = v;
        }
    }
}

// ---- border ring: first / last low-res row and column.  Block = one 16-pixel segment of uniform (row class, column
// class); wave = output phase; operands gathered straight from global memory with clamped coordinates. ----------
__global__ void __launch_bounds__(256) wdg_upconv4_border_kernel(const WdgUp4 p) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int ph = wave >> 1, pw = wave & 1;
    const int per_img = 2 * (p.nW + 2) + 2 * p.nH;
    const int img = blockIdx.x / per_img;
    int s = blockIdx.x - img * per_img;
    // segment -> (start, direction, count, classes)
    int si, sj, di = 0, dj = 0, cnt, rc, cc;
    if (s < 2 * (p.nW + 2)) {           // first / last row
        const int bottom = s >= p.nW + 2;
        if (bottom) s -= p.nW + 2;
        si = bottom ? p.H - 1 : 0;
        rc = bottom ? 2 : 0;
        if (s == 0) { sj = 0; cnt = 1; cc = 0; }
        else if (s == p.nW + 1) { sj = p.W - 1; cnt = 1; cc = 2; }
        else { sj = 1 + 16 * (s - 1); cnt = min(16, p.W - 1 - sj); cc = 1; dj = 1; }
    } else {                            // first / last column, rows 1 .. H-2
        s -= 2 * (p.nW + 2);
        const int right = s >= p.nH;
        if (right) s -= p.nH;
        sj = right ? p.W - 1 : 0;
        cc = right ? 2 : 0;
        si = 1 + 16 * s; cnt = min(16, p.H - 1 - si); rc = 1; di = 1;
    }
    if (cnt <= 0) return;
    const int k = min(li, cnt - 1);     // lanes past the segment end recompute its last pixel (not stored)
    const int i = si + k * di, j = sj + k * dj;
    const float* Ximg = p.X + (long long)img * p.imgStrideA;
    const float* Wv = p.Wc + ((long long)((rc * 3 + cc) * 4 + wave) * 16 * 16 + li) * p.Cp;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ngrp = (p.C4 + 3) >> 2;   // 16-channel chunks
    // four independent accumulator chains and batches of four chunks per tap: 8 loads in flight per lane
    f32x4 acc4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc4[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < 16; ++tap) {
        const int gy = min(max(i + (tap >> 2) + ph - 2, 0), p.H - 1), gx = min(max(j + (tap & 3) + pw - 2, 0), p.W - 1);
        const float* xp = Ximg + ((long long)gy * p.W + gx) * p.ldA;
        const float* wp = Wv + (long long)tap * 16 * p.Cp;
        for (int ck = 0; ck < ngrp; ck += 4) {
            f32x4 af[4], bf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = 4 * (ck + u) + lg;
                af[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                bf[u] = af[u];
                if (g < p.C4) {
                    af[u] = *reinterpret_cast<const f32x4*>(xp + 4 * g);
                    bf[u] = *reinterpret_cast<const f32x4*>(wp + 4 * g);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc4[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[u][q], af[u][q], acc4[u], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (acc4[0][q] + acc4[1][q]) + (acc4[2][q] + acc4[3][q]);
    // transposed accumulator: reg q of lane (li, lg) = output channel 4*lg + q of segment pixel li
    if (4 * lg < ((p.N + 3) & ~3) && li < cnt) {
        f32x4 v = acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] += (p.bias && 4 * lg + q < p.N) ? p.bias[4 * lg + q] : 0.f;
            if (p.act) v[q] = wdg_lrelu(v[q], p.slope);
        }
        const int oi = si + li * di, oj = sj + li * dj;
        *reinterpret_cast<f32x4*>(p.Out + (long long)img * p.imgStrideO +
                                  ((long long)(2 * oi + ph) * (2 * p.W) + (2 * oj + pw)) * p.ldO + 4 * lg) = v;
    }
}

// ---- host -------------------------------------------------------------------------------------------------------------
extern "C" size_t wdg_upconv4_weight_floats(int N, int C) {
    (void)N;
    return (size_t)9 * 4 * 16 * 16 * (size_t)wdg_round_up(C, 4);
}

extern "C" int wdg_upconv4_supported(int N, int C, int H, int W) { return N >= 1 && N <= 16 && C >= 4 && H >= 3 && W >= 3; }

extern "C" int wdg_upconv4_pack(const float* w_hwio, int N, int C, float* wc, wdg_stream stream) {
    WDG_CHECK_ARG(w_hwio && wc && N >= 1 && N <= 16 && C >= 1, "bad argument");
    const int Cp = wdg_round_up(C, 4);
    const long long total = 9LL * 4 * 16 * 16 * Cp;
    hipLaunchKernelGGL(wdg_upconv4_pack_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, w_hwio, wc, N, C, Cp);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_upconv4_fwd(const float* x_low, int ld_low, int64_t img_stride_low, int n_img, int H, int W, int C,
                               const float* wc, const float* bias, float* y, int ldy, int64_t img_stride_y, int N,
                               int act, float slope, wdg_stream stream) {
    WDG_CHECK_ARG(x_low && wc && y, "null argument");
    WDG_CHECK_ARG(wdg_upconv4_supported(N, C, H, W), "unsupported shape");
    WDG_CHECK_ARG(((uintptr_t)x_low & 15) == 0 && ld_low % 4 == 0 && ld_low >= wdg_round_up(C, 4), "x_low alignment / ld");
    WDG_CHECK_ARG(((uintptr_t)y & 15) == 0 && ldy % 4 == 0 && ldy >= wdg_round_up(N, 4), "y alignment / ld");
    WDG_CHECK_ARG((long long)H * W * ld_low < (1LL << 31), "low-res image too large for 32-bit offsets");
    WdgUp4 p;
    memset(&p, 0, sizeof(p));
    p.X = x_low; p.Wc = wc; p.bias = bias; p.Out = y;
    p.imgStrideA = img_stride_low; p.imgStrideO = img_stride_y;
    p.n_img = n_img; p.H = H; p.W = W; p.ldA = ld_low; p.ldO = ldy;
    p.Cp = wdg_round_up(C, 4); p.C4 = p.Cp / 4; p.N = N;
    p.act = act; p.slope = slope;
    p.tiles_h = (H - 2 + U4_TH - 1) / U4_TH;
    p.tiles_w = (W - 2 + U4_TW - 1) / U4_TW;
    p.nW = (W - 2 + 15) / 16;
    p.nH = (H - 2 + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wdg_upconv4_kernel, dim3((unsigned)((long long)n_img * p.tiles_h * p.tiles_w)), dim3(256), 0, st, p);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_upconv4_border_kernel, dim3((unsigned)((long long)n_img * (2 * (p.nW + 2) + 2 * p.nH))), dim3(256),
                       0, st, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
