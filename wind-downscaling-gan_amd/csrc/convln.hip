// convln.hip — thin 3x3 'same' conv + bias + LeakyReLU + LayerNormalization, fused, for the discriminator's
// full-resolution branches (/root/reference/src/downscaling/gan/models.py:94-97 and :102-105: SN-Conv2D
// (2->16 / 16->16) with LeakyReLU(0.2) followed by LayerNormalization over the 16 channels).
//
// These layers are HBM/latency bound (K = 18 / 144, N = 16).  One thread owns one pixel and all 16 output
// channels, so the LayerNorm statistics are thread-local; weights are wave-uniform scalar loads; there is no
// MFMA padding.
//   forward : x -> y = lrelu(conv(x) + b) (kept for the backward), z = LN(y)*gamma + beta, (mean, rstd)
//   backward: dz, y, (mean, rstd) -> dpre = LN'(dz) * lrelu'(y) on an 8x32 tile + 1-pixel halo (recomputed),
//             dx = conv^T(dpre) from the LDS tile, dpre written once for the weight-gradient kernel,
//             dgamma / dbeta / dbias accumulated (block reduction + fp32 atomics, as wdg_ln_bwd does).
// Same fp32 operations as wdg_conv_fwd + wdg_ln_fwd / wdg_ln_bwd + wdg_conv_dgrad, different summation order.
#include "common.h"
#include <algorithm>

constexpr int CN_TH = 8, CN_TW = 32;
constexpr int CN_CO = 16;

struct WdgConvLn {
    const float* X;   // [n,H,W,ldx]
    float* Y;         // [n,H,W,ldy]  (forward out / backward in)
    float* Z;         // forward out view (ldz)
    float* MR;        // [P][2] mean, rstd
    const float* dZ;  // backward in (lddz)
    float* dPre;      // backward out, dense [P][16]
    float* dWpart;    // recomputing backward with fused weight gradient: per-block partials [gridDim.x][RT*16][16]
    float* dX;        // backward out view (lddx), optional
    float* dgamma;    // accumulate, optional
    float* dbeta;
    float* dbias;
    long long isx, isy, isz, isdz, isdx;
    int n_img, H, W, ldx, ldy, ldz, lddz, lddx;
    float eps, slope;
    int tiles_h, tiles_w;
};

template <int CIN>
__global__ void __launch_bounds__(256) wdg_convln_fwd_kernel(const WdgConvLn p, const float* __restrict__ Wt,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta) {
    constexpr int C4 = (CIN + 3) / 4;
    // outputs leave through LDS: a thread owns the 16 channels of one pixel, but a wave's 16-byte stores then touch 64
    // different 128-byte lines each (the pixel stride of z inside the concatenation) — transposed, four consecutive lanes write
    // the 64 contiguous bytes of a pixel and a store instruction covers 16 pixels
    __shared__ float tz[256 * (CN_CO + 1)];
    const long long P = (long long)p.n_img * p.H * p.W;
    const long long pix0 = (long long)blockIdx.x * 256;
    const long long pix = pix0 + threadIdx.x;
    const bool live = pix < P;
    const int img = live ? (int)(pix / ((long long)p.H * p.W)) : 0;
    const int rem = live ? (int)(pix - (long long)img * p.H * p.W) : 0;
    const int oy = rem / p.W, ox = rem - oy * p.W;
    float acc[CN_CO];
#pragma unroll
    for (int o = 0; o < CN_CO; ++o) acc[o] = bias[o];
    // every tap's input first: one exposed load latency instead of nine.  Branch-free buffer loads (padding: bit 31 of the
    // offset set arithmetically -> beyond the descriptor -> zeros): written as `inside ? load : 0` the nine loads compiled to
    // nine exec-masked blocks, each with its own s_waitcnt vmcnt(0) — nine round trips in sequence after all
    // (the descriptor must be wave-uniform — a per-lane one is served by a readfirstlane loop: it starts at the image of the
    // workgroup's first pixel, the offsets carry the (at most few) images a workgroup's 256 pixels run on from there)
    f32x4 xv[9][C4];
    const int img0 = (int)(pix0 / ((long long)p.H * p.W));
    const wdg_srd srdX = wdg_make_srd(p.X + (long long)img0 * p.isx);
    const int dead = live ? 0 : -1;
    const int ibase = (img - img0) * (int)p.isx;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int gy = oy + tap / 3 - 1, gx = ox + tap % 3 - 1;
        const unsigned neg = (unsigned)((dead | gy | (p.H - 1 - gy) | gx | (p.W - 1 - gx)) >> 31);      // all ones outside
        const unsigned off = ((unsigned)((ibase + (gy * p.W + gx) * p.ldx) * 4) & ~neg) | (neg & 0x80000000u);
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) xv[tap][c4] = wdg_buffer_load_f32x4(srdX, off + 16 * c4);
    }
    // all nine requested before the first is consumed (the scheduler pairs them with their uses), and every destination kept
    // whole until here: with two channels in use the allocator recycled the dead half of a 16-byte destination for the next
    // tap's address arithmetic — and had to wait for the load in between
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) asm volatile("" : "+v"(xv[tap][c4]));
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * c4 + j;
                if (c < CIN) {
                    const float* w = Wt + (tap * CIN + c) * CN_CO;   // HWIO, wave-uniform -> scalar loads
#pragma unroll
                    for (int o = 0; o < CN_CO; ++o) acc[o] = fmaf(xv[tap][c4][j], w[o], acc[o]);
                }
            }
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < CN_CO; ++o) {
        acc[o] = acc[o] > 0.f ? acc[o] : acc[o] * p.slope;
        s += acc[o];
    }
    const float mean = s * (1.f / CN_CO);
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < CN_CO; ++o) {
        const float d = acc[o] - mean;
        q += d * d;
    }
    const float rstd = 1.f / sqrtf(q * (1.f / CN_CO) + p.eps);
    if (live && p.MR) {
        p.MR[2 * pix] = mean;
        p.MR[2 * pix + 1] = rstd;
    }
    static_assert(CN_CO == 16, "the store phase moves 4 float4 per pixel");
    for (int pass = 0; pass < (p.Y ? 2 : 1); ++pass) {          // z, then (when kept) y through the same LDS tile
        if (pass) __syncthreads();
#pragma unroll
        for (int o = 0; o < CN_CO; ++o)
            tz[threadIdx.x * (CN_CO + 1) + o] = pass ? acc[o] : (acc[o] - mean) * rstd * gamma[o] + beta[o];
        __syncthreads();
        float* base = pass ? p.Y : p.Z;
        const int ld = pass ? p.ldy : p.ldz;
        const long long is = pass ? p.isy : p.isz;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = (threadIdx.x >> 2) + 64 * i, o4 = threadIdx.x & 3;
            const long long q_ = pix0 + pl;
            if (q_ < P) {
                const int im = (int)(q_ / ((long long)p.H * p.W));
                const long long r_ = q_ - (long long)im * p.H * p.W;
                const float* src = &tz[pl * (CN_CO + 1) + 4 * o4];
                *reinterpret_cast<f32x4*>(base + (long long)im * is + r_ * ld + 4 * o4) = (f32x4){src[0], src[1], src[2], src[3]};
            }
        }
    }
}

template <int CIN>
__global__ void __launch_bounds__(256) wdg_convln_bwd_kernel(const WdgConvLn p, const float* __restrict__ Wt,
                                                             const float* __restrict__ gamma) {
    constexpr int GH = CN_TH + 2, GW = CN_TW + 2;
    __shared__ __attribute__((aligned(16))) float dps[GH * GW * CN_CO];   // dpre on the halo
    __shared__ float red[3 * CN_CO * 4];                                  // per-wave partials of dgamma/dbeta/dbias
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int b = blockIdx.x;
    const int tx = b % p.tiles_w;
    b /= p.tiles_w;
    const int ty = b % p.tiles_h;
    const int img = b / p.tiles_h;
    const int oy0 = ty * CN_TH, ox0 = tx * CN_TW;
    const float* dZimg = p.dZ + (long long)img * p.isdz;
    const float* Yimg = p.Y + (long long)img * p.isy;

    float ag[CN_CO], ab[CN_CO], abias[CN_CO];
#pragma unroll
    for (int o = 0; o < CN_CO; ++o) ag[o] = ab[o] = abias[o] = 0.f;

    // 1. dpre on the halo (LN backward + LeakyReLU'), zero outside the image
    for (int hp = t; hp < GH * GW; hp += 256) {
        const int hy = hp / GW, hx = hp - hy * GW;
        const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
        float d[CN_CO];
#pragma unroll
        for (int o = 0; o < CN_CO; ++o) d[o] = 0.f;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
            const long long pl = ((long long)img * p.H + gy) * p.W + gx;
            const float mean = p.MR[2 * pl], rstd = p.MR[2 * pl + 1];
            const float* dzp = dZimg + ((long long)gy * p.W + gx) * p.lddz;
            const float* yp = Yimg + ((long long)gy * p.W + gx) * p.ldy;
            float g[CN_CO], xh[CN_CO], yv[CN_CO];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int o4 = 0; o4 < CN_CO / 4; ++o4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(dzp + 4 * o4);
                const f32x4 y4 = *reinterpret_cast<const f32x4*>(yp + 4 * o4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = 4 * o4 + j;
                    yv[o] = y4[j];
                    xh[o] = (y4[j] - mean) * rstd;
                    g[o] = a[j];
                    const float gg = a[j] * gamma[o];
                    s1 += gg;
                    s2 += gg * xh[o];
                }
            }
            s1 *= (1.f / CN_CO);
            s2 *= (1.f / CN_CO);
            const bool centre = hy >= 1 && hy <= CN_TH && hx >= 1 && hx <= CN_TW;
#pragma unroll
            for (int o = 0; o < CN_CO; ++o) {
                float v = rstd * (g[o] * gamma[o] - s1 - xh[o] * s2);
                v *= (yv[o] > 0.f ? 1.f : p.slope);
                d[o] = v;
                if (centre) {                       // parameter gradients: every pixel counted once
                    ag[o] += g[o] * xh[o];
                    ab[o] += g[o];
                    abias[o] += v;
                }
            }
            if (centre) {
                float* dp = p.dPre + pl * CN_CO;
#pragma unroll
                for (int o4 = 0; o4 < CN_CO / 4; ++o4)
                    *reinterpret_cast<f32x4*>(dp + 4 * o4) = (f32x4){d[4 * o4], d[4 * o4 + 1], d[4 * o4 + 2], d[4 * o4 + 3]};
            }
        }
#pragma unroll
        for (int o4 = 0; o4 < CN_CO / 4; ++o4)
            // channel-group planes [o4][halo pixel]: consecutive lanes touch consecutive 16-byte slots (the pixel-major
            // layout was a 4-way bank conflict on every access: 37 % of the LDS cycles, profiles/r01ay_pmc_summary.csv)
            *reinterpret_cast<f32x4*>(&dps[(o4 * (GH * GW) + hp) * 4]) = (f32x4){d[4 * o4], d[4 * o4 + 1], d[4 * o4 + 2], d[4 * o4 + 3]};
    }
    // 2. parameter-gradient partials: wave reduction, then 4 waves through LDS, one atomic per value
    if (p.dgamma) {
#pragma unroll
        for (int o = 0; o < CN_CO; ++o) {
            const float a0 = wdg_wave_sum(ag[o]), a1 = wdg_wave_sum(ab[o]), a2 = wdg_wave_sum(abias[o]);
            if (lane == 0) {
                red[(0 * CN_CO + o) * 4 + wave] = a0;
                red[(1 * CN_CO + o) * 4 + wave] = a1;
                red[(2 * CN_CO + o) * 4 + wave] = a2;
            }
        }
    }
    __syncthreads();
    if (p.dgamma && t < 3 * CN_CO) {
        const float v = (red[t * 4] + red[t * 4 + 1]) + (red[t * 4 + 2] + red[t * 4 + 3]);
        float* dst = t < CN_CO ? p.dgamma + t : t < 2 * CN_CO ? p.dbeta + (t - CN_CO) : p.dbias + (t - 2 * CN_CO);
        atomicAdd(dst, v);
    }
    if (!p.dX) return;
    // 3. dx[c] = sum_tap sum_o dpre[pixel + (1 - th, 1 - tw)][o] * W[tap][c][o]
    const int py = t >> 5, px = t & 31;
    float dx[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) dx[c] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int th = tap / 3, tw = tap % 3;
        const float* dp = &dps[((py + 2 - th) * GW + px + 2 - tw) * 4];
        float v[CN_CO];
#pragma unroll
        for (int o4 = 0; o4 < CN_CO / 4; ++o4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(dp + o4 * (GH * GW) * 4);
            v[4 * o4] = a[0]; v[4 * o4 + 1] = a[1]; v[4 * o4 + 2] = a[2]; v[4 * o4 + 3] = a[3];
        }
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            const float* w = Wt + (tap * CIN + c) * CN_CO;
#pragma unroll
            for (int o = 0; o < CN_CO; ++o) dx[c] = fmaf(v[o], w[o], dx[c]);
        }
    }
    const int gy = oy0 + py, gx = ox0 + px;
    if (gy < p.H && gx < p.W) {
        float* dst = p.dX + (long long)img * p.isdx + ((long long)gy * p.W + gx) * p.lddx;
#pragma unroll
        for (int c = 0; c < CIN; ++c) dst[c] = dx[c];
    }
}


// ---- backward that recomputes y and the LayerNorm statistics from x -------------------------------------------
// The layer's input has 2 channels (8 bytes per pixel) and its output 16: keeping y and (mean, rstd) for the backward
// costs 72 bytes per pixel written in the forward and read back here, the dense dpre tensor another 64 written here and
// read by a separate weight-gradient kernel.  This kernel reads only x (+2-pixel halo) and dz (+1-pixel halo):
// y = lrelu(conv(x) + b) and its statistics are recomputed per halo pixel (288 FMAs), dpre goes to LDS only, dx is
// gathered from it, and (WG) the kernel gradient dW[(tap,ci)][o] += sum_p x[p + tap - 1][ci] * dpre[p][o] is two
// 16x16 MFMA tiles over the tile's 256 pixels (persistent blocks, block partials summed in block order afterwards).
template <int CIN, bool WG>
__global__ void __launch_bounds__(256) wdg_convln_bwdx_kernel(const WdgConvLn p, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ gamma) {
    constexpr int C4 = (CIN + 3) / 4;
    constexpr int XH = CN_TH + 4, XW = CN_TW + 4;
    constexpr int GH = CN_TH + 2, GW = CN_TW + 2;
    constexpr int ROWS = 9 * CIN, RT = (ROWS + 15) / 16;
    __shared__ __attribute__((aligned(16))) f32x4 xs[C4 * XH * XW];       // [c4][halo pixel]
    __shared__ __attribute__((aligned(16))) float dps[GH * GW * CN_CO];   // dpre on the halo, planes [o4][halo pixel]
    __shared__ float red[3 * CN_CO * 4];
    __shared__ float wred[WG ? 4 * RT * 256 : 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lq = lane >> 4;

    float ag[CN_CO], ab[CN_CO], abias[CN_CO];
#pragma unroll
    for (int o = 0; o < CN_CO; ++o) ag[o] = ab[o] = abias[o] = 0.f;
    f32x4 wacc[WG ? RT : 1];
    int a_base[RT];
    bool a_on[RT];
#pragma unroll
    for (int rt = 0; rt < (WG ? RT : 1); ++rt) wacc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = rt * 16 + li;
        a_on[rt] = row < ROWS;
        const int tap = a_on[rt] ? row / CIN : 0, ci = a_on[rt] ? row - tap * CIN : 0;
        const int th = tap / 3, tw = tap - 3 * th;
        a_base[rt] = (((ci >> 2) * (XH * XW) + (1 + th) * XW + 1 + tw) << 2) + (ci & 3);
    }
    const float* xsf = reinterpret_cast<const float*>(xs);

    const int ntiles = p.n_img * p.tiles_h * p.tiles_w;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tx = b % p.tiles_w;
        b /= p.tiles_w;
        const int ty = b % p.tiles_h;
        const int img = b / p.tiles_h;
        const int oy0 = ty * CN_TH, ox0 = tx * CN_TW;
        const float* dZimg = p.dZ + (long long)img * p.isdz;
        const float* Ximg = p.X + (long long)img * p.isx;
        // 1. x halo (zero outside the image = the conv's zero padding)
        for (int idx = t; idx < C4 * XH * XW; idx += 256) {
            const int c4 = idx / (XH * XW), pix = idx - c4 * (XH * XW);
            const int hy = pix / XW, hx = pix - hy * XW;
            const int gy = oy0 - 2 + hy, gx = ox0 - 2 + hx;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const f32x4*>(Ximg + ((long long)gy * p.W + gx) * p.ldx + 4 * c4);
            xs[idx] = v;
        }
        __syncthreads();
        // 2. y, its statistics and dpre on the (GH x GW) halo
        for (int hp = t; hp < GH * GW; hp += 256) {
            const int hy = hp / GW, hx = hp - hy * GW;
            const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            float d[CN_CO];
#pragma unroll
            for (int o = 0; o < CN_CO; ++o) d[o] = 0.f;
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
                float yv[CN_CO];
#pragma unroll
                for (int o = 0; o < CN_CO; ++o) yv[o] = bias[o];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int c4 = 0; c4 < C4; ++c4) {
                        const f32x4 xv = xs[c4 * (XH * XW) + (hy + tap / 3) * XW + hx + tap % 3];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = 4 * c4 + j;
                            if (c < CIN) {
                                const float* w = Wt + (tap * CIN + c) * CN_CO;
#pragma unroll
                                for (int o = 0; o < CN_CO; ++o) yv[o] = fmaf(xv[j], w[o], yv[o]);
                            }
                        }
                    }
                float s = 0.f;
#pragma unroll
                for (int o = 0; o < CN_CO; ++o) {
                    yv[o] = yv[o] > 0.f ? yv[o] : yv[o] * p.slope;
                    s += yv[o];
                }
                const float mean = s * (1.f / CN_CO);
                float q = 0.f;
#pragma unroll
                for (int o = 0; o < CN_CO; ++o) {
                    const float dd = yv[o] - mean;
                    q += dd * dd;
                }
                const float rstd = 1.f / sqrtf(q * (1.f / CN_CO) + p.eps);
                // (staging dz through LDS with coalesced loads — four lanes per pixel — measured SLOWER: 187 vs 167 us)
                const float* dzp = dZimg + ((long long)gy * p.W + gx) * p.lddz;
                float g[CN_CO], xh[CN_CO];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int o4 = 0; o4 < CN_CO / 4; ++o4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(dzp + 4 * o4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o = 4 * o4 + j;
                        xh[o] = (yv[o] - mean) * rstd;
                        g[o] = a[j];
                        const float gg = a[j] * gamma[o];
                        s1 += gg;
                        s2 += gg * xh[o];
                    }
                }
                s1 *= (1.f / CN_CO);
                s2 *= (1.f / CN_CO);
                const bool centre = hy >= 1 && hy <= CN_TH && hx >= 1 && hx <= CN_TW;
#pragma unroll
                for (int o = 0; o < CN_CO; ++o) {
                    float v = rstd * (g[o] * gamma[o] - s1 - xh[o] * s2);
                    v *= (yv[o] > 0.f ? 1.f : p.slope);
                    d[o] = v;
                    if (centre) {
                        ag[o] += g[o] * xh[o];
                        ab[o] += g[o];
                        abias[o] += v;
                    }
                }
            }
#pragma unroll
            for (int o4 = 0; o4 < CN_CO / 4; ++o4)
                *reinterpret_cast<f32x4*>(&dps[(o4 * (GH * GW) + hp) * 4]) = (f32x4){d[4 * o4], d[4 * o4 + 1], d[4 * o4 + 2], d[4 * o4 + 3]};
        }
        __syncthreads();
        if constexpr (WG) {
            // 2b. kernel gradient of this tile: wave w takes centre rows 2w, 2w+1 (64 pixels, 16 steps of 4)
#pragma unroll 4
            for (int st = 0; st < 16; ++st) {
                const int py = 2 * wave + (st >> 3), px_ = 4 * (st & 7) + lq;
                const int xo = (py * XW + px_) << 2;
                const float bv = dps[(((li >> 2) * (GH * GW) + (py + 1) * GW + px_ + 1) << 2) + (li & 3)];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float av = a_on[rt] ? xsf[a_base[rt] + xo] : 0.f;
                    wacc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, wacc[rt], 0, 0, 0);
                }
            }
        }
        if (p.dX) {
            // 3. dx[c] = sum_tap sum_o dpre[pixel + (1 - th, 1 - tw)][o] * W[tap][c][o]
            const int py = t >> 5, px = t & 31;
            float dx[CIN];
#pragma unroll
            for (int c = 0; c < CIN; ++c) dx[c] = 0.f;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int th = tap / 3, tw = tap % 3;
                const float* dp = &dps[((py + 2 - th) * GW + px + 2 - tw) * 4];
                float v[CN_CO];
#pragma unroll
                for (int o4 = 0; o4 < CN_CO / 4; ++o4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(dp + o4 * (GH * GW) * 4);
                    v[4 * o4] = a[0]; v[4 * o4 + 1] = a[1]; v[4 * o4 + 2] = a[2]; v[4 * o4 + 3] = a[3];
                }
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const float* w = Wt + (tap * CIN + c) * CN_CO;
#pragma unroll
                    for (int o = 0; o < CN_CO; ++o) dx[c] = fmaf(v[o], w[o], dx[c]);
                }
            }
            const int gy = oy0 + py, gx = ox0 + px;
            if (gy < p.H && gx < p.W) {
                float* dst = p.dX + (long long)img * p.isdx + ((long long)gy * p.W + gx) * p.lddx;
#pragma unroll
                for (int c = 0; c < CIN; ++c) dst[c] = dx[c];
            }
        }
        if (gridDim.x < (unsigned)ntiles) __syncthreads();   // persistent blocks: the next tile overwrites xs / dps
    }
    // parameter-gradient partials of all this block's tiles: wave reduction, 4 waves through LDS, one atomic per value
    if (p.dgamma) {
#pragma unroll
        for (int o = 0; o < CN_CO; ++o) {
            const float a0 = wdg_wave_sum(ag[o]), a1 = wdg_wave_sum(ab[o]), a2 = wdg_wave_sum(abias[o]);
            if (lane == 0) {
                red[(0 * CN_CO + o) * 4 + wave] = a0;
                red[(1 * CN_CO + o) * 4 + wave] = a1;
                red[(2 * CN_CO + o) * 4 + wave] = a2;
            }
        }
    }
    if constexpr (WG) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) wred[wave * (RT * 256) + rt * 256 + (4 * lq + r) * 16 + li] = wacc[rt][r];
    }
    __syncthreads();
    if (p.dgamma && t < 3 * CN_CO) {
        const float v = (red[t * 4] + red[t * 4 + 1]) + (red[t * 4 + 2] + red[t * 4 + 3]);
        float* dst = t < CN_CO ? p.dgamma + t : t < 2 * CN_CO ? p.dbeta + (t - CN_CO) : p.dbias + (t - 2 * CN_CO);
        atomicAdd(dst, v);
    }
    if constexpr (WG) {
        float* dst = p.dWpart + (long long)blockIdx.x * (RT * 256);
        for (int idx = t; idx < RT * 256; idx += 256)
            dst[idx] = (wred[idx] + wred[RT * 256 + idx]) + (wred[2 * RT * 256 + idx] + wred[3 * RT * 256 + idx]);
    }
}

// block partials [nblocks][RT*16][16] summed in block order -> dW [3][3][CIN][16] +=  (one wave per element)
template <int CIN>
__global__ void __launch_bounds__(256) wdg_convln_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks,
                                                                      float* __restrict__ dW) {
    constexpr int ROWS = 9 * CIN, RT = (ROWS + 15) / 16;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= ROWS * CN_CO) return;
    float v = 0.f;
    for (int b = lane; b < nblocks; b += 64) v += part[(long long)b * (RT * 256) + idx];
    v = wdg_wave_sum(v);
    if (lane == 0) dW[idx] += v;
}

// Measured (profiles/r01l): for 16 -> 16 the scalar-weight FMA form only ties the MFMA halo kernel forward and
// loses 20 % backward, so only the 2 -> 16 layer (K = 18, where MFMA padding dominates) is routed here.
extern "C" int wdg_convln_supported(int cin, int cout) { return cout == 16 && cin == 2; }

extern "C" int wdg_convln_fwd(const float* x, int ldx, int64_t isx, const float* w_hwio, const float* bias,
                              const float* gamma, const float* beta, float eps, float slope, float* y, int ldy,
                              int64_t isy, float* z, int ldz, int64_t isz, float* mean_rstd, int n_img, int H, int W,
                              int cin, int cout, wdg_stream stream) {
    WDG_CHECK_ARG(x && w_hwio && bias && gamma && beta && z, "null argument");
    WDG_CHECK_ARG((y != nullptr) == (mean_rstd != nullptr), "y and mean_rstd: both (for wdg_convln_bwd) or neither (wdg_convln_bwd_x)");
    WDG_CHECK_ARG(wdg_convln_supported(cin, cout), "unsupported (cin, cout)");
    WDG_CHECK_ARG(((long long)H * W * ldx + (256 / ((long long)H * W) + 1) * (long long)isx) * 4 < (1LL << 31),
                  "the images of x a workgroup's 256 pixels touch must stay below 2 GiB (one buffer descriptor)");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ldx % 4 == 0 && ((uintptr_t)y & 15) == 0 && ldy % 4 == 0 &&
                      ((uintptr_t)z & 15) == 0 && ldz % 4 == 0, "alignment");
    WdgConvLn p;
    memset(&p, 0, sizeof(p));
    p.X = x; p.Y = y; p.Z = z; p.MR = mean_rstd;
    p.isx = isx; p.isy = isy; p.isz = isz;
    p.n_img = n_img; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy; p.ldz = ldz;
    p.eps = eps; p.slope = slope;
    const long long P = (long long)n_img * H * W;
    dim3 grid((unsigned)((P + 255) / 256)), block(256);
    hipLaunchKernelGGL(wdg_convln_fwd_kernel<2>, grid, block, 0, (hipStream_t)stream, p, w_hwio, bias, gamma, beta);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_convln_bwd(const float* dz, int lddz, int64_t isdz, const float* y, int ldy, int64_t isy,
                              const float* mean_rstd, const float* w_hwio, const float* gamma, float slope,
                              float* dpre, float* dx, int lddx, int64_t isdx, float* dgamma, float* dbeta,
                              float* dbias, int n_img, int H, int W, int cin, int cout, wdg_stream stream) {
    WDG_CHECK_ARG(dz && y && mean_rstd && w_hwio && gamma && dpre, "null argument");
    WDG_CHECK_ARG(wdg_convln_supported(cin, cout), "unsupported (cin, cout)");
    WDG_CHECK_ARG((dgamma && dbeta && dbias) || (!dgamma && !dbeta && !dbias), "parameter gradients: all or none");
    WDG_CHECK_ARG(((uintptr_t)dz & 15) == 0 && lddz % 4 == 0 && ((uintptr_t)y & 15) == 0 && ldy % 4 == 0, "alignment");
    WdgConvLn p;
    memset(&p, 0, sizeof(p));
    p.dZ = dz; p.Y = const_cast<float*>(y); p.MR = const_cast<float*>(mean_rstd); p.dPre = dpre; p.dX = dx;
    p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
    p.isdz = isdz; p.isy = isy; p.isdx = isdx;
    p.n_img = n_img; p.H = H; p.W = W; p.lddz = lddz; p.ldy = ldy; p.lddx = lddx;
    p.slope = slope;
    p.tiles_h = (H + CN_TH - 1) / CN_TH;
    p.tiles_w = (W + CN_TW - 1) / CN_TW;
    dim3 grid((unsigned)((long long)n_img * p.tiles_h * p.tiles_w)), block(256);
    hipLaunchKernelGGL(wdg_convln_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, p, w_hwio, gamma);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" size_t wdg_convln_wgrad_ws_bytes(int n_img, int H, int W, int cin) {
    const long long ntiles = (long long)n_img * ((H + CN_TH - 1) / CN_TH) * ((W + CN_TW - 1) / CN_TW);
    return (size_t)std::min<long long>(ntiles, 4096) * ((9 * cin + 15) / 16) * 256 * sizeof(float);
}

// Backward from x instead of y / (mean, rstd) (pair it with wdg_convln_fwd(y = NULL, mean_rstd = NULL)): dx (optional),
// dgamma / dbeta / dbias += (all or none), dw [3][3][cin][cout] += (optional; needs ws of wdg_convln_wgrad_ws_bytes()).
extern "C" int wdg_convln_bwd_x(const float* dz, int lddz, int64_t isdz, const float* x, int ldx, int64_t isx,
                                const float* w_hwio, const float* bias, const float* gamma, float eps, float slope,
                                float* dx, int lddx, int64_t isdx, float* dgamma, float* dbeta, float* dbias, float* dw,
                                void* ws, size_t ws_bytes, int n_img, int H, int W, int cin, int cout, wdg_stream stream) {
    WDG_CHECK_ARG(dz && x && w_hwio && bias && gamma, "null argument");
    WDG_CHECK_ARG(wdg_convln_supported(cin, cout), "unsupported (cin, cout)");
    WDG_CHECK_ARG((dgamma && dbeta && dbias) || (!dgamma && !dbeta && !dbias), "parameter gradients: all or none");
    WDG_CHECK_ARG(((uintptr_t)dz & 15) == 0 && lddz % 4 == 0 && ((uintptr_t)x & 15) == 0 && ldx % 4 == 0, "alignment");
    WdgConvLn p;
    memset(&p, 0, sizeof(p));
    p.dZ = dz; p.X = x; p.dX = dx;
    p.dgamma = dgamma; p.dbeta = dbeta; p.dbias = dbias;
    p.isdz = isdz; p.isx = isx; p.isdx = isdx;
    p.n_img = n_img; p.H = H; p.W = W; p.lddz = lddz; p.ldx = ldx; p.lddx = lddx;
    p.eps = eps; p.slope = slope;
    p.tiles_h = (H + CN_TH - 1) / CN_TH;
    p.tiles_w = (W + CN_TW - 1) / CN_TW;
    const long long ntiles = (long long)n_img * p.tiles_h * p.tiles_w;
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256);
    if (!dw) {
        hipLaunchKernelGGL((wdg_convln_bwdx_kernel<2, false>), dim3((unsigned)ntiles), block, 0, st, p, w_hwio, bias, gamma);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    const int nb = (int)std::min<long long>(ntiles, (long long)cus * 3);   // 154 registers -> 3 resident workgroups per CU
    const size_t need = (size_t)nb * ((9 * cin + 15) / 16) * 256 * sizeof(float);
    if (!ws || ws_bytes < need) {
        wdg_set_error("wdg_convln_bwd_x: scratch too small (%zu < %zu)", ws_bytes, need);
        return WDG_ERR_WORKSPACE;
    }
    p.dWpart = (float*)ws;
    hipLaunchKernelGGL((wdg_convln_bwdx_kernel<2, true>), dim3((unsigned)nb), block, 0, st, p, w_hwio, bias, gamma);
    hipLaunchKernelGGL(wdg_convln_wgrad_reduce_kernel<2>, dim3((unsigned)((9 * cin * CN_CO + 3) / 4)), block, 0, st,
                       p.dWpart, nb, dw);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
