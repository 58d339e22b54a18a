// conv_halo_bf16.hip — inference-precision halo-tile convolution (bf16 operands on v_mfma_f32_16x16x32_bf16,
// fp32 accumulation, fp32 activations in memory): the bf16 counterpart of conv_halo.hip for the generator's last
// two layers, UpSampling2D(bilinear) + Conv2DTranspose(5x5) (models.py:62-64) and the 16 -> 2 output conv (:70).
//
// With the matrix work 8x cheaper the kernel is bound by how fast the input halo can be produced, so in
// upsample mode the staging is two-stage: (1) the low-resolution fp32 source tile (8 x 20 pixels x 32 channels)
// is copied to LDS once per channel chunk, (2) the 12 x 36 upsampled halo is interpolated LDS -> LDS in fp32 and
// rounded to bf16 (4 LDS reads per output slot instead of 8 global loads).  Weight fragments are read straight
// from global memory (bf16 copy of the packed weights, L1/L2 resident).
#include "conv_plan.h"
#include "h16.h"
#include <algorithm>

// operand format FMT: 0 = bf16, 1 = IEEE fp16 (same kernels, see conv_igemm_bf16.hip)
constexpr int HB_TH = 8, HB_TW = 32;

struct WdgHaloBf16 {
    const float* A;
    float* Out;
    const float* bias;
    const float* affine;
    const int4* taps;   // {dh, dw, b_off0, 0}
    long long imgStrideA, imgStrideO;
    int n_img, H, W, ldA;   // A as stored (low-res dims in upsample mode)
    int Hc, Wc;             // conv-input dims
    int Ho, Wo, ldO;
    int ntaps, C8;          // taps, channel groups of 8
    int Ncols, ldB;
    int dh_min, dw_min, halo_h, halo_w, npix;
    int act, accumulate, upsample;
    float slope;
    int tiles_h, tiles_w;
    int lr_h, lr_w;         // low-res staging tile (upsample mode)
    int in16;               // (thin kernel) A holds 16-bit elements of the operand format; ldA / imgStrideA in elements
};

template <int FMT>
__device__ __forceinline__ wdg_h16x8<FMT> hb_pack_t(const f32x4& a, const f32x4& b) {
    wdg_h16x8<FMT> v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (wdg_h16<FMT>)a[j];
        v[4 + j] = (wdg_h16<FMT>)b[j];
    }
    return v;
}

template <int NT, int FMT>
__global__ void __launch_bounds__(256) wdg_conv_halo_bf16_kernel(const WdgHaloBf16 p, const wdg_h16<FMT>* __restrict__ Bw) {
    typedef wdg_h16x8<FMT> bf16x8;
    auto hb_pack = [](const f32x4& a, const f32x4& b) { return hb_pack_t<FMT>(a, b); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);                       // [4][npix] bf16 halo
    f32x4* lds_lr = reinterpret_cast<f32x4*>(smem_raw + (size_t)4 * p.npix * 16);  // [lr_h*lr_w][8] fp32 low-res tile

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h;
    const int img = bid / p.tiles_h;
    const int oy0 = ty * HB_TH, ox0 = tx * HB_TW;
    const int hy0 = oy0 + p.dh_min, hx0 = ox0 + p.dw_min;
    const int npr = p.halo_h * p.halo_w;
    const float* Aimg = p.A + (long long)img * p.imgStrideA;
    // low-res origin of the staging tile: row of the first upsampled halo row minus one (clamped per pixel)
    const int ly0 = (hy0 >> 1) - 1, lx0 = (hx0 >> 1) - 1;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 acc[4][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = z4;

    const int nchunk = (p.C8 + 3) >> 2;             // chunks of 32 channels
    for (int ck = 0; ck < nchunk; ++ck) {
        const int kgs = min(4, p.C8 - 4 * ck);
        __syncthreads();
        if (p.upsample) {
            // ---- stage 1: low-res fp32 tile, coordinates clamped (= bilinear edge clamp)
            const int nl = p.lr_h * p.lr_w;
            for (int idx = t; idx < nl * kgs * 2; idx += 256) {
                const int c4 = idx % (kgs * 2);
                const int pix = idx / (kgs * 2);
                const int y = pix / p.lr_w, x = pix - y * p.lr_w;
                const int gy = min(max(ly0 + y, 0), p.H - 1), gx = min(max(lx0 + x, 0), p.W - 1);
                lds_lr[pix * 8 + c4] = *reinterpret_cast<const f32x4*>(Aimg + ((long long)gy * p.W + gx) * p.ldA + (8 * ck) * 4 + 4 * c4);
            }
            __syncthreads();
            // ---- stage 2: upsampled halo, fp32 interpolation, bf16 rounding
            for (int idx = t; idx < kgs * npr; idx += 256) {
                const int kg = idx / npr;
                const int pix = idx - kg * npr;
                const int hy = pix / p.halo_w, hx = pix - hy * p.halo_w;
                const int gy = hy0 + hy, gx = hx0 + hx;
                bf16x8 v = hb_pack(z4, z4);
                if ((unsigned)gy < (unsigned)p.Hc && (unsigned)gx < (unsigned)p.Wc) {
                    const int jh = gy >> 1, jw = gx >> 1;
                    const int h0 = (gy & 1) ? jh : max(jh - 1, 0), h1 = (gy & 1) ? min(jh + 1, p.H - 1) : jh;
                    const int w0 = (gx & 1) ? jw : max(jw - 1, 0), w1 = (gx & 1) ? min(jw + 1, p.W - 1) : jw;
                    const float fh = (gy & 1) ? 0.25f : 0.75f, fw = (gx & 1) ? 0.25f : 0.75f;
                    // clamped global coordinates map to staged slots: slot(y) = clamp(y) - clamp-free origin
                    const int r0 = (h0 - ly0) * p.lr_w, r1 = (h1 - ly0) * p.lr_w;
                    const int c0 = w0 - lx0, c1 = w1 - lx0;
                    f32x4 o[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 a00 = lds_lr[(r0 + c0) * 8 + 2 * kg + q], a01 = lds_lr[(r0 + c1) * 8 + 2 * kg + q];
                        const f32x4 a10 = lds_lr[(r1 + c0) * 8 + 2 * kg + q], a11 = lds_lr[(r1 + c1) * 8 + 2 * kg + q];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float top = a00[j] + (a01[j] - a00[j]) * fw;
                            const float bot = a10[j] + (a11[j] - a10[j]) * fw;
                            o[q][j] = top + (bot - top) * fh;
                        }
                    }
                    v = hb_pack(o[0], o[1]);
                }
                lds_a[kg * p.npix + pix] = v;
            }
        } else {
            for (int idx = t; idx < kgs * npr; idx += 256) {
                const int kg = idx / npr;
                const int pix = idx - kg * npr;
                const int hy = pix / p.halo_w, hx = pix - hy * p.halo_w;
                const int gy = hy0 + hy, gx = hx0 + hx;
                f32x4 a = z4, b = z4;
                if ((unsigned)gy < (unsigned)p.Hc && (unsigned)gx < (unsigned)p.Wc) {
                    const float* src = Aimg + ((long long)gy * p.W + gx) * p.ldA + (4 * ck + kg) * 8;
                    a = *reinterpret_cast<const f32x4*>(src);
                    b = *reinterpret_cast<const f32x4*>(src + 4);
                }
                lds_a[kg * p.npix + pix] = hb_pack(a, b);
            }
        }
        __syncthreads();
        const bool kvalid = lg < kgs;
        for (int tap = 0; tap < p.ntaps; ++tap) {
            const int4 e = p.taps[tap];
            const int rowoff = (e.x - p.dh_min) * p.halo_w + (e.y - p.dw_min);
            bf16x8 af[4], bf[NT];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int row = 2 * wave + (a >> 1), col = (a & 1) * 16 + li;
                af[a] = kvalid ? lds_a[lg * p.npix + row * p.halo_w + col + rowoff] : hb_pack(z4, z4);
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = b * 16 + li;
                bf[b] = (kvalid && n < p.Ncols)
                            ? *reinterpret_cast<const bf16x8*>(Bw + (long long)n * p.ldB + e.z + (4 * ck + lg) * 8)
                            : hb_pack(z4, z4);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = wdg_mfma16<FMT>(af[a], bf[b], acc[a][b]);
        }
    }

#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int oy = oy0 + 2 * wave + (a >> 1);
        if (oy >= p.Ho) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ox = ox0 + (a & 1) * 16 + lg * 4 + r;
            if (ox >= p.Wo) continue;
            float* dst = p.Out + (long long)img * p.imgStrideO + ((long long)oy * p.Wo + ox) * p.ldO;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = b * 16 + li;
                if (n < p.Ncols) {
                    float v = acc[a][b][r];
                    if (p.bias) v += p.bias[n];
                    if (p.act) v = wdg_lrelu(v, p.slope);
                    if (p.affine) v = v * p.affine[n] + p.affine[p.Ncols + n];
                    if (p.accumulate) v += dst[n];
                    dst[n] = v;
                }
            }
        }
    }
}

// ---- 3 x 3, 16 input channels, <= 4 output channels (the generator's 16 -> 2 output conv, models.py:70) -------------------
// The general kernel above spends this layer's time outside the matrix pipe: its staging loop is one dependent load per
// iteration, its weight fragments come from global memory inside the tap loop, and with 2 live output columns of a 16-column
// tile its epilogue is sixteen predicated 4-byte stores per thread.  Here: persistent workgroups (the next tile's halo is
// requested branch-free under this tile's MFMAs), the nine taps' weight fragments in registers for the workgroup's life,
// operands swapped (A = weights, B = pixels) so that a lane holds the four (padded) output channels of ONE pixel and stores
// them as 16 bytes, and two taps per MFMA (k = 32 = 2 taps x 16 channels: five MFMAs per 16-pixel fragment).
template <int FMT>
__global__ void __launch_bounds__(256) wdg_conv_thin16_h16_kernel(const WdgHaloBf16 p, const wdg_h16<FMT>* __restrict__ Bw) {
    typedef wdg_h16x8<FMT> bf16x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);           // [2 kg][npix] (+ 1 slot for the staging slots beyond the halo)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int npr = p.halo_h * p.halo_w;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight fragments: MFMA i multiplies taps 2 i and 2 i + 1; this lane's k-octet lg is channel group lg & 1 of tap 2 i + (lg >> 1)
    bf16x8 wf[5];
    int poff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tap = 2 * i + (lg >> 1), kg = lg & 1;
        const int4 e = p.taps[tap < p.ntaps ? tap : 0];
        wf[i] = hb_pack_t<FMT>(z4, z4);
        if (tap < p.ntaps && li < p.Ncols) wf[i] = *reinterpret_cast<const bf16x8*>(Bw + (long long)li * p.ldB + e.z + kg * 8);
        poff[i] = kg * p.npix + (e.x - p.dh_min) * p.halo_w + (e.y - p.dw_min);     // (tap 9: any slot, its weights are zero)
    }
    // epilogue constants of this lane's four channels (lanes lg > 0 hold padding columns)
    f32x4 bias4 = z4, sc4 = (f32x4){1.f, 1.f, 1.f, 1.f}, sh4 = z4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool on = lg == 0 && r < p.Ncols;
        if (p.bias && on) bias4[r] = p.bias[r];
        if (p.affine) { sc4[r] = on ? p.affine[r] : 0.f; sh4[r] = on ? p.affine[p.Ncols + r] : 0.f; }
    }
    // staging slots (tile independent): slot = (pixel, channel group), two consecutive lanes = the 64 contiguous bytes of a pixel
    constexpr int NSL = 3;                                         // 2 * 10 * 34 = 680 slots over 256 threads
    int shy[NSL], shx[NSL], sslot[NSL], soff[NSL];
#pragma unroll
    for (int s_ = 0; s_ < NSL; ++s_) {
        const int idx = t + 256 * s_;
        const int pix = idx >> 1, kg = idx & 1;
        const bool on = pix < npr;
        shy[s_] = on ? pix / p.halo_w : (1 << 28);
        shx[s_] = pix - (pix / p.halo_w) * p.halo_w;
        soff[s_] = ((pix / p.halo_w) * p.W + shx[s_]) * p.ldA + kg * 8;       // elements
        sslot[s_] = on ? kg * p.npix + pix : 2 * p.npix;
    }
    const int ntiles = p.n_img * p.tiles_h * p.tiles_w;
    f32x4 rs[NSL][2];
    auto request = [&](int tile) __attribute__((always_inline)) {
        const int img = tile / (p.tiles_h * p.tiles_w);
        const int rem = tile - img * (p.tiles_h * p.tiles_w);
        const int ty = rem / p.tiles_w, tx = rem - ty * p.tiles_w;
        const int hy0 = ty * HB_TH + p.dh_min, hx0 = tx * HB_TW + p.dw_min;
        const int org = (hy0 * p.W + hx0) * p.ldA;
        if (p.in16) {
            // the producer stored the activations in the operand format already (the same rounding, one step earlier): eight
            // channels are one 16-byte request, no conversion
            const wdg_srd srdA = wdg_make_srd(reinterpret_cast<const wdg_h16<FMT>*>(p.A) + (long long)img * p.imgStrideA);
#pragma unroll
            for (int s_ = 0; s_ < NSL; ++s_) {
                const int gy = hy0 + shy[s_], gx = hx0 + shx[s_];
                const unsigned neg = (unsigned)((gy | (p.Hc - 1 - gy) | gx | (p.Wc - 1 - gx)) >> 31);
                rs[s_][0] = wdg_buffer_load_f32x4(srdA, ((unsigned)((org + soff[s_]) * 2) & ~neg) | (neg & 0x80000000u));
            }
            return;
        }
        const wdg_srd srdA = wdg_make_srd(p.A + (long long)img * p.imgStrideA);
#pragma unroll
        for (int s_ = 0; s_ < NSL; ++s_) {
            const int gy = hy0 + shy[s_], gx = hx0 + shx[s_];
            const unsigned neg = (unsigned)((gy | (p.Hc - 1 - gy) | gx | (p.Wc - 1 - gx)) >> 31);
            const unsigned off = ((unsigned)((org + soff[s_]) * 4) & ~neg) | (neg & 0x80000000u);
            rs[s_][0] = wdg_buffer_load_f32x4(srdA, off);
            rs[s_][1] = wdg_buffer_load_f32x4(srdA, off + 16);
        }
    };
    if ((int)blockIdx.x < ntiles) request(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                           // the previous tile's fragment reads are done
#pragma unroll
        for (int s_ = 0; s_ < NSL; ++s_)
            lds_a[sslot[s_]] = p.in16 ? __builtin_bit_cast(bf16x8, rs[s_][0]) : hb_pack_t<FMT>(rs[s_][0], rs[s_][1]);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);
        const int img = tile / (p.tiles_h * p.tiles_w);
        const int rem = tile - img * (p.tiles_h * p.tiles_w);
        const int ty = rem / p.tiles_w, tx = rem - ty * p.tiles_w;
        f32x4 acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int pb = (2 * wave + (f >> 1)) * p.halo_w + (f & 1) * 16 + li;
            bf16x8 b[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) b[i] = lds_a[poff[i] + pb];
            acc[f] = z4;
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[f] = wdg_mfma16<FMT>(wf[i], b[i], acc[f]);
        }
        // register r of lane (li, lg): output channel 4 lg + r of pixel li of the fragment
        if (lg == 0) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int oy = ty * HB_TH + 2 * wave + (f >> 1), ox = tx * HB_TW + (f & 1) * 16 + li;
                if (oy >= p.Ho || ox >= p.Wo) continue;
                f32x4 v = acc[f] + bias4;
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], p.slope);
                }
                if (p.affine) v = v * sc4 + sh4;
                *reinterpret_cast<f32x4*>(p.Out + (long long)img * p.imgStrideO + ((long long)oy * p.Wo + ox) * p.ldO) = v;
            }
        }
    }
}

static int g_thin16 = 1;     // wdg_set_tuning("halo16_thin", 0/1)
void wdg_halo_bf16_set_thin(int v) { g_thin16 = v != 0; }

static int launch_halo_bf16(const wdg_conv_plan* pl, bool dgrad, const float* A, int ldA, long long imgStrideA,
                            int upsample, const void* B16, const float* bias, const float* affine, float* Out, int act,
                            float slope, int accumulate, int fmt, hipStream_t st, int in16 = 0) {
    const wdg_conv_geom& g = pl->g;
    WdgHaloBf16 p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.Out = Out; p.bias = bias; p.affine = affine;
    p.n_img = g.n_img; p.ldA = ldA; p.imgStrideA = imgStrideA;
    p.ntaps = pl->taps;
    p.act = act; p.slope = slope; p.accumulate = accumulate; p.upsample = upsample;
    int nt, cp;
    if (!dgrad) {
        p.taps = pl->d_taps_fwd;
        p.Hc = g.H; p.Wc = g.W; p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy; p.imgStrideO = g.img_stride_y;
        cp = pl->Cin_p; p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p;
        p.dh_min = -g.pad_h; p.dw_min = -g.pad_w;
        nt = pl->halo_fwd_nt;
    } else {
        p.taps = pl->d_taps_dgrad;
        p.Hc = g.Ho; p.Wc = g.Wo; p.Ho = g.H; p.Wo = g.W; p.ldO = g.ldx; p.imgStrideO = g.img_stride_x;
        cp = pl->Cout_p; p.Ncols = g.Cin; p.ldB = pl->Cout_p;
        p.dh_min = g.pad_h - (g.kh - 1); p.dw_min = g.pad_w - (g.kw - 1);
        nt = pl->halo_dgrad_nt;
    }
    if (!nt || cp % 8 != 0) {
        wdg_set_error("halo_bf16: plan not eligible (stride 1, k <= 5, <= 64 output channels, channels %% 8 == 0)");
        return WDG_ERR_ARG;
    }
    p.C8 = cp / 8;
    if (upsample) {
        if ((p.Hc & 1) || (p.Wc & 1)) {
            wdg_set_error("halo_bf16: upsample mode needs even conv-input dims");
            return WDG_ERR_ARG;
        }
        p.H = p.Hc / 2; p.W = p.Wc / 2;
    } else {
        p.H = p.Hc; p.W = p.Wc;
    }
    p.halo_h = HB_TH + g.kh - 1; p.halo_w = HB_TW + g.kw - 1;
    p.npix = wdg_round_up(p.halo_h * p.halo_w, 16);
    p.tiles_h = (p.Ho + HB_TH - 1) / HB_TH;
    p.tiles_w = (p.Wo + HB_TW - 1) / HB_TW;
    p.lr_h = p.halo_h / 2 + 3; p.lr_w = p.halo_w / 2 + 3;
    const size_t lds = (size_t)4 * p.npix * 16 + (upsample ? (size_t)p.lr_h * p.lr_w * 8 * 16 : 0);
    dim3 grid((unsigned)((long long)g.n_img * p.tiles_h * p.tiles_w)), block(256);
    // the 16 -> 2 output conv (3 x 3, 16 padded input channels, <= 4 output channels whose padded pixel stride is 4 floats)
    p.in16 = in16;
    const bool thin = !upsample && !accumulate && p.ntaps == 9 && g.kh == 3 && g.kw == 3 && p.C8 == 2 && p.Ncols <= 4 && p.ldO == 4 &&
                      ldA % (in16 ? 8 : 4) == 0 && ((uintptr_t)Out & 15) == 0 && ((uintptr_t)A & 15) == 0 &&
                      (long long)p.Hc * p.Wc * ldA * 4 < (1LL << 31);
    if (in16 && !thin) {
        wdg_set_error("halo_bf16: 16-bit activations are read by the 3 x 3, 16 -> (<= 4) channel kernel only");
        return WDG_ERR_ARG;
    }
    if ((g_thin16 || in16) && thin) {
        const size_t lds1 = ((size_t)2 * p.npix + 1) * 16;
        const unsigned nb = (unsigned)std::min<long long>((long long)grid.x, (long long)pl->cus * 8);
        if (fmt == 0) hipLaunchKernelGGL((wdg_conv_thin16_h16_kernel<0>), dim3(nb), block, lds1, st, p, (const __bf16*)B16);
        else hipLaunchKernelGGL((wdg_conv_thin16_h16_kernel<1>), dim3(nb), block, lds1, st, p, (const _Float16*)B16);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
#define WDG_HB_CASE(NT_)                                                                                              \
    if (nt == NT_) {                                                                                              \
        if (fmt == 0) hipLaunchKernelGGL((wdg_conv_halo_bf16_kernel<NT_, 0>), grid, block, lds, st, p, (const __bf16*)B16);   \
        else hipLaunchKernelGGL((wdg_conv_halo_bf16_kernel<NT_, 1>), grid, block, lds, st, p, (const _Float16*)B16);          \
    }
    WDG_HB_CASE(1) WDG_HB_CASE(2) WDG_HB_CASE(4)
#undef WDG_HB_CASE
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// 16-bit forward of a thin stride-1 conv (<= 64 output channels): y = affine(act(conv(x, wF16) + bias))
static int halo_fwd_h16(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias, const float* affine,
                        float* y, int act, float slope, int fmt, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF16 && y, "null argument");
    return launch_halo_bf16(pl, false, x, pl->g.ldx, pl->g.img_stride_x, 0, wF16, bias, affine, y, act, slope, 0, fmt,
                            (hipStream_t)stream);
}
extern "C" int wdg_conv_halo_fwd_bf16(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias,
                                      const float* affine, float* y, int act, float slope, wdg_stream stream) {
    return halo_fwd_h16(pl, x, wF16, bias, affine, y, act, slope, 0, stream);
}
extern "C" int wdg_conv_halo_fwd_f16(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias,
                                     const float* affine, float* y, int act, float slope, wdg_stream stream) {
    return halo_fwd_h16(pl, x, wF16, bias, affine, y, act, slope, 1, stream);
}

// 16-bit counterpart of wdg_upconv_fwd: y = affine(act(convT(upsample2x(x_low), wD16) + bias))
static int upconv_fwd_h16(const wdg_conv_plan* pl, const float* x_low, int ld_low, int64_t img_stride_low, const void* wD16,
                          const float* bias, const float* affine, float* y, int act, float slope, int fmt, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x_low && wD16 && y, "null argument");
    WDG_CHECK_ARG(((uintptr_t)x_low & 15) == 0 && ld_low % 4 == 0, "x_low must be 16-byte aligned, ld % 4 == 0");
    return launch_halo_bf16(pl, true, x_low, ld_low, img_stride_low, 1, wD16, bias, affine, y, act, slope, 0, fmt,
                            (hipStream_t)stream);
}
extern "C" int wdg_upconv_fwd_bf16(const wdg_conv_plan* pl, const float* x_low, int ld_low, int64_t img_stride_low,
                                   const void* wD16, const float* bias, const float* affine, float* y, int act,
                                   float slope, wdg_stream stream) {
    return upconv_fwd_h16(pl, x_low, ld_low, img_stride_low, wD16, bias, affine, y, act, slope, 0, stream);
}
extern "C" int wdg_upconv_fwd_f16(const wdg_conv_plan* pl, const float* x_low, int ld_low, int64_t img_stride_low,
                                  const void* wD16, const float* bias, const float* affine, float* y, int act,
                                  float slope, wdg_stream stream) {
    return upconv_fwd_h16(pl, x_low, ld_low, img_stride_low, wD16, bias, affine, y, act, slope, 1, stream);
}

// The same convolution reading its input in the 16-BIT operand format (x16 [n, H, W, ldx16 >= 16] bf16 (fmt 0) / fp16 (fmt 1),
// strides in elements): for the generator's output conv (models.py:70) behind wdg_upconv_fused_h16(out16 = 1) — the value the
// kernel multiplies is the rounded one either way, so the result is the one of wdg_conv_halo_fwd_bf16 / _f16 bit for bit.
// 3 x 3, stride 1, 16 (padded) input channels, <= 4 output channels with a pixel stride of 4 floats.
extern "C" int wdg_conv_thin16_fwd_h16(const wdg_conv_plan* pl, const void* x16, int ldx16, int64_t img_stride_x16, const void* wF16,
                                       int fmt, const float* bias, const float* affine, float* y, int act, float slope,
                                       wdg_stream stream) {
    WDG_CHECK_ARG(pl && x16 && wF16 && y && (fmt == 0 || fmt == 1), "bad argument");
    return launch_halo_bf16(pl, false, reinterpret_cast<const float*>(x16), ldx16, img_stride_x16, 0, wF16, bias, affine, y, act, slope, 0,
                            fmt, (hipStream_t)stream, 1);
}
