// conv_halo.hip — halo-tile convolution for stride-1 layers with few output channels (<= 64).
//
// The implicit-GEMM kernel re-fetches the shifted input window from L2 for every tap; for layers whose
// output-channel count is small (N <= 64: the 5x5 transposed conv after the bilinear upsample,
// models.py:60-64; the 16->2 output conv, :70; the discriminator's full-resolution 2/5/16-channel
// ConvLSTM / conv layers, :93-104, and all their data gradients) that traffic, not the MFMA, is the
// bound.  Here a block owns an 8x32 output tile, stages the (8+kh-1)x(32+kw-1) input halo of 16 channels
// ONCE into LDS and reuses it for all kh*kw taps, so every input element is read from L2/HBM once
// (plus the halo overlap) instead of kh*kw times.  Optionally the staged input is the bilinear x2
// upsampling of a low-resolution tensor computed on the fly (UpSampling2D + Conv2DTranspose fused:
// the 160-channel upsampled tensor, 1.3 GB at batch 32, is never materialised).
//
// MFMA mapping as in conv_igemm.hip: rows = 16 consecutive output pixels of one image row,
// cols = output channels, k = 4 channels per lane group; LDS halo layout [kg][pixel] float4 with the
// pixel count padded to a multiple of 16, which makes every ds_read_b128 fragment read conflict-free
// for any tap displacement.
#include "conv_plan.h"
#include "convlstm2.h"
#include <algorithm>

constexpr int HALO_TH = 8, HALO_TW = 32;

struct WdgHalo {
    const float* A;
    const float* B;
    float* Out;
    const float* bias;
    const int4* taps;  // {dh, dw, b_off0, 0}
    long long imgStrideA, imgStrideO;
    int n_img, H, W, ldA;   // A tensor as stored (low-res dims in upsample mode)
    int Hc, Wc;             // conv-input dims (= 2H, 2W in upsample mode, else H, W)
    int Ho, Wo, ldO;
    int ntaps, C4;          // taps, channel groups of 4
    int Ncols, ldB;
    int dh_min, dw_min, halo_h, halo_w, npix;  // npix = halo_h*halo_w rounded up to 16
    int act, accumulate, upsample;
    float slope;
    int tiles_h, tiles_w;
    int lr_h, lr_w;   // low-res staging tile (upsample mode): rows/cols of the source covering the halo
    // ConvLSTM2D recurrent step (lstm_F > 0; models.py:93,101 at n_timesteps > 1): Out holds the input part of the gates
    // (columns [i | f | c~ | o], lstm_F each) and receives the pre-activation sums the backward pass reads; the cell update
    // runs on the accumulators and writes c_out / h_out.  lstm_F = 16 (4 column tiles = 4 gates) or 2 (8 columns in one tile).
    int lstm_F, ldc, ldh;
    const float* c_prev;
    float* c_out;
    float* h_out;
    // backward step (lstm_bwd != 0, data-gradient direction, Ncols = lstm_F): Out is dh_{t-1} — the accumulated result is the
    // complete gradient of h_{t-1}, so the cell backward of timestep t-1 follows in the epilogue: reads gates / c of t-1 and the
    // dc flowing in from t, writes dgates_{t-1} and (dc_out != NULL) the dc flowing on to t-2.  c_prev == NULL at t-1 = 0.
    // LayerNormalization over the 16 output channels in the epilogue of the persistent 3x3 kernel (ln_gamma != NULL; models.py
    // :102-105): Out keeps y = act(conv + bias) for the backward pass, Out2 (own pixel / image stride) receives z, mean_rstd the
    // per-pixel statistics
    float* Out2;
    const float* ln_gamma;
    const float* ln_beta;
    float* mean_rstd;
    float ln_eps;
    int ldO2;
    long long imgStrideO2;
    int lstm_bwd;
    const float* gates_t;     // [pixel][4 * lstm_F] pre-activations of the timestep whose cell is differentiated
    const float* c_cur;       // its cell state
    const float* dc_in;
    float* dgates_out;
    float* dc_out;
};

// WG = 1: the weight fragments are read straight from global memory (they are a few hundred KB, L1/L2
// resident, and each lane needs exactly one float4 per tap and column tile), which frees the LDS weight
// stage (25.6 of 52 KB for the 5x5 layer) and lets more blocks share a CU.
// TH = 8 (default) or 4 output rows per tile: a launch with fewer than two 8-row tiles per CU (the per-timestep recurrent
// convolutions at n_timesteps > 1: 288 tiles on 256 CUs) halves its tiles to double the number of workgroups.
template <int NT, int WG, int TH = HALO_TH>
__global__ void __launch_bounds__(256) wdg_conv_halo_kernel(const WdgHalo p, const float* __restrict__ Bw) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    f32x4* lds_a = smem;                 // [4][npix]
    f32x4* lds_w = smem + 4 * p.npix;    // [ntaps][4][NT*16]   (WG = 0 only)
    f32x4* lds_lr = smem + 4 * p.npix + (WG ? 0 : p.ntaps * 4 * NT * 16);   // [lr_h*lr_w][4] low-res tile (upsample)
    constexpr int NW = NT * 16;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h;
    const int img = bid / p.tiles_h;
    constexpr int RPW = TH / 4, NA = 2 * RPW;   // output rows per wave, 16-pixel row tiles per wave
    const int oy0 = ty * TH, ox0 = tx * HALO_TW;
    const int hy0 = oy0 + p.dh_min, hx0 = ox0 + p.dw_min;
    const int npr = p.halo_h * p.halo_w;
    const float* Aimg = p.A + (long long)img * p.imgStrideA;
    const int ly0 = (hy0 >> 1) - 1, lx0 = (hx0 >> 1) - 1;   // low-res origin of the staging tile

    f32x4 acc[NA][NT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // accumulate mode (the per-timestep recurrent convolutions): the tile's current contents are requested NOW, all at once,
    // and consumed in the epilogue — read there one by one (each behind the previous store, which may alias) they cost one
    // exposed memory round trip per 16-byte piece
    constexpr bool PRE = TH == 4;          // (the 8-row tiles have no registers to spare: 216 -> 276 VGPRs with the prefetch)
    f32x4 old[PRE ? NA : 1][PRE ? NT : 1];
    if (PRE && p.accumulate) {
        const int NcP0 = (p.Ncols + 3) & ~3;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int oy = oy0 + RPW * wave + (a >> 1);
            const int ox = ox0 + (a & 1) * 16 + li;
            const bool ok = oy < p.Ho && ox < p.Wo;
            const float* src = p.Out + (long long)img * p.imgStrideO + ((long long)oy * p.Wo + ox) * p.ldO;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = b * 16 + 4 * lg;
                old[PRE ? a : 0][PRE ? b : 0] = (ok && n < NcP0) ? *reinterpret_cast<const f32x4*>(src + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    const int nchunk = (p.C4 + 3) >> 2;
    const bool flat = p.C4 < 4;   // fewer than 16 channels: pack (tap, channel-group) pairs densely into the MFMA k
    for (int ck = 0; ck < nchunk; ++ck) {
        const int kgs = min(4, p.C4 - 4 * ck);
        __syncthreads();  // previous chunk's fragment reads are done
        if (p.upsample) {
            // ---- two-stage staging (the fill, not the MFMA, bounds this layer): (1) the low-resolution source tile
            // (coordinates clamped = bilinear edge clamp) goes to LDS once, (2) the upsampled halo is interpolated
            // LDS -> LDS: 4 LDS reads per slot instead of 4 dependent global loads
            const int nl = p.lr_h * p.lr_w;
            for (int idx = t; idx < nl * kgs; idx += 256) {
                const int kg = idx % kgs;
                const int pix = idx / kgs;
                const int y = pix / p.lr_w, x = pix - y * p.lr_w;
                const int gy = min(max(ly0 + y, 0), p.H - 1), gx = min(max(lx0 + x, 0), p.W - 1);
                lds_lr[pix * 4 + kg] = *reinterpret_cast<const f32x4*>(Aimg + ((long long)gy * p.W + gx) * p.ldA + (4 * ck + kg) * 4);
            }
            __syncthreads();
            for (int idx = t; idx < kgs * npr; idx += 256) {
                const int kg = idx / npr;
                const int pix = idx - kg * npr;
                const int hy = pix / p.halo_w, hx = pix - hy * p.halo_w;
                const int gy = hy0 + hy, gx = hx0 + hx;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((unsigned)gy < (unsigned)p.Hc && (unsigned)gx < (unsigned)p.Wc) {
                    // bilinear x2, half-pixel centres, edge clamp (same arithmetic as wdg_up2_fwd_kernel)
                    const int jh = gy >> 1, jw = gx >> 1;
                    const int h0 = (gy & 1) ? jh : max(jh - 1, 0), h1 = (gy & 1) ? min(jh + 1, p.H - 1) : jh;
                    const int w0 = (gx & 1) ? jw : max(jw - 1, 0), w1 = (gx & 1) ? min(jw + 1, p.W - 1) : jw;
                    const float fh = (gy & 1) ? 0.25f : 0.75f, fw = (gx & 1) ? 0.25f : 0.75f;
                    const int r0 = (h0 - ly0) * p.lr_w, r1 = (h1 - ly0) * p.lr_w, c0 = w0 - lx0, c1 = w1 - lx0;
                    const f32x4 a00 = lds_lr[(r0 + c0) * 4 + kg], a01 = lds_lr[(r0 + c1) * 4 + kg];
                    const f32x4 a10 = lds_lr[(r1 + c0) * 4 + kg], a11 = lds_lr[(r1 + c1) * 4 + kg];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float top = a00[j] + (a01[j] - a00[j]) * fw;
                        const float bot = a10[j] + (a11[j] - a10[j]) * fw;
                        v[j] = top + (bot - top) * fh;
                    }
                }
                lds_a[kg * p.npix + pix] = v;
            }
        } else
        // ---- stage the input halo: lanes run over pixels (conflict-free ds_write_b128); four slots per thread in flight (one
        // memory round trip per batch, not per slot: the small per-timestep launches are bound by these latency chains)
        for (int base = 0; base < kgs * npr; base += 4 * 256) {
            f32x4 v[4];
            int slot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + t;
                const int kg = idx / npr;
                const int pix = idx - kg * npr;
                const int hy = pix / p.halo_w;
                const int hx = pix - hy * p.halo_w;
                const int gy = hy0 + hy, gx = hx0 + hx;
                const bool in = idx < kgs * npr;
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (in && (unsigned)gy < (unsigned)p.Hc && (unsigned)gx < (unsigned)p.Wc)
                    v[u] = *reinterpret_cast<const f32x4*>(Aimg + ((long long)gy * p.W + gx) * p.ldA + (4 * ck + kg) * 4);
                slot[u] = in ? kg * p.npix + pix : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (slot[u] >= 0) lds_a[slot[u]] = v[u];
        }
        // ---- stage this chunk's weights: [tap][kg][n] (kg < kgs only); the tap offsets first, then four weight slots in flight
        if (!WG)
        for (int base = 0; base < p.ntaps * kgs * NW; base += 4 * 256) {
            int woff[4], n_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + t;
                const int n = idx % NW;
                const int r = idx / NW;
                const int kg = r % kgs;
                const int tap = r / kgs;
                n_[u] = (idx < p.ntaps * kgs * NW && n < p.Ncols) ? n : -1;
                woff[u] = (idx < p.ntaps * kgs * NW ? p.taps[tap].z : 0) + (4 * ck + kg) * 4;
            }
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (n_[u] >= 0) v[u] = *reinterpret_cast<const f32x4*>(Bw + (long long)n_[u] * p.ldB + woff[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + t;
                if (idx < p.ntaps * kgs * NW) lds_w[idx] = v[u];
            }
        }
        __syncthreads();
        if (!flat) {
            // ---- all taps from LDS; lane group lg takes channel group lg of the chunk
            const bool kvalid = lg < kgs;
            for (int tap = 0; tap < p.ntaps; ++tap) {
                const int4 e = p.taps[tap];
                const int rowoff = (e.x - p.dh_min) * p.halo_w + (e.y - p.dw_min);
                f32x4 af[NA], bf[NT];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int row = RPW * wave + (a >> 1), col = (a & 1) * 16 + li;
                    af[a] = kvalid ? lds_a[lg * p.npix + row * p.halo_w + col + rowoff] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int n = b * 16 + li;
                    if (WG)
                        bf[b] = (kvalid && n < p.Ncols)
                                    ? *reinterpret_cast<const f32x4*>(Bw + (long long)n * p.ldB + e.z + (4 * ck + lg) * 4)
                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
                    else
                        bf[b] = kvalid ? lds_w[(tap * kgs + lg) * NW + n] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][j], af[a][j], acc[a][b], 0, 0, 0);   // transposed tile
            }
        } else {
            // ---- flattened (tap, channel-group) entries, 4 per MFMA k-group: lane group lg takes entry 4q+lg
            const int nent = p.ntaps * kgs;
            for (int q = 0; q < nent; q += 4) {
                const int ent = q + lg;
                const bool kvalid = ent < nent;
                const int tap = kvalid ? ent / kgs : 0;
                const int kg = ent - tap * kgs;
                const int4 e = p.taps[tap];
                const int rowoff = (e.x - p.dh_min) * p.halo_w + (e.y - p.dw_min);
                f32x4 af[NA], bf[NT];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int row = RPW * wave + (a >> 1), col = (a & 1) * 16 + li;
                    af[a] = kvalid ? lds_a[kg * p.npix + row * p.halo_w + col + rowoff] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const int n = b * 16 + li;
                    if (WG)
                        bf[b] = (kvalid && n < p.Ncols)
                                    ? *reinterpret_cast<const f32x4*>(Bw + (long long)n * p.ldB + e.z + (4 * ck + kg) * 4)
                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
                    else
                        bf[b] = kvalid ? lds_w[ent * NW + n] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][j], af[a][j], acc[a][b], 0, 0, 0);   // transposed tile
            }
        }
    }

    // ---- epilogue: the MFMA operands are swapped (A = weights, B = pixels), so accumulator reg r of lane (li, lg) is
    // output channel b*16 + 4*lg + r of pixel column li: one 16-byte store per lane and tile
    const int NcP = (p.Ncols + 3) & ~3;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int oy = oy0 + RPW * wave + (a >> 1);
        const int ox = ox0 + (a & 1) * 16 + li;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        float* dst = p.Out + (long long)img * p.imgStrideO + ((long long)oy * p.Wo + ox) * p.ldO;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = b * 16 + 4 * lg;
            if (n >= NcP) continue;
            f32x4 v = acc[a][b];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (p.bias && n + r < p.Ncols) v[r] += p.bias[n + r];
                if (p.act) v[r] = wdg_lrelu(v[r], p.slope);
            }
            if (p.accumulate) v += PRE ? old[PRE ? a : 0][PRE ? b : 0] : *reinterpret_cast<const f32x4*>(dst + n);
            *reinterpret_cast<f32x4*>(dst + n) = v;
            if constexpr (NT == 4 || NT == 1) acc[a][b] = v;     // (the cell update below needs the complete pre-activations)
        }
        if constexpr (NT == 1) {
            if (p.lstm_bwd) {
                // ConvLSTM cell backward (the arithmetic of wdg_lstm_bwd, pointwise.hip) for the features this lane holds
                const int F = p.lstm_F;
                const long long pix = (long long)img * p.Ho * p.Wo + (long long)oy * p.Wo + ox;
                auto hs = [](float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); };
                auto hsg = [](float x) { const float v = 0.2f * x + 0.5f; return (v >= 0.f && v <= 1.f) ? 0.2f : 0.f; };
                if ((F & 3) == 0) {
                    // 16 features: this lane's four features are contiguous in every slab -> 16-byte accesses
                    const int f0 = 4 * lg;
                    const float* g = p.gates_t + pix * 4 * F + f0;
                    const f32x4 xi = *reinterpret_cast<const f32x4*>(g), xf = *reinterpret_cast<const f32x4*>(g + F);
                    const f32x4 xc = *reinterpret_cast<const f32x4*>(g + 2 * F), xo = *reinterpret_cast<const f32x4*>(g + 3 * F);
                    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const f32x4 cp = p.c_prev ? *reinterpret_cast<const f32x4*>(p.c_prev + pix * p.ldc + f0) : z4;
                    const f32x4 cc = *reinterpret_cast<const f32x4*>(p.c_cur + pix * p.ldc + f0);
                    const f32x4 dci = *reinterpret_cast<const f32x4*>(p.dc_in + pix * p.ldc + f0);
                    f32x4 di, df, dcc, dob, dcp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gi = hs(xi[r]), gf = hs(xf[r]), gc = wdg_tanh(xc[r]), go = hs(xo[r]);
                        const float tc = wdg_tanh(cc[r]);
                        const float dhv = acc[a][0][r];
                        const float dc = dhv * go * (1.f - tc * tc) + dci[r];
                        di[r] = dc * gc * hsg(xi[r]);
                        df[r] = dc * cp[r] * hsg(xf[r]);
                        dcc[r] = dc * gi * (1.f - gc * gc);
                        dob[r] = dhv * tc * hsg(xo[r]);
                        dcp[r] = dc * gf;
                    }
                    float* dg = p.dgates_out + pix * 4 * F + f0;
                    *reinterpret_cast<f32x4*>(dg) = di;
                    *reinterpret_cast<f32x4*>(dg + F) = df;
                    *reinterpret_cast<f32x4*>(dg + 2 * F) = dcc;
                    *reinterpret_cast<f32x4*>(dg + 3 * F) = dob;
                    if (p.dc_out) *reinterpret_cast<f32x4*>(p.dc_out + pix * p.ldc + f0) = dcp;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = 4 * lg + r;
                        if (f >= F) continue;
                        const float* g = p.gates_t + pix * 4 * F;
                        const float xi = g[f], xf = g[F + f], xc = g[2 * F + f], xo = g[3 * F + f];
                        const float gi = hs(xi), gf = hs(xf), gc = wdg_tanh(xc), go = hs(xo);
                        const float cp = p.c_prev ? p.c_prev[pix * p.ldc + f] : 0.f;
                        const float tc = wdg_tanh(p.c_cur[pix * p.ldc + f]);
                        const float dhv = acc[a][0][r];
                        const float dc = dhv * go * (1.f - tc * tc) + p.dc_in[pix * p.ldc + f];
                        float* dg = p.dgates_out + pix * 4 * F;
                        dg[f] = dc * gc * hsg(xi);
                        dg[F + f] = dc * cp * hsg(xf);
                        dg[2 * F + f] = dc * gi * (1.f - gc * gc);
                        dg[3 * F + f] = dhv * tc * hsg(xo);
                        if (p.dc_out) p.dc_out[pix * p.ldc + f] = dc * gf;
                    }
                }
            }
        }
        if constexpr (NT == 4 || NT == 1) {
            if (p.lstm_F && !p.lstm_bwd) {
                // cell update on the accumulators: Keras hard_sigmoid / tanh, c = f * c_prev + i * c~, h = o * tanh(c)
                // (the same arithmetic as wdg_lstm_fwd, pointwise.hip)
                auto hs = [](float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); };
                const long long pix = (long long)img * p.Ho * p.Wo + (long long)oy * p.Wo + ox;
                if constexpr (NT == 4) {
                    // 16 features: column tile b is gate b, this lane holds features 4*lg .. 4*lg+3 of its pixel
                    const f32x4 cp = *reinterpret_cast<const f32x4*>(p.c_prev + pix * p.ldc + 4 * lg);
                    f32x4 cn, hn;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cn[r] = hs(acc[a][1][r]) * cp[r] + hs(acc[a][0][r]) * wdg_tanh(acc[a][2][r]);
                        hn[r] = hs(acc[a][3][r]) * wdg_tanh(cn[r]);
                    }
                    *reinterpret_cast<f32x4*>(p.c_out + pix * p.ldc + 4 * lg) = cn;
                    *reinterpret_cast<f32x4*>(p.h_out + pix * p.ldh + 4 * lg) = hn;
                } else {
                    // 2 features: columns (i0 i1 f0 f1) in lane group 0, (c~0 c~1 o0 o1) in lane group 1 of the same pixel
                    f32x4 other;
#pragma unroll
                    for (int r = 0; r < 4; ++r) other[r] = __shfl(acc[a][0][r], li + 16, 64);
                    if (lg == 0) {
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            const float cn = hs(acc[a][0][2 + f]) * p.c_prev[pix * p.ldc + f] + hs(acc[a][0][f]) * wdg_tanh(other[f]);
                            p.c_out[pix * p.ldc + f] = cn;
                            p.h_out[pix * p.ldh + f] = hs(other[2 + f]) * wdg_tanh(cn);
                        }
                    }
                }
            }
        }
    }
}

// ---- persistent single-chunk variant ----------------------------------------------------------------
// 3x3 layers with exactly 16 (padded) input channels and <= 16 outputs on large maps (the discriminator's 16->16
// full-resolution conv and its data gradient, the generator's 16->2 output conv) are latency-bound in the kernel
// above (PMC: 42 % of wave cycles waiting, 1.9 TB/s): one tile per block, the halo staged by dependent loads between
// two barriers.  Here a block walks tiles with a grid stride, the NEXT tile's halo is fetched into registers while
// the current one is multiplied (one barrier pair per tile, no exposed load latency), and the nine taps' weight
// fragments stay in registers for the block's whole life.
// LN: the LayerNorm epilogue as a separate instantiation — in one kernel its registers cost the plain launches (data gradient,
// 16 -> 2 output conv) their fourth wave per SIMD (110 -> 142 registers: 104 -> ~125 us per launch)
// STG: how the 256 threads share the halo's 16-byte pieces.  0: four consecutive lanes = the 64 contiguous bytes of one pixel,
// LDS slot = kg * npix + pixel — the staging stores of a quad land on ONE 16-byte bank column (4-way conflict on all six
// ds_write_b128 per tile: the whole of this kernel's SQ_LDS_BANK_CONFLICT, 0.31-0.36 of its LDS cycles).  1: sixteen consecutive
// lanes still cover the 256 contiguous bytes of four pixels, but as (kg 0 | kg 2 | kg 1 | kg 3) x four pixels, and the channel
// groups 2, 3 are rotated by four slots (slot = kg * npix + pixel + 4 * (kg >> 1)): every 8-lane store group then hits eight
// distinct columns, and the fragment reads stay conflict-free (their 16-lane groups pair kg 0 with 1 and 2 with 3, which keep
// equal rotations).
template <int TAPS, bool LN = false, int STG = 0>
__global__ void __launch_bounds__(256) wdg_conv_halo1_kernel(const WdgHalo p, const float* __restrict__ Bw) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem1[];
    f32x4* lds_a = smem1;   // [4][npix] (+ 4 slots of rotation slack)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int npr = p.halo_h * p.halo_w;
    constexpr int NLMAX = 6;                      // (8+2)*(32+2)*4 / 256 = 5.3 slots per thread for 3x3
    const int nslots = 4 * npr;

    // weight fragments of all taps in LDS for the block's whole life: slot [tap][lg][li]
    f32x4* lds_w = lds_a + 4 * p.npix + 8;       // (slot 4 npix + 4: where the staging slots beyond the halo are written, never read)
    for (int idx = t; idx < TAPS * 64; idx += 256) {
        const int tap = idx >> 6, l = idx & 63;
        const int wli = l & 15, wlg = l >> 4;
        lds_w[idx] = wli < p.Ncols ? *reinterpret_cast<const f32x4*>(Bw + (long long)wli * p.ldB + p.taps[tap].z + wlg * 4)
                                   : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int rowoff[TAPS];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int4 e = p.taps[tap];
        rowoff[tap] = (e.x - p.dh_min) * p.halo_w + (e.y - p.dw_min);
    }
    const int ntiles = p.n_img * p.tiles_h * p.tiles_w;
    f32x4 rs[NLMAX];
    // slot geometry is tile independent: halo row / column and the in-image element offset of every staging slot
    int shy[NLMAX], shx[NLMAX], soff[NLMAX], sslot[NLMAX];
#pragma unroll
    for (int i = 0; i < NLMAX; ++i) {
        const int idx = t + 256 * i;
        int kg, pix;
        if constexpr (STG == 0) {
            kg = idx & 3; pix = idx >> 2;           // 4 consecutive lanes = the 64 contiguous bytes of one pixel
        } else {
            const int j = idx & 15;
            pix = (idx >> 4) * 4 + (j & 3);
            kg = ((j >> 2) & 1) * 2 + (j >> 3);
        }
        const bool on = STG == 0 ? idx < nslots : (pix < npr && idx < 4 * ((npr + 3) & ~3));
        shy[i] = on ? pix / p.halo_w : (1 << 28);
        shx[i] = pix - (pix / p.halo_w) * p.halo_w;
        soff[i] = (((pix / p.halo_w) * p.W + shx[i]) * p.ldA + kg * 4) * 4;     // bytes
        sslot[i] = on ? kg * p.npix + pix + (STG ? (kg >> 1) * 4 : 0) : 4 * p.npix + 4;
    }
    auto load_tile = [&](int tile) {
        const int img = tile / (p.tiles_h * p.tiles_w);
        const int rem = tile - img * (p.tiles_h * p.tiles_w);
        const int ty = rem / p.tiles_w, tx = rem - ty * p.tiles_w;
        const int hy0 = ty * HALO_TH + p.dh_min, hx0 = tx * HALO_TW + p.dw_min;
        // branch-free: a buffer descriptor per image, padding / unused slots get offset 0x80000000 (beyond it -> zeros) by
        // arithmetic on the signs of the range checks.  Written as `if (inside) v = load` the six requests became six exec-masked
        // blocks and the compiler waited for ALL of them at their join — in front of the MFMAs they were meant to travel under
        const wdg_srd srdA = wdg_make_srd(p.A + (long long)img * p.imgStrideA);
        const int org = (hy0 * p.W + hx0) * p.ldA * 4;
#pragma unroll
        for (int i = 0; i < NLMAX; ++i) {
            const int gy = hy0 + shy[i], gx = hx0 + shx[i];
            const unsigned neg = (unsigned)((gy | (p.Hc - 1 - gy) | gx | (p.Wc - 1 - gx)) >> 31);
            rs[i] = wdg_buffer_load_f32x4(srdA, ((unsigned)(org + soff[i]) & ~neg) | (neg & 0x80000000u));
        }
    };
    // epilogue constants once per workgroup (inside the tile loop they were reloaded per tile under per-element branches, and
    // their wait also waited for the next tile's halo: the wait counter is in order)
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f}, gm = bv, bt = bv;
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = p.bias[4 * lg + r < p.Ncols ? 4 * lg + r : 0];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = 4 * lg + r < p.Ncols ? bv[r] : 0.f;
    }
    if constexpr (LN) {
        gm = *reinterpret_cast<const f32x4*>(p.ln_gamma + 4 * lg);
        bt = *reinterpret_cast<const f32x4*>(p.ln_beta + 4 * lg);
    }
    if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < NLMAX; ++i) {
            const int idx = t + 256 * i;
            // slot = kg*npix + pixel (npix = 0 mod 16): conflict-free fragment reads (36 per tile); the 6 staging writes
            // per tile are 4-way conflicted, the price of coalesced 64-byte-per-pixel global loads (an XOR swizzle that
            // fixed the writes made 30 % of the read cycles conflicts: profiles/r01aq_pmc_summary.csv)
            (void)idx;
            lds_a[sslot[i]] = rs[i];       // (unconditional: under `if (slot >= 0)` the compiler sinks the slot's LOAD into the branch)
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
        f32x4 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int pbase = 2 * wave * p.halo_w + li;
        const f32x4* lds_g = lds_a + lg * p.npix + (STG ? (lg >> 1) * 4 : 0);
        // fragments of tap+1 are read while tap is multiplied; the scheduling barrier per tap keeps the compiler from
        // hoisting all 36 fragment reads (144 registers, 2 waves/SIMD) in front of the MFMAs
        f32x4 af[2][4], bfr[2];
        auto read_tap = [&](int tap, f32x4 (&a4)[4], f32x4& b1) {
#pragma unroll
            for (int a = 0; a < 4; ++a) a4[a] = lds_g[pbase + (a >> 1) * p.halo_w + (a & 1) * 16 + rowoff[tap]];
            b1 = lds_w[tap * 64 + lane];
        };
        read_tap(0, af[0], bfr[0]);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap + 1 < TAPS) read_tap(tap + 1, af[(tap + 1) & 1], bfr[(tap + 1) & 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    // operands swapped (A = weights, B = pixels): the accumulator is the TRANSPOSED tile, lane (li, lg)
                    // holds channels 4lg..4lg+3 of pixel li -> one 16-byte store per row tile instead of four 4-byte ones
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[tap & 1][j], af[tap & 1][a][j], acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        int b = tile;
        const int tx = b % p.tiles_w;
        b /= p.tiles_w;
        const int ty = b % p.tiles_h;
        const int img = b / p.tiles_h;
        if (4 * lg < ((p.Ncols + 3) & ~3)) {
            if constexpr (LN) {
                // conv -> bias -> LeakyReLU -> LayerNormalization over the 16 channels of a pixel: they sit in the four lanes
                // li, li + 16, li + 32, li + 48 (four registers each), so the two reductions (sum, centred sum of squares — the
                // two-pass form of wdg_ln_fwd) are two xor-shuffles each; y and z leave as 16-byte stores, no pass re-reads y
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int oy = ty * HALO_TH + 2 * wave + (a >> 1);
                    const int ox = tx * HALO_TW + (a & 1) * 16 + li;
                    f32x4 v = acc[a] + bv;
                    if (p.act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], p.slope);
                    }
                    float s_ = (v[0] + v[1]) + (v[2] + v[3]);
                    s_ += __shfl_xor(s_, 16, 64);
                    s_ += __shfl_xor(s_, 32, 64);
                    const float mean = s_ * (1.f / 16.f);
                    const f32x4 d = v - mean;
                    float q_ = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                    q_ += __shfl_xor(q_, 16, 64);
                    q_ += __shfl_xor(q_, 32, 64);
                    const float rstd = 1.f / sqrtf(q_ * (1.f / 16.f) + p.ln_eps);
                    if (oy >= p.Ho || ox >= p.Wo) continue;
                    const long long pix = (long long)oy * p.Wo + ox;
                    *reinterpret_cast<f32x4*>(p.Out + (long long)img * p.imgStrideO + pix * p.ldO + 4 * lg) = v;
                    f32x4 z;
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = d[r] * rstd * gm[r] + bt[r];
                    *reinterpret_cast<f32x4*>(p.Out2 + (long long)img * p.imgStrideO2 + pix * p.ldO2 + 4 * lg) = z;
                    if (p.mean_rstd && lg == 0) {
                        const long long pi = (long long)img * p.Ho * p.Wo + pix;
                        p.mean_rstd[2 * pi] = mean;
                        p.mean_rstd[2 * pi + 1] = rstd;
                    }
                }
            } else {
            // accumulate: the four previous values requested together (from clamped, always valid addresses — per-element
            // `if (inside) v += *dst` is four load - wait - store sequences)
            f32x4 prev[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) prev[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.accumulate) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int oy = min(ty * HALO_TH + 2 * wave + (a >> 1), p.Ho - 1);
                    const int ox = min(tx * HALO_TW + (a & 1) * 16 + li, p.Wo - 1);
                    prev[a] = *reinterpret_cast<const f32x4*>(p.Out + (long long)img * p.imgStrideO + ((long long)oy * p.Wo + ox) * p.ldO + 4 * lg);
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy = ty * HALO_TH + 2 * wave + (a >> 1);
                const int ox = tx * HALO_TW + (a & 1) * 16 + li;
                if (oy >= p.Ho || ox >= p.Wo) continue;
                f32x4* dst = reinterpret_cast<f32x4*>(p.Out + (long long)img * p.imgStrideO + ((long long)oy * p.Wo + ox) * p.ldO + 4 * lg);
                f32x4 v = acc[a] + bv;
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], p.slope);
                }
                *dst = v + prev[a];
            }
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------
static int halo_nt(int ncols) { return ncols <= 16 ? 1 : ncols <= 32 ? 2 : ncols <= 64 ? 4 : 0; }

static int g_halo_wg_small = 0;   // weight path of the 4-row (small-launch) tiles: 0 LDS, 1 global, -1 follow g_halo_wg
static int g_halo_wg = 1;   // wdg_set_tuning("halo_weights_global", 0/1)
void wdg_halo_set_wg(int v) { g_halo_wg = v != 0; g_halo_wg_small = v >= 2 ? -1 : (v != 0); }   // 0 / 1: both tile sizes; 2: 8-row tiles global, 4-row tiles follow -> used by A/B runs
static int g_halo_persistent = 1;   // wdg_set_tuning("halo_persistent", 0/1)
static int g_halo_th4 = 1;          // wdg_set_tuning("halo_th4", 0/1): 4-row tiles for launches with fewer than two tiles per CU
void wdg_halo_set_th4(int v) { g_halo_th4 = v != 0; }
static int g_halo_max_cin = 64;     // reduction channels above which conv_fwd / conv_dgrad do not take the halo kernel (0 = no limit)
void wdg_halo_set_max_cin(int v) { g_halo_max_cin = v; }
static int g_halo1_bpc = 4;         // resident workgroups per CU of the persistent kernel (126 registers -> 4 waves per SIMD; measured 2: 122, 3: 113, 4: 111, 5: 127 us)
void wdg_halo_set_persistent(int v) { g_halo_persistent = v != 0; if (v > 1) g_halo1_bpc = v; }
static int g_halo1_stage = 1;       // wdg_set_tuning("halo1_stage", 0/1): staging assignment of the persistent kernel (STG)
void wdg_halo_set_stage(int v) { g_halo1_stage = v != 0; }

static size_t halo_lds_bytes(int kh, int kw, int nt, int wg = 0, int upsample = 0, int th = HALO_TH) {
    const int hh = th + kh - 1, hw = HALO_TW + kw - 1;
    const int npix = wdg_round_up(hh * hw, 16);
    const int lr = upsample ? (hh / 2 + 3) * (hw / 2 + 3) * 4 : 0;
    return (size_t)(4 * npix + (wg ? 0 : kh * kw * 4 * nt * 16) + lr) * sizeof(f32x4);
}

int wdg_halo_plan_init(wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    pl->halo_fwd_nt = pl->halo_dgrad_nt = 0;
    pl->halo_auto = pl->halo_auto_fwd = pl->halo_auto_dgrad = 0;
    if (g.stride != 1 || g.kh > 5 || g.kw > 5) return WDG_OK;
    // the conv entry points switch to the halo kernel only where the im2col-free gather is
    // traffic-bound (few output channels, large maps); wdg_upconv_fwd uses it at any size
    const long long pixels = (long long)g.n_img * g.Ho * g.Wo;
    pl->halo_auto = pixels >= 65536;
    int nf = halo_nt(g.Cout), nd = halo_nt(g.Cin);
    // deep reductions (3x3 128 -> 64) are MFMA bound, not traffic bound: there the implicit GEMM's 128x64 tile wins
    // (1.09 -> 0.79 ms per step for that layer); the kernels stay available to wdg_upconv_fwd / explicit callers
    pl->halo_auto_fwd = pl->halo_auto && !(g_halo_max_cin > 0 && g.Cin > g_halo_max_cin);
    pl->halo_auto_dgrad = pl->halo_auto && !(g_halo_max_cin > 0 && g.Cout > g_halo_max_cin);
    if (nf && halo_lds_bytes(g.kh, g.kw, nf) > 64 * 1024) nf = 0;
    if (nd && halo_lds_bytes(g.kh, g.kw, nd) > 64 * 1024) nd = 0;
    std::vector<int4> tf, td;
    for (int th = 0; th < g.kh; ++th)
        for (int tw = 0; tw < g.kw; ++tw) {
            const int tap = th * g.kw + tw;
            tf.push_back((int4){th - g.pad_h, tw - g.pad_w, tap * pl->Cin_p, 0});
            td.push_back((int4){g.pad_h - th, g.pad_w - tw, tap * g.Cin * pl->Cout_p, 0});
        }
    if (nf) {
        WDG_HIP(hipMalloc((void**)&pl->d_taps_fwd, tf.size() * sizeof(int4)));
        WDG_HIP(hipMemcpy(pl->d_taps_fwd, tf.data(), tf.size() * sizeof(int4), hipMemcpyHostToDevice));
        pl->halo_fwd_nt = nf;
    }
    if (nd) {
        WDG_HIP(hipMalloc((void**)&pl->d_taps_dgrad, td.size() * sizeof(int4)));
        WDG_HIP(hipMemcpy(pl->d_taps_dgrad, td.data(), td.size() * sizeof(int4), hipMemcpyHostToDevice));
        pl->halo_dgrad_nt = nd;
    }
    return WDG_OK;
}

void wdg_halo_plan_free(wdg_conv_plan* pl) {
    if (pl->d_taps_fwd) (void)hipFree(pl->d_taps_fwd);
    if (pl->d_taps_dgrad) (void)hipFree(pl->d_taps_dgrad);
}

// forward conv of this plan runs in the persistent 3x3 kernel with exactly 16 output channels: its epilogue can normalise them
static int g_halo_ln = 1;           // wdg_set_tuning("halo_ln", 0/1)
void wdg_halo_set_ln(int v) { g_halo_ln = v != 0; }
bool wdg_halo_ln_eligible(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    return g_halo_ln && pl->halo_auto_fwd && pl->halo_fwd_nt == 1 && g_halo_persistent && pl->Cin_p == 16 && pl->taps == 9 && g.kh == 3 &&
           g.Cout == 16;
}

int wdg_halo_launch(const wdg_conv_plan* pl, bool dgrad, const float* A, int ldA, long long imgStrideA, int upsample,
                    const float* Bw, const float* bias, float* Out, int act, float slope, int accumulate,
                    hipStream_t st, const WdgHaloLstm* cell, const WdgHaloLn* ln) {
    const wdg_conv_geom& g = pl->g;
    WdgHalo p;
    memset(&p, 0, sizeof(p));
    if (ln) {
        p.Out2 = ln->z; p.ldO2 = ln->ldz; p.imgStrideO2 = ln->img_stride_z; p.ln_gamma = ln->gamma; p.ln_beta = ln->beta;
        p.ln_eps = ln->eps; p.mean_rstd = ln->mean_rstd;
    }
    if (cell) {
        p.lstm_F = cell->F; p.c_prev = cell->c_prev; p.c_out = cell->c_out; p.h_out = cell->h_out; p.ldc = cell->ldc; p.ldh = cell->ldh;
        p.lstm_bwd = cell->bwd; p.gates_t = cell->gates_t; p.c_cur = cell->c_cur; p.dc_in = cell->dc_in;
        p.dgates_out = cell->dgates_out; p.dc_out = cell->dc_out;
    }
    p.A = A; p.B = Bw; p.Out = Out; p.bias = bias;
    p.n_img = g.n_img; p.ldA = ldA; p.imgStrideA = imgStrideA;
    p.ntaps = pl->taps;
    p.act = act; p.slope = slope; p.accumulate = accumulate; p.upsample = upsample;
    int nt;
    if (!dgrad) {
        p.taps = pl->d_taps_fwd;
        p.Hc = g.H; p.Wc = g.W;
        p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy; p.imgStrideO = g.img_stride_y;
        p.C4 = pl->Cin_p / 4; p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p;
        p.dh_min = -g.pad_h; p.dw_min = -g.pad_w;
        nt = pl->halo_fwd_nt;
    } else {
        p.taps = pl->d_taps_dgrad;
        p.Hc = g.Ho; p.Wc = g.Wo;
        p.Ho = g.H; p.Wo = g.W; p.ldO = g.ldx; p.imgStrideO = g.img_stride_x;
        p.C4 = pl->Cout_p / 4; p.Ncols = g.Cin; p.ldB = pl->Cout_p;
        p.dh_min = g.pad_h - (g.kh - 1); p.dw_min = g.pad_w - (g.kw - 1);
        nt = pl->halo_dgrad_nt;
    }
    if (upsample) {
        if ((p.Hc & 1) || (p.Wc & 1)) {
            wdg_set_error("halo: upsample mode needs even conv-input dims");
            return WDG_ERR_ARG;
        }
        p.H = p.Hc / 2; p.W = p.Wc / 2;
    } else {
        p.H = p.Hc; p.W = p.Wc;
    }
    if (!nt) {
        wdg_set_error("halo: plan is not eligible for the halo-tile kernel");
        return WDG_ERR_ARG;
    }
    // small launches (fewer than two 8-row tiles per CU) use 4-row tiles: twice the workgroups (not the upsampling form,
    // whose low-resolution staging tile is sized for 8 rows, nor the persistent 3x3 kernel below)
    const bool persistent1 = g_halo_persistent && !upsample && !cell && nt == 1 && p.C4 == 4 && pl->taps == 9 && g.kh == 3;
    const long long tiles8 = (long long)g.n_img * ((p.Ho + HALO_TH - 1) / HALO_TH) * ((p.Wo + HALO_TW - 1) / HALO_TW);
    const int th = (g_halo_th4 && !upsample && !persistent1 && tiles8 < 2LL * pl->cus) ? 4 : HALO_TH;
    p.halo_h = th + g.kh - 1; p.halo_w = HALO_TW + g.kw - 1;
    p.npix = wdg_round_up(p.halo_h * p.halo_w, 16);
    p.tiles_h = (p.Ho + th - 1) / th;
    p.tiles_w = (p.Wo + HALO_TW - 1) / HALO_TW;
    // weights: read from global memory inside the tap loop (big launches: many workgroups hide the latency, LDS stays free), or
    // staged into LDS once per workgroup (small launches — the per-timestep recurrent steps: -1 % of the T = 24 step);
    // upsample mode: the low-res staging tile takes the LDS of the weight stage
    const int wg = upsample ? 1 : (th == 4 && g_halo_wg_small >= 0) ? g_halo_wg_small : g_halo_wg;
    p.lr_h = p.halo_h / 2 + 3; p.lr_w = p.halo_w / 2 + 3;
    const size_t lds = halo_lds_bytes(g.kh, g.kw, nt, wg, upsample, th);
    dim3 grid((unsigned)((long long)g.n_img * p.tiles_h * p.tiles_w)), block(256);
    if (ln && !persistent1) {
        wdg_set_error("halo: the LayerNorm epilogue exists in the persistent 3x3 kernel only (wdg_halo_ln_eligible)");
        return WDG_ERR_ARG;
    }
    if (persistent1) {
        // latency-bound thin 3x3 layer: persistent blocks with next-tile prefetch (4 resident blocks per CU)
        const size_t lds1 = ((size_t)4 * p.npix + 8 + 9 * 64) * sizeof(f32x4);
        if ((long long)p.Hc * p.Wc * p.ldA * 4 >= (1LL << 31)) {
            wdg_set_error("halo: an image of the input must stay below 2 GiB (buffer descriptor per image)");
            return WDG_ERR_ARG;
        }
        const unsigned nb = (unsigned)std::min<long long>((long long)grid.x, (long long)pl->cus * (ln ? std::min(g_halo1_bpc, 3) : g_halo1_bpc));
        if (ln && g_halo1_stage)
            hipLaunchKernelGGL((wdg_conv_halo1_kernel<9, true, 1>), dim3(nb), block, lds1, st, p, Bw);
        else if (ln)
            hipLaunchKernelGGL((wdg_conv_halo1_kernel<9, true>), dim3(nb), block, lds1, st, p, Bw);
        else if (g_halo1_stage)
            hipLaunchKernelGGL((wdg_conv_halo1_kernel<9, false, 1>), dim3(nb), block, lds1, st, p, Bw);
        else
            hipLaunchKernelGGL((wdg_conv_halo1_kernel<9>), dim3(nb), block, lds1, st, p, Bw);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
#define WDG_HALO_CASE(NT_)                                                                             \
    if (nt == NT_) {                                                                                   \
        if (th == 4 && wg) hipLaunchKernelGGL((wdg_conv_halo_kernel<NT_, 1, 4>), grid, block, lds, st, p, Bw);   \
        else if (th == 4) hipLaunchKernelGGL((wdg_conv_halo_kernel<NT_, 0, 4>), grid, block, lds, st, p, Bw);    \
        else if (wg) hipLaunchKernelGGL((wdg_conv_halo_kernel<NT_, 1>), grid, block, lds, st, p, Bw);  \
        else hipLaunchKernelGGL((wdg_conv_halo_kernel<NT_, 0>), grid, block, lds, st, p, Bw);          \
    }
    WDG_HALO_CASE(1)
    WDG_HALO_CASE(2)
    WDG_HALO_CASE(4)
#undef WDG_HALO_CASE
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// Fused UpSampling2D(2,'bilinear') + Conv2DTranspose forward (models.py:62-64): the plan describes the
// transposed conv on the UPSAMPLED grid (its conv-output side is 2H x 2W); `x_low` is the H x W tensor.
extern "C" int wdg_upconv_fwd(const wdg_conv_plan* pl, const float* x_low, int ld_low, int64_t img_stride_low,
                              const float* wD, const float* bias, float* y, int act, float slope,
                              wdg_stream stream) {
    WDG_CHECK_ARG(pl && x_low && wD && y, "null argument");
    WDG_CHECK_ARG(((uintptr_t)x_low & 15) == 0 && ld_low % 4 == 0, "x_low must be 16-byte aligned, ld % 4 == 0");
    WDG_CHECK_ARG(pl->halo_dgrad_nt != 0, "plan not eligible (needs stride 1, k <= 5, Cin <= 64)");
    return wdg_halo_launch(pl, true, x_low, ld_low, img_stride_low, 1, wD, bias, y, act, slope, 0,
                           (hipStream_t)stream);
}

static int g_lstm_step_fused = 7;       // bit 0: forward step, bit 1: backward step, bit 2: the two-feature layer on its pixel-per-thread kernels
// ---- two-feature ConvLSTM recurrent steps without the matrix pipe (the discriminator's first ConvLSTM, models.py:93: 2 -> 2
// features at full resolution).  The work per timestep is tiny (144 multiply-adds per pixel); through the halo-tile MFMA
// kernel a step costs ~27 us of staging / barrier / fragment latency.  Here a thread owns a pixel: nine 8-byte neighbour
// loads, the weights as wave-uniform scalars, the cell update in registers — two dependent memory round trips per launch.
__global__ void __launch_bounds__(256) wdg_convlstm2_fwd_kernel(const float* __restrict__ h_prev, int ldx, long long imgStrideX,
                                                                const float* __restrict__ wF, float* gates,
                                                                const float* __restrict__ c_prev, float* c_out, int ldc,
                                                                float* h_out, int ldh, int n_img, int H, int W) {
    wdg_convlstm2_fwd_body((long long)blockIdx.x * 256 + threadIdx.x, h_prev, ldx, imgStrideX, wF, gates, c_prev, c_out, ldc, h_out, ldh, n_img, H, W);
}

__global__ void __launch_bounds__(256) wdg_convlstm2_bwd_kernel(const float* __restrict__ dg_next, const float* __restrict__ wD,
                                                                float* dh_prev, int ldx, long long imgStrideX,
                                                                const float* __restrict__ gates_t, const float* __restrict__ c_prev,
                                                                const float* __restrict__ c_cur, const float* __restrict__ dc_in,
                                                                float* dgates_out, float* dc_out, int ldc, int n_img, int H, int W) {
    wdg_convlstm2_bwd_body((long long)blockIdx.x * 256 + threadIdx.x, dg_next, wD, dh_prev, ldx, imgStrideX, gates_t, c_prev, c_cur, dc_in,
                           dgates_out, dc_out, ldc, n_img, H, W);
}
bool wdg_lstm2_geom(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    return (g_lstm_step_fused & 4) && g.Cin == 2 && g.Cout == 8 && g.kh == 3 && g.kw == 3 && g.stride == 1 && g.pad_h == 1 && g.pad_w == 1 &&
           g.H == g.Ho && g.W == g.Wo && g.ldx >= 2 && g.ldx % 2 == 0 && pl->Cin_p == 4 && pl->Cout_p == 8;
}

// ---- ConvLSTM2D recurrent step in one launch (models.py:93,101 at n_timesteps > 1): recurrent 3x3 convolution of h_{t-1}
// accumulated onto the input part of the gates (left in `gates` as the pre-activations the backward pass reads) + cell update.
void wdg_halo_set_lstm_fused(int v) { g_lstm_step_fused = v; }
extern "C" int wdg_convlstm_step_supported(const wdg_conv_plan* pl, int F) {
    if (!pl || !(g_lstm_step_fused & 1)) return 0;
    const wdg_conv_geom& g = pl->g;
    if (g.Cout != 4 * F || g.stride != 1 || g.ldy != 4 * F || g.img_stride_y != (int64_t)g.Ho * g.Wo * 4 * F) return 0;
    return (F == 16 && pl->halo_fwd_nt == 4) || (F == 2 && pl->halo_fwd_nt == 1);
}
extern "C" int wdg_convlstm_step(const wdg_conv_plan* pl, const float* h_prev, const float* wF, float* gates, const float* c_prev,
                                 float* c_out, int ldc, float* h_out, int ldh, int F, wdg_stream stream) {
    WDG_CHECK_ARG(pl && h_prev && wF && gates && c_prev && c_out && h_out && wdg_convlstm_step_supported(pl, F), "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= F && ldh >= F && (F != 16 || (ldc % 4 == 0 && ldh % 4 == 0 && (((uintptr_t)c_prev | (uintptr_t)c_out | (uintptr_t)h_out) & 15) == 0)),
                  "bad strides / alignment");
    if (F == 2 && wdg_lstm2_geom(pl) && (((uintptr_t)h_prev | (uintptr_t)gates) & 15) == 0) {
        const wdg_conv_geom& g = pl->g;
        const long long P = (long long)g.n_img * g.H * g.W;
        hipLaunchKernelGGL(wdg_convlstm2_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h_prev, g.ldx,
                           (long long)g.img_stride_x, wF, gates, c_prev, c_out, ldc, h_out, ldh, g.n_img, g.H, g.W);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
    WdgHaloLstm cell;
    memset(&cell, 0, sizeof(cell));
    cell.F = F; cell.ldc = ldc; cell.ldh = ldh; cell.c_prev = c_prev; cell.c_out = c_out; cell.h_out = h_out;
    return wdg_halo_launch(pl, false, h_prev, pl->g.ldx, pl->g.img_stride_x, 0, wF, nullptr, gates, 0, 0.f, 1, (hipStream_t)stream, &cell);
}

// backward counterpart: dh_prev += conv_transpose(dgates_next, wD) — now the complete gradient of h_{t-1} — followed by the cell
// backward of timestep t-1 in the epilogue (gates_t, c_prev (NULL at t-1 = 0), c_cur, dc_in -> dgates_out, dc_out (NULL: not needed))
extern "C" int wdg_convlstm_bwd_step_supported(const wdg_conv_plan* pl, int F) {
    if (!pl || !(g_lstm_step_fused & 2)) return 0;
    const wdg_conv_geom& g = pl->g;
    if (g.Cout != 4 * F || g.Cin != F || g.stride != 1 || g.ldy != 4 * F || g.img_stride_y != (int64_t)g.Ho * g.Wo * 4 * F ||
        g.H != g.Ho || g.W != g.Wo)
        return 0;
    return (F == 16 || F == 2) && pl->halo_dgrad_nt == 1;
}
extern "C" int wdg_convlstm_bwd_step(const wdg_conv_plan* pl, const float* dgates_next, const float* wD, float* dh_prev,
                                     const float* gates_t, const float* c_prev, const float* c_cur, const float* dc_in,
                                     float* dgates_out, float* dc_out, int ldc, int F, wdg_stream stream) {
    WDG_CHECK_ARG(pl && dgates_next && wD && dh_prev && gates_t && c_cur && dc_in && dgates_out && wdg_convlstm_bwd_step_supported(pl, F),
                  "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= F, "bad stride");
    if (F == 2 && wdg_lstm2_geom(pl) && (((uintptr_t)dgates_next | (uintptr_t)dgates_out) & 15) == 0) {
        const wdg_conv_geom& g = pl->g;
        const long long P = (long long)g.n_img * g.H * g.W;
        hipLaunchKernelGGL(wdg_convlstm2_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dgates_next, wD,
                           dh_prev, g.ldx, (long long)g.img_stride_x, gates_t, c_prev, c_cur, dc_in, dgates_out, dc_out, ldc, g.n_img, g.H, g.W);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
    WdgHaloLstm cell;
    memset(&cell, 0, sizeof(cell));
    cell.F = F; cell.ldc = ldc; cell.bwd = 1; cell.c_prev = c_prev; cell.gates_t = gates_t; cell.c_cur = c_cur; cell.dc_in = dc_in;
    cell.dgates_out = dgates_out; cell.dc_out = dc_out;
    return wdg_halo_launch(pl, true, dgates_next, pl->g.ldy, pl->g.img_stride_y, 0, wD, nullptr, dh_prev, 0, 0.f, 1, (hipStream_t)stream, &cell);
}
