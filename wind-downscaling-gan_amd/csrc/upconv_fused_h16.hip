// upconv_fused_h16.hip — inference precision: UpSampling2D(2, 'bilinear') + Conv2DTranspose(5 x 5) of the generator
// (/root/reference/src/downscaling/gan/models.py:62-64) in ONE kernel, column form, the column tensor never leaving the CU.
//
// The column form (upconv_col.hip) is  z[r, (t, o)] = sum_c x[r, c] w[t][o][c]  on the low-resolution grid (a 1 x 1 GEMM with
// 25 * C columns) followed by  y[p, o] = act(bias + sum k_r(a) k_r(b) z[r, (t, o)])  over the (r, t, a, b) with
// 2 r - 1 + (a, b) + t - 2 = p.  As two launches, z — 400 floats per low-resolution pixel, 1.4 GB for a 16-tile group of the
// shipped generator — is written once and read once with a 2.25 x halo: both launches run at the memory system's rate
// (0.48 + 0.54 ms of the group's 3.5 ms; the 16-bit MFMA work of the GEMM is ~0.1 ms).  A 16-bit z was tried twice in
// round 3 and lost to its access granularity.  Here a workgroup owns an 8 x 8 low-resolution tile (16 x 16 output pixels):
//   * the 12 x 12 window of x (K = 160 channels) is staged ONCE, rounded to the operand format, as [k-octet][pixel] 16-byte
//     slots (pixel index XOR-swizzled by the octet: conflict-free staging stores and fragment reads);
//   * per tap row ty the weights' 80 x K slice streams into LDS (a straight copy of the layer's weight tensor viewed as
//     [25 * C][K]), the 144 x 80 slice of z is formed by 45 MFMA tiles dealt round-robin to the four waves and written to the
//     LDS window the gather passes of upconv_col.hip read (horizontal pass -> H, vertical pass -> the thread's output pixel);
//   * bias, LeakyReLU and the inference BatchNorm affine in the epilogue, as in wdg_upconv_gather.
// Same rounding points as the two-launch route (x and W rounded to nearest even, fp32 accumulation, fp32 z and interpolation);
// the GEMM is computed on the window, i.e. 2.25 x the multiply-adds — at 16-bit MFMA rates that is cheaper than z's traffic.
#include "common.h"
#include "h16.h"
#include <algorithm>

namespace {
constexpr int F_TS = 8;                      // tile edge on the low-res grid
constexpr int F_ZW = F_TS + 4;               // window edge (12)
constexpr int F_NPX = F_ZW * F_ZW;           // 144 window pixels = 9 MFMA pixel tiles
constexpr int F_CQ = 4;                      // output channels / 4 (C = 16)
constexpr int F_PX = 5 * F_CQ;               // float4 per window pixel and tap row
constexpr int F_HP = F_CQ + 1;               // H pixel pitch (see wdg_upconv_gather_kernel)
constexpr int F_NCOL = 5 * 4 * F_CQ;         // 80 columns of z per tap row
constexpr int F_NT = 512;                    // threads: 133 KB of LDS leave ONE workgroup per CU — eight waves (two per SIMD) give its phases
                                             // (staging, MFMA tiles, the two gather passes, four barriers per tap row) something to overlap with
constexpr int F_NW = F_NT / 64;

__device__ __forceinline__ float f_coef(int r, int a, int Hl) {
    const int q = 2 * r - 1 + a;
    if ((unsigned)q >= (unsigned)(2 * Hl)) return 0.f;
    float c = (a == 0 || a == 3) ? 0.25f : 0.75f;
    if (q == 0 || q == 2 * Hl - 1) c += 0.25f;
    return c;
}

template <int FMT, int K>
__global__ void __launch_bounds__(F_NT) wdg_upconv_fused_h16_kernel(const float* __restrict__ x, int ldx, long long isx,
                                                                   const wdg_h16<FMT>* __restrict__ w16, const float* __restrict__ bias,
                                                                   const float* __restrict__ affine, float* __restrict__ y, int ldy,
                                                                   long long isy, int Hl, int Wl, int act, float slope) {
    typedef wdg_h16x8<FMT> h16x8;
    static_assert(K % 32 == 0, "whole MFMA K-steps");
    constexpr int KO = K / 8, KS = K / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    h16x8* Xs = reinterpret_cast<h16x8*>(smem);                       // [KO][144]
    h16x8* Ws = Xs + KO * F_NPX;                                      // [KO][80]
    f32x4* Z = reinterpret_cast<f32x4*>(Ws + KO * F_NCOL);            // [144][20]
    f32x4* Hs = Z + F_NPX * F_PX;                                     // [12 * 16][5]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lq = lane >> 4;
    const int tiles_x = (Wl + F_TS - 1) / F_TS;
    const int i0 = (blockIdx.x / tiles_x) * F_TS, j0 = (blockIdx.x % tiles_x) * F_TS;
    const long long n = blockIdx.y;
    const float* ximg = x + n * isx;
    const wdg_srd srdW = wdg_make_srd(w16);

    // ---- weights of tap row 0 requested first, then the window of x
    constexpr int W_CH = KO * F_NCOL;                // 16-byte chunks of a tap row's weights
    constexpr int W_LD = (W_CH + F_NT - 1) / F_NT;
    u32x4 wr[W_LD];
    auto fetch_w = [&](int ty) {
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            const int q = t + F_NT * i;
            const int c = q / KO, oct = q - c * KO;
            wr[i] = __builtin_amdgcn_raw_buffer_load_b128(srdW, q < W_CH ? (int)((((unsigned)(ty * F_NCOL + c) * K) + oct * 8) << 1) : (int)WDG_SRD_OOB, 0, 0);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            const int q = t + F_NT * i;
            const int c = q / KO, oct = q - c * KO;
            if (q < W_CH) Ws[oct * F_NCOL + (c ^ (oct & 7))] = __builtin_bit_cast(h16x8, wr[i]);
        }
    };
    fetch_w(0);
    constexpr int X_SL = KO * F_NPX;                 // slots of the window
    constexpr int X_B = 12 * 256 / F_NT;                        // slots per thread and batch: the whole window in ONE round trip (133 KB of LDS
                                                     // leave one workgroup per CU = one wave per SIMD: registers are not the limit)
    for (int base = 0; base < X_SL; base += F_NT * X_B) {
        f32x4 v[X_B][2];
        int slot[X_B];
#pragma unroll
        for (int u = 0; u < X_B; ++u) {
            const int s = base + u * F_NT + t;
            const int pix = s / KO, oct = s - pix * KO;          // consecutive lanes: consecutive octets of one pixel (contiguous bytes)
            const int wy = pix / F_ZW, wx = pix - wy * F_ZW;
            const int gy = i0 - 2 + wy, gx = j0 - 2 + wx;
            const bool ok = s < X_SL && (unsigned)gy < (unsigned)Hl && (unsigned)gx < (unsigned)Wl;
            v[u][0] = v[u][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float* src = ximg + ((long long)gy * Wl + gx) * ldx + oct * 8;
                v[u][0] = *reinterpret_cast<const f32x4*>(src);
                v[u][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            slot[u] = s < X_SL ? oct * F_NPX + (pix ^ (oct & 7)) : -1;
        }
#pragma unroll
        for (int u = 0; u < X_B; ++u)
            if (slot[u] >= 0) Xs[slot[u]] = wdg_pack_h16<FMT>(v[u][0], v[u][1]);
    }

    // horizontal-pass operands of this thread: its outputs i = t + 256 k share channel group and output column, so the ten
    // (window column, coefficient) pairs do not depend on the tap row or the window row
    int hz[10];
    float hc[10];
    {
        const int o4 = t % F_CQ, hq = (t / F_CQ) % (2 * F_TS);
        const int gq = 2 * j0 + hq;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const int sx = gq + 3 - tx;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int b = (sx & 1) + 2 * e;
                const int rx = (sx - b) >> 1;
                const int rxl = rx - (j0 - 2);
                const bool on = (unsigned)rxl < (unsigned)F_ZW;
                hz[2 * tx + e] = on ? rxl * F_PX + tx * F_CQ + o4 : 0;
                hc[2 * tx + e] = on ? f_coef(rx, b, Wl) : 0.f;
            }
        }
    }
    // this thread's output pixel within the 16 x 16 tile
    constexpr int OQ = F_CQ * 256 / F_NT;            // channel groups per thread (the threads beyond 256 take the upper groups)
    const int tp = t & 255, og0 = (t >> 8) * OQ;
    const int qyl = tp >> 4, qxl = tp & 15;
    const int qy = 2 * i0 + qyl, qx = 2 * j0 + qxl;
    f32x4 acc[OQ];
#pragma unroll
    for (int o4 = 0; o4 < OQ; ++o4) acc[o4] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ty = 0; ty < 5; ++ty) {
        store_w();                                   // (Ws is free: the previous tap row's MFMA phase ended behind a barrier)
        __syncthreads();
        if (ty + 1 < 5) fetch_w(ty + 1);
        // ---- 1. z slice of this tap row: 9 pixel tiles x 5 column tiles (= tx), dealt round-robin to the waves
        // (two tiles at a time: their five-MFMA chains are independent, one hides the other's dependent-accumulator latency)
#pragma unroll
        for (int i = 0; i < (45 + F_NW - 1) / F_NW + 1; i += 2) {
            const int id0 = wave + F_NW * i, id1 = id0 + F_NW;
            if (id0 < 45) {
                const int pt0 = id0 / 5, ct0 = id0 - pt0 * 5;
                const bool two = id1 < 45;
                const int pt1 = two ? id1 / 5 : pt0, ct1 = two ? id1 - (id1 / 5) * 5 : ct0;
                f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
                h16x8 a0[KS], b0[KS], a1[KS], b1[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int oct = ks * 4 + lq;
                    a0[ks] = Ws[oct * F_NCOL + ((ct0 * 16 + li) ^ (oct & 7))];
                    b0[ks] = Xs[oct * F_NPX + ((pt0 * 16 + li) ^ (oct & 7))];
                    a1[ks] = Ws[oct * F_NCOL + ((ct1 * 16 + li) ^ (oct & 7))];
                    b1[ks] = Xs[oct * F_NPX + ((pt1 * 16 + li) ^ (oct & 7))];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    c0 = wdg_mfma16<FMT>(a0[ks], b0[ks], c0);
                    c1 = wdg_mfma16<FMT>(a1[ks], b1[ks], c1);
                }
                // register r of lane (li, lq): column ct * 16 + 4 lq + r (= output channel 4 lq + r of tap column tx = ct) of pixel li
                Z[(pt0 * 16 + li) * F_PX + ct0 * F_CQ + lq] = c0;
                if (two) Z[(pt1 * 16 + li) * F_PX + ct1 * F_CQ + lq] = c1;
            }
        }
        __syncthreads();
        // ---- 2. horizontal pass: H[ryl][qxl][o4] = sum_tx sum_{two (rx, b)} k_rx(b) z[ry, rx][(ty, tx), o4]
        for (int i = t; i < F_ZW * 2 * F_TS * F_CQ; i += F_NT) {
            const f32x4* zr = Z + (i / (F_CQ * 2 * F_TS)) * F_ZW * F_PX;
            f32x4 h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 10; ++j) h += hc[j] * zr[hz[j]];
            Hs[(i / F_CQ) * F_HP + (i % F_CQ)] = h;
        }
        __syncthreads();
        // ---- 3. vertical pass: the two (ry, a) pairs of this tap row
        {
            const int sy = qy + 3 - ty;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int a = (sy & 1) + 2 * e;
                const int ry = (sy - a) >> 1;
                const int ryl = ry - (i0 - 2);
                if ((unsigned)ryl < (unsigned)F_ZW) {
                    const float c = f_coef(ry, a, Hl);
#pragma unroll
                    for (int o4 = 0; o4 < OQ; ++o4) acc[o4] += c * Hs[(ryl * 2 * F_TS + qxl) * F_HP + og0 + o4];
                }
            }
        }
        __syncthreads();
    }
    if (qy < 2 * Hl && qx < 2 * Wl) {
        float* dst = y + n * isy + ((long long)qy * (2 * Wl) + qx) * ldy;
#pragma unroll
        for (int o = 0; o < OQ; ++o) {
            const int o4 = og0 + o;
            f32x4 v = acc[o];
            if (bias) v += *reinterpret_cast<const f32x4*>(bias + 4 * o4);
            if (act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], slope);
            }
            if (affine) v = v * *reinterpret_cast<const f32x4*>(affine + 4 * o4) + *reinterpret_cast<const f32x4*>(affine + 4 * F_CQ + 4 * o4);
            *reinterpret_cast<f32x4*>(dst + 4 * o4) = v;
        }
    }
}

template <int FMT, int K>
int fused_launch(dim3 grid, size_t lds, hipStream_t st, const float* x, int ldx, long long isx, const void* w16, const float* bias,
                 const float* affine, float* y, int ldy, long long isy, int Hl, int Wl, int act, float slope) {
    static bool attr = false;
    if (!attr) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_upconv_fused_h16_kernel<FMT, K>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((wdg_upconv_fused_h16_kernel<FMT, K>), grid, dim3(F_NT), lds, st, x, ldx, isx,
                       reinterpret_cast<const wdg_h16<FMT>*>(w16), bias, affine, y, ldy, isy, Hl, Wl, act, slope);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
}  // namespace

extern "C" int wdg_upconv_fused_h16_supported(int Cin, int C) { return C == 16 && Cin == 160; }

// y [n, 2 Hl, 2 Wl, >= C] = affine(act(bias + convT5x5(bilinear_x2(x_low)))), x_low [n, Hl, Wl, ldx >= Cin] fp32 (rounded to the
// operand format while staged), w16 = the layer's weight tensor [25 * C][Cin] in bf16 (fmt 0) / fp16 (fmt 1).
extern "C" int wdg_upconv_fused_h16(const float* x_low, int ldx, int64_t img_stride_x, const void* w16, int fmt, const float* bias,
                                    const float* affine, float* y, int ldy, int64_t img_stride_y, int n_img, int Hl, int Wl, int Cin,
                                    int C, int act, float slope, wdg_stream stream) {
    WDG_CHECK_ARG(x_low && w16 && y && (fmt == 0 || fmt == 1) && n_img > 0 && n_img < 65536 && Hl > 0 && Wl > 0, "bad argument");
    WDG_CHECK_ARG(wdg_upconv_fused_h16_supported(Cin, C), "unsupported channel counts (Cin 160, C 16)");
    WDG_CHECK_ARG(ldx % 4 == 0 && ldx >= Cin && ldy % 4 == 0 && ldy >= C, "pixel strides");
    WDG_CHECK_ARG(((uintptr_t)x_low & 15) == 0 && ((uintptr_t)w16 & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
                  ((uintptr_t)affine & 15) == 0, "x / w / y / bias / affine must be 16-byte aligned");
    constexpr int K = 160;
    const size_t lds = (size_t)(K / 8) * (F_NPX + F_NCOL) * 16 + (size_t)(F_NPX * F_PX + F_ZW * 2 * F_TS * F_HP) * 16;
    dim3 grid(((Hl + F_TS - 1) / F_TS) * ((Wl + F_TS - 1) / F_TS), n_img);
    hipStream_t st = (hipStream_t)stream;
    if (fmt == 0) return fused_launch<0, K>(grid, lds, st, x_low, ldx, img_stride_x, w16, bias, affine, y, ldy, img_stride_y, Hl, Wl, act, slope);
    return fused_launch<1, K>(grid, lds, st, x_low, ldx, img_stride_x, w16, bias, affine, y, ldy, img_stride_y, Hl, Wl, act, slope);
}
