// conv_igemm_bf16.hip — inference-precision variant of the implicit-GEMM convolution:
// bf16 operands on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate), fp32 accumulation, fp32 activations in
// HBM.  Used for the generator forward of the tiled-inference configuration (BASELINE.json configs[3]); the
// reference itself is fp32-only (gan/models.py), so the tolerance of this path is defined by the build
// (tests: <= 2e-2 relative to the fp32 path).
//
// Same GEMM view, same per-plan index tables and phases as conv_igemm.hip.  Differences:
//   * activations are read as fp32 (two float4 = 8 channels per slot) and rounded to bf16
//     (v_cvt_pk_bf16_f32, round-to-nearest-even) while being staged; weights are pre-converted bf16 copies of
//     the packed fp32 layouts (wdg_convert_bf16), so one 16-byte load is one LDS slot;
//   * a K-step is 64 (8 slots of 8 channels); a lane's ds_read_b128 is a whole MFMA operand (k = 8 per lane);
//   * the kernel is L2-bandwidth bound, not MFMA bound, so the epilogue can also apply the inference-mode
//     BatchNorm (per-channel scale/shift after the LeakyReLU) instead of a separate pass.
#include "conv_plan.h"
#include "h16.h"
#include <algorithm>

// Two 16-bit operand formats share every kernel of this file (template parameter FMT): 0 = bf16
// (v_mfma_f32_16x16x32_bf16), 1 = IEEE fp16 (v_mfma_f32_16x16x32_f16, BASELINE.json configs[4]).  Both round to
// nearest even from the fp32 activations while staging and accumulate in fp32.

struct WdgIgemmBf16 {
    const float* A;
    const void* B;         // 16-bit weights (format = the kernel's FMT)
    float* Out;
    const float* bias;
    const float* affine;   // optional [2*Ncols]: scale | shift applied after the activation
    const int4* ktab;
    long long imgStrideA, imgStrideO;
    int n_img, H, W, ldA;
    int Ho, Wo, ldO;
    int Ncols, ldB;
    int a_mul, o_mul;
    int act, accumulate;
    float slope;
    int Mmax, nphase;
    WdgPhase ph[9];
};

template <int BM, int BN, int WGM, int WGN, int FMT>
__global__ void __launch_bounds__(256) wdg_igemm_bf16_kernel(const WdgIgemmBf16 p) {
    typedef wdg_h16<FMT> h16;
    typedef wdg_h16x8<FMT> bf16x8;
    const h16* Bp = static_cast<const h16*>(p.B);
    constexpr int MT = BM / WGM / 16;
    constexpr int NT = BN / WGN / 16;
    constexpr int A_SLOTS = BM / 32;                // slots (8 channels of one row) per thread per K-step
    constexpr int B_SLOTS = (BN + 31) / 32;
    static_assert(WGM * WGN == 4, "4 waves");
    __shared__ bf16x8 ldsA[8 * BM];
    __shared__ bf16x8 ldsB[8 * BN];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int kg = t & 7, lrow = t >> 3;

    const WdgPhase ph = p.ph[blockIdx.z];
    const int tiles_m = (p.Mmax + BM - 1) / BM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int PaPb = ph.Pa * ph.Pb;
    const int Mph = p.n_img * PaPb;
    if (m0 >= Mph) return;
    const int nk = (ph.K4 + 15) >> 4;               // K-steps of 16 table entries (64 channels-taps)

    long long a_base[A_SLOTS];
    int a_ih0[A_SLOTS], a_iw0[A_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < Mph) {
            const int img = m / PaPb;
            const int rem = m - img * PaPb;
            const int pa = rem / ph.Pb, pb = rem - pa * ph.Pb;
            a_ih0[i] = pa * p.a_mul + ph.a_off_h;
            a_iw0[i] = pb * p.a_mul + ph.a_off_w;
            a_base[i] = (long long)img * p.imgStrideA + ((long long)a_ih0[i] * p.W + a_iw0[i]) * p.ldA;
        } else {
            a_ih0[i] = a_iw0[i] = -(1 << 28);
            a_base[i] = 0;
        }
    }
    long long b_base[B_SLOTS];
    bool b_ok[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
        const int nl = lrow + 32 * i;
        b_ok[i] = (nl < BN) && (n0 + nl < p.Ncols);
        b_base[i] = (long long)(n0 + nl) * p.ldB;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[A_SLOTS][2];
    bf16x8 rb[B_SLOTS];
    const int4* tab = p.ktab + ph.tab_off;
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kt) {
        const int k4 = kt * 16 + 2 * kg;            // this thread's pair of table entries
        int4 e0 = (int4){0, -(1 << 28), -(1 << 28), -1}, e1 = e0;
        if (k4 < ph.K4) e0 = tab[k4];
        if (k4 + 1 < ph.K4) e1 = tab[k4 + 1];
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int ih = a_ih0[i] + e0.y, iw = a_iw0[i] + e0.z;
            const bool ok0 = ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && (e0.w >= 0);
            const int ih1 = a_ih0[i] + e1.y, iw1 = a_iw0[i] + e1.z;
            const bool ok1 = ((unsigned)ih1 < (unsigned)p.H) && ((unsigned)iw1 < (unsigned)p.W) && (e1.w >= 0);
            ra[i][0] = ok0 ? *reinterpret_cast<const f32x4*>(p.A + a_base[i] + e0.x) : z4;
            ra[i][1] = ok1 ? *reinterpret_cast<const f32x4*>(p.A + a_base[i] + e1.x) : z4;
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (h16)0.f;
            // the weight rows are K-contiguous and zero-padded, and e1 directly follows e0 (Cin_p % 8 == 0)
            if (b_ok[i] && e0.w >= 0) v = *reinterpret_cast<const bf16x8*>(Bp + b_base[i] + e0.w);
            rb[i] = v;
        }
    };

    if (nk > 0) load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = (h16)ra[i][0][j];
                v[4 + j] = (h16)ra[i][1][j];
            }
            const int row = lrow + 32 * i;
            ldsA[kg * BM + (row ^ kg)] = v;
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int row = lrow + 32 * i;
            if (row < BN) ldsB[kg * BN + (row ^ kg)] = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kgr = 4 * h + (lane >> 4);
            bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = ldsA[kgr * BM + ((wm * (BM / WGM) + a * 16 + (lane & 15)) ^ kgr)];
#pragma unroll
            for (int b = 0; b < NT; ++b) bf[b] = ldsB[kgr * BN + ((wn * (BN / WGN) + b * 16 + (lane & 15)) ^ kgr)];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = wdg_mfma16<FMT>(af[a], bf[b], acc[a][b]);
        }
        __syncthreads();
    }

    const int col = lane & 15;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * (BM / WGM) + a * 16 + (lane >> 4) * 4 + r;
            if (m >= Mph) continue;
            const int img = m / PaPb;
            const int rem = m - img * PaPb;
            const int pa = rem / ph.Pb, pb = rem - pa * ph.Pb;
            const int oh = pa * p.o_mul + ph.o_off_h, ow = pb * p.o_mul + ph.o_off_w;
            float* dst = p.Out + (long long)img * p.imgStrideO + ((long long)oh * p.Wo + ow) * p.ldO;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + wn * (BN / WGN) + b * 16 + col;
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

template <int FMT>
__global__ void __launch_bounds__(256) wdg_convert_h16_kernel(const float* __restrict__ src, wdg_h16<FMT>* dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = (wdg_h16<FMT>)src[i];
}

static int convert_h16(const float* src, void* dst, int64_t n, int fmt, wdg_stream stream) {
    WDG_CHECK_ARG(src && dst && n >= 0, "bad argument");
    if (n == 0) return WDG_OK;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192));
    if (fmt == 0)
        hipLaunchKernelGGL(wdg_convert_h16_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n);
    else
        hipLaunchKernelGGL(wdg_convert_h16_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, n);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
extern "C" int wdg_convert_bf16(const float* src, void* dst_bf16, int64_t n, wdg_stream stream) {
    return convert_h16(src, dst_bf16, n, 0, stream);
}
extern "C" int wdg_convert_f16(const float* src, void* dst_f16, int64_t n, wdg_stream stream) {
    return convert_h16(src, dst_f16, n, 1, stream);
}

static int g_h16_small_tiles = 2;   // 0 off, 1 = 128x64, 2 = 64x64 tiles for launches with few 128x128 tiles (measured 6.45 / 6.25 / 6.03 ms per T=24 forward)
void wdg_h16_set_small_tiles(int v) { g_h16_small_tiles = v; }
static int launch_bf16(WdgIgemmBf16& p, int nphase, int fmt, hipStream_t st) {
    if (p.Mmax <= 0) return WDG_OK;
    p.nphase = nphase;
    bool wide = p.Ncols > 64;
    int BM = 128, BN = wide ? 128 : 64;
    // small maps (the per-timestep recurrent convolution at T > 1: 9,216 rows): the 16-bit MFMA finishes a 128x128 tile's
    // K-step in ~0.2 us, so a launch with about one workgroup per CU is bound by each workgroup's own load latency chain;
    // smaller tiles put several workgroups on every CU
    const long long t128 = (long long)((p.Mmax + 127) / 128) * ((p.Ncols + 127) / 128) * nphase;
    const int small = g_h16_small_tiles && wide && t128 <= 2 * 256 ? g_h16_small_tiles : 0;
    if (small == 1) { wide = false; BN = 64; }
    if (small == 2) { BM = 64; BN = 64; }
    const int tiles_m = (p.Mmax + BM - 1) / BM, tiles_n = (p.Ncols + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, nphase), block(256);
    if (small == 2) {
        if (fmt == 0)
            hipLaunchKernelGGL((wdg_igemm_bf16_kernel<64, 64, 2, 2, 0>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((wdg_igemm_bf16_kernel<64, 64, 2, 2, 1>), grid, block, 0, st, p);
    } else if (wide && fmt == 0)
        hipLaunchKernelGGL((wdg_igemm_bf16_kernel<128, 128, 2, 2, 0>), grid, block, 0, st, p);
    else if (wide)
        hipLaunchKernelGGL((wdg_igemm_bf16_kernel<128, 128, 2, 2, 1>), grid, block, 0, st, p);
    else if (fmt == 0)
        hipLaunchKernelGGL((wdg_igemm_bf16_kernel<128, 64, 2, 2, 0>), grid, block, 0, st, p);
    else
        hipLaunchKernelGGL((wdg_igemm_bf16_kernel<128, 64, 2, 2, 1>), grid, block, 0, st, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// y = affine(act(conv(x, wF16) + bias)); wF16 = bf16 copy of the forward-packed weights [Cout][taps][Cin_p]
static int conv_fwd_h16(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias,
                        const float* affine, float* y, int act, float slope, int accumulate, int fmt,
                        wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF16 && y, "null argument");
    WDG_CHECK_ARG(pl->Cin_p % 8 == 0, "bf16 path needs the padded input channel count to be a multiple of 8");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wF16 & 15) == 0, "alignment");
    const wdg_conv_geom& g = pl->g;
    // input patch resident in LDS (conv_patch_h16.hip) when the output map divides into its tiles
    const int prc = wdg_patch_h16_launch(pl, 0, x, wF16, bias, affine, y, act, slope, accumulate, fmt, (hipStream_t)stream);
    if (prc != 1) return prc;
    WdgIgemmBf16 p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.B = wF16; p.Out = y; p.bias = bias; p.affine = affine; p.ktab = pl->d_tab_fwd;
    p.imgStrideA = g.img_stride_x; p.imgStrideO = g.img_stride_y;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldA = g.ldx;
    p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy;
    p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p;
    p.a_mul = g.stride; p.o_mul = 1;
    p.act = act; p.slope = slope; p.accumulate = accumulate;
    p.Mmax = g.n_img * g.Ho * g.Wo;
    WdgPhase ph;
    ph.Pa = g.Ho; ph.Pb = g.Wo; ph.a_off_h = -g.pad_h; ph.a_off_w = -g.pad_w;
    ph.o_off_h = 0; ph.o_off_w = 0; ph.K4 = pl->K4_fwd; ph.tab_off = 0;
    p.ph[0] = ph;
    return launch_bf16(p, 1, fmt, (hipStream_t)stream);
}
extern "C" int wdg_conv_fwd_bf16(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias,
                                 const float* affine, float* y, int act, float slope, int accumulate,
                                 wdg_stream stream) {
    return conv_fwd_h16(pl, x, wF16, bias, affine, y, act, slope, accumulate, 0, stream);
}
extern "C" int wdg_conv_fwd_f16(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias,
                                const float* affine, float* y, int act, float slope, int accumulate,
                                wdg_stream stream) {
    return conv_fwd_h16(pl, x, wF16, bias, affine, y, act, slope, accumulate, 1, stream);
}

// dx = affine(act(conv_transpose(dy, wD16) + bias)) — the Conv2DTranspose forward in bf16.
static int conv_dgrad_h16(const wdg_conv_plan* pl, const float* dy, const void* wD16, const float* bias,
                          const float* affine, float* dx, int act, float slope, int accumulate, int fmt,
                          wdg_stream stream) {
    WDG_CHECK_ARG(pl && dy && wD16 && dx, "null argument");
    WDG_CHECK_ARG(pl->Cout_p % 8 == 0, "bf16 path needs the padded channel count of dy to be a multiple of 8");
    WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)wD16 & 15) == 0, "alignment");
    const wdg_conv_geom& g = pl->g;
    // 1 x 1 transposed conv (the column GEMM of the column-form upsample layer): a plain GEMM whose result is the larger
    // side -> the patch kernel's full-line stores (conv_patch_h16.hip)
    const int prc = wdg_patch_h16_launch(pl, 1, dy, wD16, bias, affine, dx, act, slope, accumulate, fmt, (hipStream_t)stream);
    if (prc != 1) return prc;
    // transposed k x k stride-k layer (the generator's 2 x 2 stride-2 Conv2DTranspose, models.py:58): taps do not overlap, so it is
    // the same GEMM with k * k * Cin columns and a scattering epilogue — 197 -> ~60 us on the shipped shape against the phase form below
    if (g.kh == g.kw && g.stride == g.kh && g.kh >= 2 && !g.pad_h && !g.pad_w) {
        const int src = wdg_patch_h16_launch(pl, 2, dy, wD16, bias, affine, dx, act, slope, accumulate, fmt, (hipStream_t)stream);
        if (src != 1) return src;
    }
    WdgIgemmBf16 p;
    memset(&p, 0, sizeof(p));
    p.A = dy; p.B = wD16; p.Out = dx; p.bias = bias; p.affine = affine; p.ktab = pl->d_tab_dgrad;
    p.imgStrideA = g.img_stride_y; p.imgStrideO = g.img_stride_x;
    p.n_img = g.n_img; p.H = g.Ho; p.W = g.Wo; p.ldA = g.ldy;
    p.Ho = g.H; p.Wo = g.W; p.ldO = g.ldx;
    p.Ncols = g.Cin; p.ldB = pl->Cout_p;
    p.a_mul = 1; p.o_mul = g.stride;
    p.act = act; p.slope = slope; p.accumulate = accumulate;
    int Mmax = 0;
    const int np = (int)pl->ph_dgrad.size();
    for (int i = 0; i < np; ++i) {
        p.ph[i] = pl->ph_dgrad[i];
        Mmax = std::max(Mmax, g.n_img * p.ph[i].Pa * p.ph[i].Pb);
    }
    p.Mmax = Mmax;
    return launch_bf16(p, np, fmt, (hipStream_t)stream);
}
extern "C" int wdg_conv_dgrad_bf16(const wdg_conv_plan* pl, const float* dy, const void* wD16, const float* bias,
                                   const float* affine, float* dx, int act, float slope, int accumulate,
                                   wdg_stream stream) {
    return conv_dgrad_h16(pl, dy, wD16, bias, affine, dx, act, slope, accumulate, 0, stream);
}
extern "C" int wdg_conv_dgrad_f16(const wdg_conv_plan* pl, const float* dy, const void* wD16, const float* bias,
                                  const float* affine, float* dx, int act, float slope, int accumulate,
                                  wdg_stream stream) {
    return conv_dgrad_h16(pl, dy, wD16, bias, affine, dx, act, slope, accumulate, 1, stream);
}

// Column GEMM of the column-form upsample + 5x5 transposed-conv layer (models.py:62-64, csrc/upconv_col.hip) with a 16-BIT
// result: z16[r, (t, o)] = sum_c x_low[r, c] * w[t][o][c] (16-bit operands, fp32 accumulation, rounded to the operand format
// on store).  z is the largest tensor of the inference forward — 400 columns per low-resolution pixel, 1.4 GB per 16-tile group
// in fp32 — and its only reader is the bilinear gather (wdg_upconv_gather_h16).  `plan`: the 1 x 1 plan whose x side is z
// ([n, H, W, 25 * C]) and whose y side is x_low.  fmt: 0 bf16, 1 fp16.
extern "C" int wdg_upconv_colgemm_h16_supported(const wdg_conv_plan* pl) {
    if (!pl || pl->g.kh != 1 || pl->g.kw != 1 || pl->g.stride != 1 || pl->g.pad_h || pl->g.pad_w || pl->Cout_p % 8) return 0;
    if (pl->g.Cin % 400 || pl->g.ldx != pl->g.Cin) return 0;          // z rows: 25 taps x a multiple of 16 channels (whole 16-column tiles in pairs of lanes), dense
    return wdg_patch_h16_eligible_t(pl);
}

extern "C" int wdg_upconv_colgemm_h16(const wdg_conv_plan* pl, const float* x_low, const void* wD16, void* z16, int fmt,
                                      wdg_stream stream) {
    WDG_CHECK_ARG(pl && x_low && wD16 && z16 && (fmt == 0 || fmt == 1), "bad argument");
    WDG_CHECK_ARG(((uintptr_t)x_low & 15) == 0 && ((uintptr_t)wD16 & 15) == 0 && ((uintptr_t)z16 & 15) == 0 && pl->Cout_p % 8 == 0, "alignment");
    const int rc = wdg_patch_h16_launch(pl, 1, x_low, wD16, nullptr, nullptr, (float*)z16, 0, 0.f, 0, fmt, (hipStream_t)stream, nullptr, 1);
    if (rc == 1) {
        wdg_set_error("wdg_upconv_colgemm_h16: geometry outside the patch kernel (use the fp32 z route)");
        return WDG_ERR_ARG;
    }
    return rc;
}

// ---- 16-bit ConvLSTM2D inference (gan/models.py:45): input part of the gates with interleaved columns, then one launch per
// timestep = recurrent convolution + cell update (conv_patch_h16.hip, LSTM epilogue)
static int g_lstm16_fused = 1;
void wdg_h16_set_lstm_fused(int v) { g_lstm16_fused = v; }
extern "C" int wdg_convlstm_h16_supported(const wdg_conv_plan* pl, int F) {
    if (!pl || !g_lstm16_fused || F <= 0 || F % 4) return 0;
    const wdg_conv_geom& g = pl->g;
    if (g.Cout != 4 * F || g.ldy != 4 * F || g.img_stride_y != (int64_t)g.Ho * g.Wo * 4 * F || g.stride != 1) return 0;
    return wdg_patch_h16_eligible(pl);
}
extern "C" int wdg_conv_fwd_h16_gates(const wdg_conv_plan* pl, const float* x, const void* wF16, const float* bias, float* y,
                                      int F, int fmt, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF16 && y && wdg_convlstm_h16_supported(pl, F), "not supported for this geometry");
    WdgPatchGates gx;
    memset(&gx, 0, sizeof(gx));
    gx.F = F;
    const int rc = wdg_patch_h16_launch(pl, 0, x, wF16, bias, nullptr, y, 0, 0.f, 0, fmt, (hipStream_t)stream, &gx);
    WDG_CHECK_ARG(rc != 1, "patch kernel refused the geometry");
    return rc;
}
extern "C" int wdg_convlstm_step_h16(const wdg_conv_plan* pl, const float* h_prev, const void* wF16, const float* gates_x,
                                     const float* c_prev, float* c_out, int ldc, float* h_out, int ldh, int F, int fmt,
                                     wdg_stream stream) {
    WDG_CHECK_ARG(pl && wF16 && gates_x && c_out && h_out && wdg_convlstm_h16_supported(pl, F), "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= F && ldh >= F && (h_prev != nullptr) == (c_prev != nullptr), "bad argument");
    WdgPatchGates gx;
    memset(&gx, 0, sizeof(gx));
    gx.F = F; gx.gates_x = gates_x; gx.c_prev = c_prev; gx.c_out = c_out; gx.ldc = ldc; gx.h_out = h_out; gx.ldh = ldh;
    gx.skip_k = h_prev == nullptr;
    const int rc = wdg_patch_h16_launch(pl, 0, h_prev ? h_prev : gates_x, wF16, nullptr, nullptr, h_out, 0, 0.f, 0, fmt,
                                        (hipStream_t)stream, &gx);
    WDG_CHECK_ARG(rc != 1, "patch kernel refused the geometry");
    return rc;
}

// The same two with the hidden state / the layer input in the 16-bit operand format (see "activations in the 16-bit operand format"
// below): x16 != 0: x holds 16-bit elements; h_prev16 / h_out16: the state the step reads / an additional (h_out != NULL) or the only
// (h_out == NULL) copy of the state it writes — what the next step and the next layer round h to while staging it.
extern "C" int wdg_conv_fwd_h16_gates_x16(const wdg_conv_plan* pl, const void* x, int x16, const void* wF16, const float* bias, float* y,
                                          int F, int fmt, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF16 && y && wdg_convlstm_h16_supported(pl, F), "not supported for this geometry");
    WDG_CHECK_ARG(!x16 || pl->g.ldx % 8 == 0, "16-bit input: pixel stride a multiple of 8 elements");
    WdgPatchGates gx;
    memset(&gx, 0, sizeof(gx));
    gx.F = F;
    const int rc = wdg_patch_h16_launch(pl, 0, reinterpret_cast<const float*>(x), wF16, bias, nullptr, y, 0, 0.f, 0, fmt, (hipStream_t)stream, &gx, 0, x16);
    WDG_CHECK_ARG(rc != 1, "patch kernel refused the geometry");
    return rc;
}
extern "C" int wdg_convlstm_step_h16x(const wdg_conv_plan* pl, const void* h_prev, int h_prev16, const void* wF16, const float* gates_x,
                                      const float* c_prev, float* c_out, int ldc, float* h_out, int ldh, void* h_out16, int ldh16, int F, int fmt,
                                      wdg_stream stream) {
    WDG_CHECK_ARG(pl && wF16 && gates_x && c_out && (h_out || h_out16) && wdg_convlstm_h16_supported(pl, F), "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= F && (!h_out || ldh >= F) && (!h_out16 || ldh16 >= F) && (h_prev != nullptr) == (c_prev != nullptr), "bad argument");
    WDG_CHECK_ARG(!h_prev16 || pl->g.ldx % 8 == 0, "16-bit state: pixel stride a multiple of 8 elements");
    WdgPatchGates gx;
    memset(&gx, 0, sizeof(gx));
    gx.F = F; gx.gates_x = gates_x; gx.c_prev = c_prev; gx.c_out = c_out; gx.ldc = ldc; gx.h_out = h_out; gx.ldh = ldh;
    gx.h16_out = h_out16; gx.ldh16 = ldh16;
    gx.skip_k = h_prev == nullptr;
    const int rc = wdg_patch_h16_launch(pl, 0, h_prev ? reinterpret_cast<const float*>(h_prev) : gates_x, wF16, nullptr, nullptr, h_out, 0, 0.f, 0, fmt,
                                        (hipStream_t)stream, &gx, 0, h_prev ? h_prev16 : 0);
    WDG_CHECK_ARG(rc != 1, "patch kernel refused the geometry (16-bit state: whole 128-column tiles, aligned rows)");
    return rc;
}

// ---- activations in the 16-bit operand format between two 16-bit layers --------------------------------------------------
// A layer of the inference-precision forward rounds its input to the operand format while staging it.  When every reader of a
// tensor is such a layer, the PRODUCER can store it rounded: the values multiplied are the same bits, the tensor has half the
// bytes.  in16: x holds 16-bit elements (the plan's ldx / image stride are in elements of x as it is stored), out16: y likewise.
// Only the input-patch kernel (conv_patch_h16.hip) takes this route: `transposed` 0 = the forward conv of the plan, 1 = its
// transposed direction (1 x 1, or k x k stride-k as one GEMM with a scattering epilogue).  The query says whether it would.
extern "C" int wdg_conv_h16_act16_supported(const wdg_conv_plan* pl, int transposed, int in16, int out16) {
    if (!pl) return 0;
    const wdg_conv_geom& g = pl->g;
    if (!transposed) {
        if (pl->Cin_p % 8 || (in16 && g.ldx % 8) || (out16 && (g.Cout % 16 || g.ldy % 8))) return 0;
        return wdg_patch_h16_eligible(pl);
    }
    if (pl->Cout_p % 8 || (in16 && g.ldy % 8) || (out16 && g.ldx % 8)) return 0;
    if (g.kh == 1 && g.kw == 1) return (!out16 || g.Cin % 16 == 0) && wdg_patch_h16_eligible_t(pl);
    return (!out16 || g.Cin % 32 == 0) && wdg_patch_h16_eligible_s(pl);
}
extern "C" int wdg_conv_fwd_h16_act16(const wdg_conv_plan* pl, const void* x, int in16, const void* wF16, int fmt, const float* bias,
                                      const float* affine, void* y, int out16, int act, float slope, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF16 && y && (fmt == 0 || fmt == 1), "bad argument");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wF16 & 15) == 0 && ((uintptr_t)y & 15) == 0, "alignment");
    const int rc = wdg_patch_h16_launch(pl, 0, reinterpret_cast<const float*>(x), wF16, bias, affine, reinterpret_cast<float*>(y), act, slope,
                                        0, fmt, (hipStream_t)stream, nullptr, out16, in16);
    if (rc == 1) {
        wdg_set_error("wdg_conv_fwd_h16_act16: geometry outside the patch kernel (wdg_conv_h16_act16_supported)");
        return WDG_ERR_ARG;
    }
    return rc;
}
extern "C" int wdg_conv_dgrad_h16_act16(const wdg_conv_plan* pl, const void* dy, int in16, const void* wD16, int fmt, const float* bias,
                                        const float* affine, void* dx, int out16, int act, float slope, wdg_stream stream) {
    WDG_CHECK_ARG(pl && dy && wD16 && dx && (fmt == 0 || fmt == 1), "bad argument");
    WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)wD16 & 15) == 0 && ((uintptr_t)dx & 15) == 0, "alignment");
    const wdg_conv_geom& g = pl->g;
    const int mode = (g.kh == 1 && g.kw == 1) ? 1 : 2;
    const int rc = wdg_patch_h16_launch(pl, mode, reinterpret_cast<const float*>(dy), wD16, bias, affine, reinterpret_cast<float*>(dx), act,
                                        slope, 0, fmt, (hipStream_t)stream, nullptr, out16, in16);
    if (rc == 1) {
        wdg_set_error("wdg_conv_dgrad_h16_act16: geometry outside the patch kernel (wdg_conv_h16_act16_supported)");
        return WDG_ERR_ARG;
    }
    return rc;
}
