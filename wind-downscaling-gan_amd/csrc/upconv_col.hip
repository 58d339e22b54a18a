// upconv_col.hip — backward of the generator's "bilinear x2 upsample -> 5x5 transposed conv" block
// (/root/reference/src/downscaling/gan/models.py:60-64) in COLUMN form, on the low-resolution grid.
//
// With X = U(x) the upsampled tensor (C_in = 160 channels at 256x256) the layer is
//     y[p, o] = sum_t sum_c X[p - t + 2, c] * w[t][o][c]            (t = 5x5 taps, o = 16 output channels)
// and the textbook backward runs two 268-GFLOP convolutions at full resolution (dX = conv(dy, w), dW = X (*) dy)
// plus the upsample adjoint.  Bilinear interpolation acts per channel, so it commutes with the channel mixing:
// pulling it through the weights leaves 1x1 GEMMs on the LOW-resolution grid — a quarter of the multiply-adds:
//
//     col[r, (t, o)] = sum_{a,b in 0..3} k_r(a) k_r(b) dy'[q + t - 2, o],   q = 2r - 1 + (a, b)
//     dx[r, c]       = sum_{(t,o)} col[r, (t,o)] * w[t][o][c]              (GEMM  M = pixels, K = 25*16, N = 160)
//     dw[t][o][c]   += sum_r col[r, (t,o)] * x[r, c]                       (GEMM  M = 25*16,  K = pixels, N = 160)
//
// k_r(a) is the adjoint of the clamped bilinear stencil: (1/4, 3/4, 3/4, 1/4) on hi-res rows q = 2r-1 .. 2r+2, zero
// where q falls outside the image, and the edge rows q = 0 / q = H-1 collect the 1/4 that the clamp redirects to them
// (dy' = dy zero-extended).  The image border enters only through these coefficients, which the column kernel
// applies exactly; both GEMMs then run in the implicit-GEMM kernels of conv_igemm.hip as 1x1 convolutions on the
// low-res grid.  w viewed as [400][160] IS the layer's weight tensor, so no extra packing exists, dx lands directly
// in the caller's buffer and x is read in place.
#include "common.h"

namespace {
constexpr int TS = 8;             // tile edge on the low-res grid
constexpr int WIN = 2 * TS + 6;   // edge of the hi-res window that feeds one tile

__device__ __forceinline__ float up_coef(int a) { return (a == 0 || a == 3) ? 0.25f : 0.75f; }
}  // namespace

// ---- column tensor of the output gradient: col[n, ry, rx, t*C + o] on the low-res grid
// One workgroup = an 8x8 tile of low-res pixels.  The 22x22 hi-res window of dy it needs is staged in LDS once; per
// tap row ty the vertical 4-tap sums go to a second LDS array, the horizontal 4-tap sums are formed on the way out.
// Writes are 16 bytes per lane, 5*C consecutive floats per (pixel, ty).
__device__ __forceinline__ float up_adj_coef(int r, int a, int Hl) {
    const int q = 2 * r - 1 + a;                 // hi-res row (column) this tap reads
    if ((unsigned)q >= (unsigned)(2 * Hl)) return 0.f;
    float c = up_coef(a);
    if (q == 0 || q == 2 * Hl - 1) c += 0.25f;   // the clamp sends the out-of-range neighbour's share here
    return c;
}

template <int CQ>
__global__ void __launch_bounds__(256) wdg_upconv_col_kernel(const float* __restrict__ dy, int ldy, long long isy,
                                                             float* __restrict__ col, int Hl, int Wl) {
    __shared__ f32x4 T[WIN * WIN * CQ];
    __shared__ f32x4 G[2][TS * WIN * CQ];
    const int H = 2 * Hl, W = 2 * Wl;
    const int tiles_x = (Wl + TS - 1) / TS;
    const int ry0 = (blockIdx.x / tiles_x) * TS, rx0 = (blockIdx.x % tiles_x) * TS;
    const long long n = blockIdx.y;
    const int y0 = 2 * ry0 - 3, x0 = 2 * rx0 - 3;   // hi-res origin of the window: row of (r = ry0, a = 0, ty = 0)
    const float* src = dy + n * isy;
    for (int i = threadIdx.x; i < WIN * WIN * CQ; i += 256) {
        const int o4 = i % CQ, c = (i / CQ) % WIN, r = i / (CQ * WIN);
        const int gy = y0 + r, gx = x0 + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = *reinterpret_cast<const f32x4*>(src + ((long long)gy * W + gx) * ldy + 4 * o4);
        T[i] = v;
    }
    __syncthreads();
    float* dst = col + n * Hl * Wl * (100LL * CQ);
    for (int ty = 0; ty < 5; ++ty) {
        f32x4* Gt = G[ty & 1];
        for (int i = threadIdx.x; i < TS * WIN * CQ; i += 256) {
            const int o4 = i % CQ, hx = (i / CQ) % WIN, ryl = i / (CQ * WIN);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a)
                acc += up_adj_coef(ry0 + ryl, a, Hl) * T[((2 * ryl + a + ty) * WIN + hx) * CQ + o4];
            Gt[i] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TS * TS * 5 * CQ; i += 256) {
            const int j = i % (5 * CQ), pix = i / (5 * CQ);
            const int tx = j / CQ, o4 = j % CQ;
            const int rxl = pix % TS, ryl = pix / TS;
            const int ry = ry0 + ryl, rx = rx0 + rxl;
            if (ry >= Hl || rx >= Wl) continue;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc += up_adj_coef(rx, b, Wl) * Gt[(ryl * WIN + 2 * rxl + b + tx) * CQ + o4];
            *reinterpret_cast<f32x4*>(dst + ((long long)ry * Wl + rx) * (100LL * CQ) + (ty * 5 + tx) * (4 * CQ) + 4 * o4) = acc;
        }
        // the next iteration writes the other G buffer; its barrier orders this iteration's reads before the
        // writes of iteration ty + 2
    }
}

extern "C" int wdg_upconv_col_supported(int C) { return C == 4 || C == 8 || C == 16; }

extern "C" int wdg_upconv_col(const float* dy, int ldy, int64_t img_stride_dy, float* col, int n_img, int Hl, int Wl,
                              int C, wdg_stream stream) {
    WDG_CHECK_ARG(dy && col && n_img > 0 && n_img < 65536 && Hl > 0 && Wl > 0 && ldy % 4 == 0, "bad argument");
    WDG_CHECK_ARG(wdg_upconv_col_supported(C), "channel count must be 4, 8 or 16");
    WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)col & 15) == 0, "dy / col must be 16-byte aligned");
    const int tiles = ((Hl + TS - 1) / TS) * ((Wl + TS - 1) / TS);
    dim3 grid(tiles, n_img), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C == 16)
        hipLaunchKernelGGL(wdg_upconv_col_kernel<4>, grid, block, 0, st, dy, ldy, (long long)img_stride_dy, col, Hl, Wl);
    else if (C == 8)
        hipLaunchKernelGGL(wdg_upconv_col_kernel<2>, grid, block, 0, st, dy, ldy, (long long)img_stride_dy, col, Hl, Wl);
    else
        hipLaunchKernelGGL(wdg_upconv_col_kernel<1>, grid, block, 0, st, dy, ldy, (long long)img_stride_dy, col, Hl, Wl);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- forward in column form: y = act(bias + gather(z)),  z[r, (t, o)] = sum_c x[r, c] * w[t][o][c]  (a 1x1 GEMM) ----
// The adjoint of the column kernel above: y[p, o] = sum over (r, t, a, b) with 2r - 1 + (a, b) + t - 2 = p of
// k_r(a) k_r(b) z[r, (t, o)].  Per dimension and tap there are exactly two (r, a) pairs (a = parity of p + 3 - t, or
// that + 2), so the sum separates into a horizontal pass (10 terms) and a vertical pass (10 terms) instead of 100.
// One workgroup = an 8x8 low-res tile = 16x16 output pixels; per tap row ty the 12x12 low-res window of z (80 floats per
// pixel) is staged in LDS, reduced horizontally into H[12][16][C], and each thread adds its two vertical terms.
// ZF: element format of z — 0 fp32, 1 bf16, 2 fp16 (16-bit: wdg_upconv_colgemm_h16 wrote it; a thread's slot is then 8 values =
// 16 bytes, widened to fp32 on the way into LDS; everything behind the load is the fp32 kernel)
static int g_gather_xcd = 1;      // wdg_upconv_set_gather_xcd (wdg_set_tuning("gather_xcd", 0/1))
void wdg_upconv_set_gather_xcd(int v) { g_gather_xcd = v != 0; }
template <int ZF> struct WdgZT { typedef float T; };
template <> struct WdgZT<1> { typedef __bf16 T; };
template <> struct WdgZT<2> { typedef _Float16 T; };
template <int CQ, int ZF = 0>
__global__ void __launch_bounds__(256) wdg_upconv_gather_kernel(const typename WdgZT<ZF>::T* __restrict__ z, const float* __restrict__ bias,
                                                                const float* __restrict__ affine, float* __restrict__ y,
                                                                int ldy, long long isy, int Hl, int Wl, int act, float slope,
                                                                double* stats, int stats_rep, int xcd) {
    constexpr int ZW = TS + 4;                       // low-res window edge (12)
    constexpr int PX = 5 * CQ;                       // float4 per window pixel and tap row
    __shared__ f32x4 Z[ZW * ZW * PX];
    // pixel pitch CQ + 1 slots: in the vertical pass a lane reads the CQ slots of ITS output pixel, consecutive lanes consecutive
    // pixels — at a pitch of CQ = 4 slots every lane of a ds_read_b128 group fell on the same four bank columns (4-way conflict
    // on all eight reads per tap row: SQ_LDS_BANK_CONFLICT 0.26 of the kernel's LDS cycles); 5 is coprime to the 16 columns
    constexpr int HP = CQ + (CQ % 2 == 0 ? 1 : 0);
    __shared__ f32x4 Hs[ZW * 2 * TS * HP];
    const int tiles_x = (Wl + TS - 1) / TS;
    // XCD-contiguous order (xcd != 0): hardware deals consecutive workgroups round-robin over the 8 XCDs, so the horizontally
    // adjacent tiles of an image — which share a 2-pixel ring of z, 2.25 x the tile in window reads — landed on eight different L2s;
    // each XCD now walks a contiguous range of (image, tile) pairs
    int tile_id = blockIdx.x;
    long long n = blockIdx.y;
    if (xcd) {
        const int w = wdg_xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
        tile_id = w % (int)gridDim.x;
        n = w / (int)gridDim.x;
    }
    const int i0 = (tile_id / tiles_x) * TS, j0 = (tile_id % tiles_x) * TS;
    const int t = threadIdx.x;
    const int qyl = t >> 4, qxl = t & 15;            // this thread's output pixel within the 16x16 tile
    const int qy = 2 * i0 + qyl, qx = 2 * j0 + qxl;
    const typename WdgZT<ZF>::T* zimg = z + n * Hl * Wl * (100LL * CQ);
    f32x4 acc[CQ];
#pragma unroll
    for (int o4 = 0; o4 < CQ; ++o4) acc[o4] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // window slots of this thread (tile-invariant): element offset within a tap-row slice, or -1 outside the image.  A slot is one
    // 16-byte load: four fp32 values, or EIGHT 16-bit values that become two consecutive float4 of the LDS window (8-byte loads of
    // four values ran at 0.6x the rate and made the 16-bit z slower than the fp32 one)
    constexpr int VPS = ZF ? 2 : 1;                  // float4 per slot
    static_assert(ZF == 0 || PX % 2 == 0, "16-bit z: channel count a multiple of 8");
    constexpr int NSL = ZW * ZW * PX / VPS;          // slots of the window
    constexpr int NSLOT = (NSL + 255) / 256;
    long long zoff[NSLOT];
    f32x4 zr[NSLOT][VPS];
#pragma unroll
    for (int s_ = 0; s_ < NSLOT; ++s_) {
        const int i = t + 256 * s_;
        const int k = i % (PX / VPS), px = i / (PX / VPS);
        const int ry = i0 - 2 + px / ZW, rx = j0 - 2 + px % ZW;
        zoff[s_] = (i < NSL && (unsigned)ry < (unsigned)Hl && (unsigned)rx < (unsigned)Wl)
                       ? ((long long)ry * Wl + rx) * (100LL * CQ) + 4 * VPS * k : -1;
    }
    auto load_slice = [&](int ty_) {
#pragma unroll
        for (int s_ = 0; s_ < NSLOT; ++s_)
        {
            if constexpr (ZF == 0) {
                zr[s_][0] = zoff[s_] >= 0 ? *reinterpret_cast<const f32x4*>(zimg + zoff[s_] + ty_ * (20 * CQ)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            } else {
                typedef typename WdgZT<ZF>::T zt8 __attribute__((ext_vector_type(8)));
                zr[s_][0] = zr[s_][VPS - 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (zoff[s_] >= 0) {
                    const zt8 q = *reinterpret_cast<const zt8*>(zimg + zoff[s_] + ty_ * (20 * CQ));
                    zr[s_][0] = (f32x4){(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
                    zr[s_][VPS - 1] = (f32x4){(float)q[4], (float)q[5], (float)q[6], (float)q[7]};
                }
            }
        }
    };
    // horizontal-pass operands of this thread (see the pass): slot offset inside a window row and coefficient, 0 * slot 0 outside
    int hz[10];
    float hc[10];
    {
        static_assert(256 % (2 * TS * CQ) == 0 || (2 * TS * CQ) % 256 == 0, "a thread's horizontal-pass outputs share (column, group)");
        const int o4 = t % CQ, hq = (t / CQ) % (2 * TS);
        const int gq = 2 * j0 + hq;
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const int sx = gq + 3 - tx;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int b = (sx & 1) + 2 * e;
                const int rx = (sx - b) >> 1;
                const int rxl = rx - (j0 - 2);
                const bool on = (unsigned)rxl < (unsigned)ZW;
                hz[2 * tx + e] = on ? rxl * PX + tx * CQ + o4 : 0;
                hc[2 * tx + e] = on ? up_adj_coef(rx, b, Wl) : 0.f;
            }
        }
    }
    load_slice(0);
    for (int ty = 0; ty < 5; ++ty) {
        // 1. window of z for this tap row (zeros outside the low-res image): staged from the registers that were
        //    loaded under the previous tap row's passes
#pragma unroll
        for (int s_ = 0; s_ < NSLOT; ++s_)
            if (t + 256 * s_ < NSL) {
#pragma unroll
                for (int v = 0; v < VPS; ++v) Z[(t + 256 * s_) * VPS + v] = zr[s_][v];
            }
        __syncthreads();
        if (ty + 1 < 5) load_slice(ty + 1);
        // 2. horizontal pass: H[ryl][qxl][o4] = sum_tx sum_{two (rx, b)} k_rx(b) z[ry, rx][(ty, tx), o4].  A thread's outputs
        // i = t + 256 k share the channel group and the output column (256 = 16 window rows' worth of (column, group) pairs), so its
        // ten (window column, coefficient) pairs are tap-row- and row-independent: formed once per workgroup (hz / hc above)
        if constexpr ((2 * TS * CQ) % 256 == 0 || 256 % (2 * TS * CQ) == 0) {
            for (int i = t; i < ZW * 2 * TS * CQ; i += 256) {
                const int ryl = i / (CQ * 2 * TS);
                const f32x4* zr = Z + ryl * ZW * PX;
                f32x4 h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 10; ++j) h += hc[j] * zr[hz[j]];
                Hs[(i / CQ) * HP + (i % CQ)] = h;
            }
        }
        __syncthreads();
        // 3. vertical pass: the two (ry, a) pairs of this tap row
        {
            const int sy = qy + 3 - ty;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int a = (sy & 1) + 2 * e;
                const int ry = (sy - a) >> 1;
                const int ryl = ry - (i0 - 2);
                if ((unsigned)ryl < (unsigned)ZW) {
                    const float c = up_adj_coef(ry, a, Hl);
#pragma unroll
                    for (int o4 = 0; o4 < CQ; ++o4) acc[o4] += c * Hs[(ryl * 2 * TS + qxl) * HP + o4];
                }
            }
        }
        __syncthreads();
    }
    const bool inside = qy < 2 * Hl && qx < 2 * Wl;
    float* dst = y + n * isy + ((long long)qy * (2 * Wl) + qx) * ldy;
#pragma unroll
    for (int o4 = 0; o4 < CQ; ++o4) {
        f32x4 v = acc[o4];
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + 4 * o4);
        if (act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], slope);
        }
        if (affine)     // fused inference BatchNorm: scale | shift per channel after the activation
            v = v * *reinterpret_cast<const f32x4*>(affine + 4 * o4) + *reinterpret_cast<const f32x4*>(affine + 4 * CQ + 4 * o4);
        if (inside) *reinterpret_cast<f32x4*>(dst + 4 * o4) = v;
        acc[o4] = inside ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (stats) {
        // training-mode BatchNorm producer: per-channel sum / sum of squares of this block's 256 pixels -> one replica
        // slab [2][4*CQ] (wave butterflies, the four waves meet in LDS, one fp64 atomic pair per channel and block)
        float* red = reinterpret_cast<float*>(Hs);       // the tap loop has ended behind a barrier
#pragma unroll
        for (int o4 = 0; o4 < CQ; ++o4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = wdg_wave_sum_fast(acc[o4][r]), s2 = wdg_wave_sum_fast(acc[o4][r] * acc[o4][r]);
                if ((t & 63) == 0) {
                    red[((t >> 6) * 4 * CQ + 4 * o4 + r) * 2 + 0] = s1;
                    red[((t >> 6) * 4 * CQ + 4 * o4 + r) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        if (t < 4 * CQ) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s1 += red[(w * 4 * CQ + t) * 2 + 0];
                s2 += red[(w * 4 * CQ + t) * 2 + 1];
            }
            double* slab = stats + (size_t)((blockIdx.x + blockIdx.y) % stats_rep) * 2 * (4 * CQ);
            atomicAdd(slab + t, (double)s1);
            atomicAdd(slab + 4 * CQ + t, (double)s2);
        }
    }
}

extern "C" int wdg_upconv_gather_h16(const void* z16, int fmt, const float* bias, const float* affine, float* y, int ldy,
                                     int64_t img_stride_y, int n_img, int Hl, int Wl, int C, int act, float slope, wdg_stream stream) {
    WDG_CHECK_ARG(z16 && y && (fmt == 0 || fmt == 1) && n_img > 0 && n_img < 65536 && Hl > 0 && Wl > 0 && ldy % 4 == 0, "bad argument");
    WDG_CHECK_ARG(wdg_upconv_col_supported(C), "channel count must be 4, 8 or 16");
    WDG_CHECK_ARG(((uintptr_t)z16 & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)affine & 15) == 0, "z / y / bias / affine must be 16-byte aligned");
    const int tiles = ((Hl + TS - 1) / TS) * ((Wl + TS - 1) / TS);
    dim3 grid(tiles, n_img), block(256);
    hipStream_t st = (hipStream_t)stream;
#define WDG_GATHER16(CQ_)                                                                                                          \
    if (fmt == 0)                                                                                                                  \
        hipLaunchKernelGGL((wdg_upconv_gather_kernel<CQ_, 1>), grid, block, 0, st, (const __bf16*)z16, bias, affine, y, ldy,      \
                           (long long)img_stride_y, Hl, Wl, act, slope, (double*)nullptr, 0, g_gather_xcd);                                     \
    else                                                                                                                           \
        hipLaunchKernelGGL((wdg_upconv_gather_kernel<CQ_, 2>), grid, block, 0, st, (const _Float16*)z16, bias, affine, y, ldy,    \
                           (long long)img_stride_y, Hl, Wl, act, slope, (double*)nullptr, 0, g_gather_xcd);
    if (C == 16) { WDG_GATHER16(4) }
    else if (C == 8) { WDG_GATHER16(2) }
    else {
        wdg_set_error("wdg_upconv_gather_h16: channel count must be 8 or 16 (16-byte slots of eight 16-bit values)");
        return WDG_ERR_ARG;
    }
#undef WDG_GATHER16
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_upconv_gather(const float* z, const float* bias, const float* affine, float* y, int ldy,
                                 int64_t img_stride_y, int n_img, int Hl, int Wl, int C, int act, float slope,
                                 double* stats, int stats_rep, wdg_stream stream) {
    WDG_CHECK_ARG(!stats || (stats_rep >= 1 && !affine), "stats: replicas >= 1, not together with affine");
    WDG_CHECK_ARG(z && y && n_img > 0 && n_img < 65536 && Hl > 0 && Wl > 0 && ldy % 4 == 0, "bad argument");
    WDG_CHECK_ARG(wdg_upconv_col_supported(C), "channel count must be 4, 8 or 16");
    WDG_CHECK_ARG(((uintptr_t)z & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)affine & 15) == 0, "z / y / bias / affine must be 16-byte aligned");
    const int tiles = ((Hl + TS - 1) / TS) * ((Wl + TS - 1) / TS);
    dim3 grid(tiles, n_img), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C == 16)
        hipLaunchKernelGGL(wdg_upconv_gather_kernel<4>, grid, block, 0, st, z, bias, affine, y, ldy, (long long)img_stride_y, Hl, Wl, act, slope, stats, stats_rep, g_gather_xcd);
    else if (C == 8)
        hipLaunchKernelGGL(wdg_upconv_gather_kernel<2>, grid, block, 0, st, z, bias, affine, y, ldy, (long long)img_stride_y, Hl, Wl, act, slope, stats, stats_rep, g_gather_xcd);
    else
        hipLaunchKernelGGL(wdg_upconv_gather_kernel<1>, grid, block, 0, st, z, bias, affine, y, ldy, (long long)img_stride_y, Hl, Wl, act, slope, stats, stats_rep, g_gather_xcd);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
