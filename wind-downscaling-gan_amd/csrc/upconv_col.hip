// upconv_col.hip — backward of the generator's "bilinear x2 upsample -> 5x5 transposed conv" block
// (/root/reference/src/downscaling/gan/models.py:60-64) in COLUMN form, on the low-resolution grid.
//
// With X = U(x) the upsampled tensor (C_in = 160 channels at 256x256) the layer is
//     y[p, o] = sum_t sum_c X[p - t + 2, c] * w[t][o][c]            (t = 5x5 taps, o = 16 output channels)
// and the textbook backward runs two 268-GFLOP convolutions at full resolution (dX = conv(dy, w), dW = X (*) dy)
// plus the upsample adjoint.  Bilinear interpolation acts per channel, so it commutes with the channel mixing:
// pulling it through the weights leaves 1x1 GEMMs on the LOW-resolution grid — a quarter of the multiply-adds:
//
//     col[r, (t, o)] = sum_{a,b in 0..3} c_a c_b m[q] dy'[q + t - 2, o],   q = 2r - 1 + (a, b),  c = (1/4, 3/4, 3/4, 1/4)
//     dx_ext[r, c]   = sum_{(t,o)} col[r, (t,o)] * w[t][o][c]              (GEMM  M = pixels, K = 25*16, N = 160)
//     dw[t][o][c]   += sum_r col[r, (t,o)] * x_ext[r, c]                   (GEMM  M = 25*16,  K = pixels, N = 160)
//
// r runs over the low-res grid EXTENDED by one replicated pixel on every side (x_ext = replicate_pad(x, 1)): clamped
// bilinear upsampling of x is the unclamped 4-tap pattern on x_ext, so every r has the same coefficients and the
// image border only enters through the masks (m[q] = q inside the hi-res image, dy' = dy zero-extended), which this
// file's column kernel applies exactly.  dx = fold(dx_ext) adds the pad ring back onto the edge pixels.
// Both GEMMs run in the implicit-GEMM kernels of conv_igemm.hip as 1x1 convolutions; w viewed as [400][160] IS the
// layer's weight tensor, so no extra packing exists.  This file holds the three memory-bound helpers.
#include "common.h"

namespace {
constexpr int TS = 8;             // tile edge on the extended low-res grid
constexpr int WIN = 2 * TS + 6;   // edge of the hi-res window that feeds one tile

__host__ __device__ inline int up_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float up_coef(int a) { return (a == 0 || a == 3) ? 0.25f : 0.75f; }
}  // namespace

// ---- x_ext = replicate_pad(x, 1):  [n,H,W,C] (channel stride ldx) -> dense [n,H+2,W+2,C]
__global__ void __launch_bounds__(256) wdg_up2_pad_kernel(const float* __restrict__ x, int ldx, long long isx,
                                                          float* __restrict__ xe, int H, int W, int CQ, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % CQ);
        long long r = i / CQ;
        const int xx = (int)(r % (W + 2));
        r /= (W + 2);
        const int yy = (int)(r % (H + 2));
        const long long n = r / (H + 2);
        const int sy = up_clampi(yy - 1, 0, H - 1), sx = up_clampi(xx - 1, 0, W - 1);
        reinterpret_cast<f32x4*>(xe)[i] =
            *reinterpret_cast<const f32x4*>(x + n * isx + ((long long)sy * W + sx) * ldx + 4 * cq);
    }
}

// ---- dx = fold(dx_ext): the adjoint of the replicate pad (edge pixels collect the pad ring)
__global__ void __launch_bounds__(256) wdg_up2_fold_kernel(const float* __restrict__ dxe, float* __restrict__ dx, int lddx,
                                                           long long isdx, int H, int W, int CQ, int accumulate,
                                                           long long total) {
    const f32x4* src = reinterpret_cast<const f32x4*>(dxe);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i % CQ);
        long long r = i / CQ;
        const int xx = (int)(r % W);
        r /= W;
        const int yy = (int)(r % H);
        const long long n = r / H;
        const int y0 = yy == 0 ? 0 : yy + 1, y1 = yy == H - 1 ? H + 1 : yy + 1;
        const int x0 = xx == 0 ? 0 : xx + 1, x1 = xx == W - 1 ? W + 1 : xx + 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int ey = y0; ey <= y1; ++ey)
            for (int ex = x0; ex <= x1; ++ex) v += src[((n * (H + 2) + ey) * (W + 2) + ex) * CQ + cq];
        f32x4* dst = reinterpret_cast<f32x4*>(dx + n * isdx + ((long long)yy * W + xx) * lddx + 4 * cq);
        if (accumulate) v += *dst;
        *dst = v;
    }
}

// ---- column tensor of the output gradient: col[n, ry, rx, t*C + o] on the extended low-res grid
// One workgroup = an 8x8 tile of extended low-res pixels.  The 22x22 hi-res window of dy it needs is staged in LDS
// once; per tap row ty the vertical 4-tap sums (with the row masks) go to a second LDS array, the horizontal 4-tap
// sums are formed on the way out.  Writes are 16 bytes per lane, 5*C consecutive floats per (pixel, ty).
template <int CQ>
__global__ void __launch_bounds__(256) wdg_upconv_col_kernel(const float* __restrict__ dy, int ldy, long long isy,
                                                             float* __restrict__ col, int Hl, int Wl) {
    __shared__ f32x4 T[WIN * WIN * CQ];
    __shared__ f32x4 G[2][TS * WIN * CQ];
    const int He = Hl + 2, We = Wl + 2, H = 2 * Hl, W = 2 * Wl;
    const int tiles_x = (We + TS - 1) / TS;
    const int ry0 = (blockIdx.x / tiles_x) * TS, rx0 = (blockIdx.x % tiles_x) * TS;
    const long long n = blockIdx.y;
    const int y0 = 2 * ry0 - 5, x0 = 2 * rx0 - 5;   // hi-res origin of the window: row of (r = ry0 - 1, a = 0, ty = 0)
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
    float* dst = col + n * He * We * (100LL * CQ);
    for (int ty = 0; ty < 5; ++ty) {
        f32x4* Gt = G[ty & 1];
        for (int i = threadIdx.x; i < TS * WIN * CQ; i += 256) {
            const int o4 = i % CQ, hx = (i / CQ) % WIN, ryl = i / (CQ * WIN);
            const int q0 = 2 * (ry0 + ryl - 1) - 1;   // hi-res row of a = 0
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float w = (unsigned)(q0 + a) < (unsigned)H ? up_coef(a) : 0.f;
                acc += w * T[((2 * ryl + a + ty) * WIN + hx) * CQ + o4];
            }
            Gt[i] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TS * TS * 5 * CQ; i += 256) {
            const int j = i % (5 * CQ), pix = i / (5 * CQ);
            const int tx = j / CQ, o4 = j % CQ;
            const int rxl = pix % TS, ryl = pix / TS;
            const int ry = ry0 + ryl, rx = rx0 + rxl;
            if (ry >= He || rx >= We) continue;
            const int q0 = 2 * (rx - 1) - 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float w = (unsigned)(q0 + b) < (unsigned)W ? up_coef(b) : 0.f;
                acc += w * Gt[(ryl * WIN + 2 * rxl + b + tx) * CQ + o4];
            }
            *reinterpret_cast<f32x4*>(dst + ((long long)ry * We + rx) * (100LL * CQ) + (ty * 5 + tx) * (4 * CQ) + 4 * o4) = acc;
        }
        // the next iteration writes the other G buffer; its barrier orders this iteration's reads before the
        // writes of iteration ty + 2
    }
}

static int up_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

extern "C" int wdg_up2_pad(const float* x, int ldx, int64_t img_stride_x, float* xe, int n_img, int H, int W, int C,
                           wdg_stream stream) {
    WDG_CHECK_ARG(x && xe && n_img > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0, "bad argument");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)xe & 15) == 0, "x / xe must be 16-byte aligned");
    const long long total = (long long)n_img * (H + 2) * (W + 2) * (C / 4);
    hipLaunchKernelGGL(wdg_up2_pad_kernel, dim3(up_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (long long)img_stride_x, xe, H, W, C / 4, total);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_up2_fold(const float* dxe, float* dx, int lddx, int64_t img_stride_dx, int n_img, int H, int W, int C,
                            int accumulate, wdg_stream stream) {
    WDG_CHECK_ARG(dxe && dx && n_img > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0 && lddx % 4 == 0, "bad argument");
    WDG_CHECK_ARG(((uintptr_t)dxe & 15) == 0 && ((uintptr_t)dx & 15) == 0, "dxe / dx must be 16-byte aligned");
    const long long total = (long long)n_img * H * W * (C / 4);
    hipLaunchKernelGGL(wdg_up2_fold_kernel, dim3(up_blocks(total)), dim3(256), 0, (hipStream_t)stream, dxe, dx, lddx,
                       (long long)img_stride_dx, H, W, C / 4, accumulate, total);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_upconv_col_supported(int C) { return C == 4 || C == 8 || C == 16; }

extern "C" int wdg_upconv_col(const float* dy, int ldy, int64_t img_stride_dy, float* col, int n_img, int Hl, int Wl,
                              int C, wdg_stream stream) {
    WDG_CHECK_ARG(dy && col && n_img > 0 && n_img < 65536 && Hl > 0 && Wl > 0 && ldy % 4 == 0, "bad argument");
    WDG_CHECK_ARG(wdg_upconv_col_supported(C), "channel count must be 4, 8 or 16");
    WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)col & 15) == 0, "dy / col must be 16-byte aligned");
    const int tiles = ((Hl + 2 + TS - 1) / TS) * ((Wl + 2 + TS - 1) / TS);
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
