// wgrad_halo.hip — weight gradient of stride-1 convolutions with few channels on large maps
// (Cin <= 16, Cout <= 64: the discriminator's full-resolution ConvLSTM / conv layers, models.py:93-104,
// and the generator's 16->2 output conv, :70).
//
// dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co] is HBM-bound for these shapes (a few
// hundred outputs reduced over millions of pixels).  The generic pixel-reduction kernel re-reads x once
// per tap; here a block walks 4x32 pixel tiles, stages the x halo and the dy tile ONCE in LDS and keeps
// all taps' accumulators in registers (taps x Cout/16 MFMA tiles), so x and dy are read exactly once.
// MFMA view per tap: C[ci][co] += A[ci][k] * B[k][co] with k = pixel.  Per-wave partials go to a slab
// and are summed in a fixed order by wdg_wgrad_halo_reduce_kernel (reproducible).
#include "conv_plan.h"
#include <algorithm>

constexpr int WH_TH = 4, WH_TW = 32;

struct WdgWgradHalo {
    const float* X;
    const float* DY;
    float* dW;
    float* partial;  // [nblocks*4][taps][16][NT*16]
    long long imgStrideX, imgStrideY;
    int n_img, H, W, ldx;   // x dims (== dy dims: stride 1, 'same' or valid geometry handled by pad)
    int Ho, Wo, ldy;
    int Cin, Cout, Cin4, Cout4;   // logical channels and padded channel groups
    int kh, kw, pad_h, pad_w;
    int halo_w, tiles_h, tiles_w, ntiles;
    int nparts, accumulate;
    int nchunks;   // column chunks of NT*16 output channels (blockIdx.y)
};

template <int TAPS, int NT>
__global__ void __launch_bounds__(256) wdg_wgrad_halo_kernel(const WdgWgradHalo p) {
    constexpr int RSY = NT * 16 + ((NT * 16) % 32 == 0 ? 16 : 0);
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int halo_h = WH_TH + p.kh - 1;
    const int npr = halo_h * p.halo_w;
    float* xh = smem_f;              // [npr][16]
    float* dyt = smem_f + npr * 16;  // [128][RSY]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int co0 = blockIdx.y * NT * 16;                       // first output channel of this chunk
    const int c4lim = min(NT * 4, p.Cout4 - blockIdx.y * NT * 4);   // staged channel groups of dy

    f32x4 acc[TAPS][NT];
#pragma unroll
    for (int a = 0; a < TAPS; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // zero the channel rows that are never staged (ci >= Cin_p): read by the MFMA, discarded later
    for (int idx = t; idx < npr * 16; idx += 256) xh[idx] = 0.f;
    for (int idx = t; idx < 128 * RSY; idx += 256) dyt[idx] = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int b = tile;
        const int tx = b % p.tiles_w;
        b /= p.tiles_w;
        const int ty = b % p.tiles_h;
        const int img = b / p.tiles_h;
        const int oy0 = ty * WH_TH, ox0 = tx * WH_TW;
        __syncthreads();  // previous tile's reads are done (and the zero fill on the first pass)
        // ---- stage x halo: pixel (oy0 - pad + hy, ox0 - pad + hx)
        const float* Ximg = p.X + (long long)img * p.imgStrideX;
        for (int idx = t; idx < npr * p.Cin4; idx += 256) {
            const int c4 = idx % p.Cin4;
            const int pix = idx / p.Cin4;
            const int hy = pix / p.halo_w, hx = pix - hy * p.halo_w;
            const int gy = oy0 - p.pad_h + hy, gx = ox0 - p.pad_w + hx;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const f32x4*>(Ximg + ((long long)gy * p.W + gx) * p.ldx + 4 * c4);
            *reinterpret_cast<f32x4*>(&xh[pix * 16 + 4 * c4]) = v;
        }
        // ---- stage dy tile
        const float* Yimg = p.DY + (long long)img * p.imgStrideY;
        for (int idx = t; idx < 128 * c4lim; idx += 256) {
            const int c4 = idx % c4lim;
            const int pix = idx / c4lim;
            const int py = pix >> 5, px = pix & 31;
            const int gy = oy0 + py, gx = ox0 + px;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy < p.Ho && gx < p.Wo)
                v = *reinterpret_cast<const f32x4*>(Yimg + ((long long)gy * p.Wo + gx) * p.ldy + co0 + 4 * c4);
            *reinterpret_cast<f32x4*>(&dyt[pix * RSY + 4 * c4]) = v;
        }
        __syncthreads();
        // ---- wave `wave` reduces tile row `wave`: 8 steps of 4 pixels, all taps
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {
            const int px = 4 * s + lg;
            float bf[NT];
#pragma unroll
            for (int b2 = 0; b2 < NT; ++b2) bf[b2] = dyt[(wave * 32 + px) * RSY + b2 * 16 + li];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                constexpr int KW = TAPS == 9 ? 3 : 5;
                const int th = tap / KW, tw = tap - th * KW;
                const float af = xh[((wave + th) * p.halo_w + px + tw) * 16 + li];
#pragma unroll
                for (int b2 = 0; b2 < NT; ++b2)
                    acc[tap][b2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[b2], acc[tap][b2], 0, 0, 0);
            }
        }
    }
    // ---- per-wave partial: [part][tap][ci = 4*lg + r][co = b*16 + li]
    float* dst = p.partial + (((long long)blockIdx.y * p.nparts + blockIdx.x * 4 + wave) * TAPS) * 16 * (NT * 16);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int b2 = 0; b2 < NT; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[(tap * 16 + 4 * lg + r) * (NT * 16) + b2 * 16 + li] = acc[tap][b2][r];
}

__global__ void __launch_bounds__(256) wdg_wgrad_halo_reduce_kernel(const WdgWgradHalo p, int taps, int nw) {
    __shared__ float red[256];
    const long long per = (long long)taps * 16 * nw;
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const float* part = p.partial + (long long)blockIdx.y * p.nparts * per;
    for (long long base = (long long)blockIdx.x * 16; base < per; base += (long long)gridDim.x * 16) {
        const long long idx = base + el;
        float v = 0.f;
        if (idx < per)
            for (int s = sl; s < p.nparts; s += 16) v += part[(long long)s * per + idx];
        red[threadIdx.x] = v;
        __syncthreads();
        if (sl == 0 && idx < per) {
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tsum += red[k * 16 + el];
            const int co = (int)(idx % nw) + blockIdx.y * nw;
            const int ci = (int)((idx / nw) % 16);
            const int tap = (int)(idx / ((long long)nw * 16));
            if (ci < p.Cin && co < p.Cout) {
                float* d = p.dW + ((long long)tap * p.Cin + ci) * p.Cout + co;
                if (p.accumulate) tsum += *d;
                *d = tsum;
            }
        }
        __syncthreads();
    }
}

static const int WH_BLOCKS = 512;

// 0 if the plan cannot use this kernel, else the number of 16-column tiles per block (column chunks of
// NT*16 output channels run as blockIdx.y, so wide dy tensors such as the 160-channel upsampled input of
// the 5x5 transposed conv are covered with the x halo re-read once per chunk)
int wdg_wgrad_halo_eligible(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    if (g.stride != 1 || g.Cin > 16) return 0;
    if ((long long)g.n_img * g.Ho * g.Wo < 65536) return 0;
    const int taps = g.kh * g.kw;
    if (taps == 9 && g.kh == 3) return g.Cout <= 16 ? 1 : g.Cout <= 32 ? 2 : 4;
    if (taps == 25 && g.kh == 5) return g.Cout <= 16 ? 1 : 2;
    return 0;
}
static int wh_chunks(const wdg_conv_plan* pl, int nt) { return (pl->Cout_p + nt * 16 - 1) / (nt * 16); }

size_t wdg_wgrad_halo_ws_bytes(const wdg_conv_plan* pl) {
    const int nt = wdg_wgrad_halo_eligible(pl);
    if (!nt) return 0;
    return (size_t)wh_chunks(pl, nt) * WH_BLOCKS * 4 * pl->taps * 16 * nt * 16 * sizeof(float);
}

int wdg_wgrad_halo_launch(const wdg_conv_plan* pl, const float* x, const float* dy, float* dw, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st) {
    const wdg_conv_geom& g = pl->g;
    const int nt = wdg_wgrad_halo_eligible(pl);
    WdgWgradHalo p;
    memset(&p, 0, sizeof(p));
    p.X = x; p.DY = dy; p.dW = dw;
    p.imgStrideX = g.img_stride_x; p.imgStrideY = g.img_stride_y;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldx = g.ldx;
    p.Ho = g.Ho; p.Wo = g.Wo; p.ldy = g.ldy;
    p.Cin = g.Cin; p.Cout = g.Cout; p.Cin4 = pl->Cin_p / 4; p.Cout4 = pl->Cout_p / 4;
    p.kh = g.kh; p.kw = g.kw; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
    p.halo_w = WH_TW + g.kw - 1;
    p.tiles_h = (g.Ho + WH_TH - 1) / WH_TH;
    p.tiles_w = (g.Wo + WH_TW - 1) / WH_TW;
    p.ntiles = g.n_img * p.tiles_h * p.tiles_w;
    const int nblocks = std::min(WH_BLOCKS, p.ntiles);
    p.nparts = nblocks * 4;
    p.accumulate = accumulate;
    p.nchunks = wh_chunks(pl, nt);
    const size_t need = (size_t)p.nchunks * p.nparts * pl->taps * 16 * nt * 16 * sizeof(float);
    if (!ws || ws_bytes < need) {
        wdg_set_error("wgrad_halo: workspace too small (%zu < %zu)", ws_bytes, need);
        return WDG_ERR_WORKSPACE;
    }
    p.partial = (float*)ws;
    const int rsy = nt * 16 + ((nt * 16) % 32 == 0 ? 16 : 0);
    const size_t lds = ((size_t)(WH_TH + g.kh - 1) * p.halo_w * 16 + 128 * rsy) * sizeof(float);
    dim3 grid(nblocks, p.nchunks), block(256);
    if (pl->taps == 9 && nt == 1)
        hipLaunchKernelGGL((wdg_wgrad_halo_kernel<9, 1>), grid, block, lds, st, p);
    else if (pl->taps == 9 && nt == 2)
        hipLaunchKernelGGL((wdg_wgrad_halo_kernel<9, 2>), grid, block, lds, st, p);
    else if (pl->taps == 9 && nt == 4)
        hipLaunchKernelGGL((wdg_wgrad_halo_kernel<9, 4>), grid, block, lds, st, p);
    else if (nt == 1)
        hipLaunchKernelGGL((wdg_wgrad_halo_kernel<25, 1>), grid, block, lds, st, p);
    else
        hipLaunchKernelGGL((wdg_wgrad_halo_kernel<25, 2>), grid, block, lds, st, p);
    WDG_LAUNCH_CHECK();
    const long long per = (long long)pl->taps * 16 * nt * 16;
    const int rblocks = (int)std::min<long long>((per + 15) / 16, 4096);
    hipLaunchKernelGGL(wdg_wgrad_halo_reduce_kernel, dim3(rblocks, p.nchunks), block, 0, st, p, pl->taps, nt * 16);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
