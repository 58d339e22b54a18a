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

// vector LDS fragment (NT consecutive columns of one pixel)
template <int W>
struct WdgFragT {
    float v[W];
};
template <int W>
__device__ __forceinline__ WdgFragT<W> wdg_lds_frag_t(const float* q) {
    WdgFragT<W> f;
    if constexpr (W == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(q);
        f.v[0] = t[0]; f.v[1] = t[1]; f.v[2] = t[2]; f.v[3] = t[3];
    } else if constexpr (W == 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 t = *reinterpret_cast<const f32x2*>(q);
        f.v[0] = t[0]; f.v[1] = t[1];
    } else {
        f.v[0] = *q;
    }
    return f;
}

// ------------------------------------------------------------------------------------------------------------
// Thin 3x3 'same' weight gradient (Cin <= 8 or Cin = 16; Cout <= 64): the HBM-bound full-resolution layers of
// the discriminator front end (ConvLSTM 2->8 / 5->64 gate tensors, conv 2->16, 16->16) and the 16->2 output conv.
// x (16-64 B per pixel) and dy (16-256 B per pixel) are each read ONCE; everything else is arranged so that the
// matrix work does not exceed that HBM time:
//   * rows of the MFMA are the flattened (tap, ci) pairs — 9*Cin = 18 / 45 / 144 rows -> 2 / 3 / 9 row tiles
//     instead of 9 tiles of 16 zero-padded channels; one spare row reads a constant 1, which makes the bias
//     gradient (column sum of dy: the ConvLSTM's separate 0.17 ms colsum pass) a by-product;
//   * dy fragments are one vector LDS read of NT columns per lane (column-permuted tiles, as wdg_wgrad_kernel);
//   * the four waves of a block are summed through LDS before the slab write (4x smaller slab and reduce).
// ------------------------------------------------------------------------------------------------------------
struct WdgWgradThin {
    const float* X;
    const float* DY;
    float* partial;     // [nblocks][TM*16][NT*16]
    float* dW;
    float* dbias;
    long long imgStrideX, imgStrideY;
    int n_img, H, W, ldx, ldy;
    int Cin, CS, Cout, Cout_p;   // CS = staged x channels (Cin rounded up to 4)
    int bias_row;                // row that reads the constant 1 (or -1)
    int tiles_h, tiles_w, ntiles;
    int nblocks, accumulate;
    wdg_fastdiv div_yc4;
};

template <int TM, int NT, int CS>
__global__ void __launch_bounds__(256) wdg_wgrad_thin_kernel(const WdgWgradThin p) {
    constexpr int TH = NT == 4 ? 4 : 8;      // tile rows (x 32 columns)
    constexpr int HW = 34, HH = TH + 2;
    constexpr int COS = NT * 16;
    constexpr int RPW = TH / 4;              // tile rows per wave
    extern __shared__ __attribute__((aligned(16))) float smem_t[];
    float* xs = smem_t;                      // [HH*HW][CS]
    float* ones = xs + HH * HW * CS;         // [HH*HW][CS] of 1.0 (CS < 16 only)
    float* ys = ones + (CS < 16 ? HH * HW * CS : 0);   // [TH*32][COS]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 15, lg = lane >> 4;

    // per-lane LDS word offset of row (tap, ci) relative to the output pixel's halo slot; the bias row points into
    // the ones array; padding rows alias slot 0 (their products land in accumulator rows that are never read)
    int aoff[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = tm * 16 + li;
        if (r < 9 * p.Cin) {
            const int tap = r / p.Cin, ci = r - tap * p.Cin;
            aoff[tm] = ((tap / 3) * HW + tap % 3) * CS + ci;
        } else {
            aoff[tm] = (r == p.bias_row && CS < 16) ? HH * HW * CS : 0;
        }
        aoff[tm] += (wave * RPW * HW + lg) * CS;
    }
    const int boff = (wave * RPW * 32 + lg) * COS + NT * li;

    f32x4 acc[TM][NT];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int idx = t; idx < TH * 32 * COS; idx += 256) ys[idx] = 0.f;   // columns >= Cout_p stay zero
    if (CS < 16)
        for (int idx = t; idx < HH * HW * CS; idx += 256) ones[idx] = 1.f;
    // staging through registers, one tile ahead: the global loads of tile i+1 are in flight while tile i is
    // reduced (x: up to 6 float4 per thread at 16 channels, dy: NY float4)
    constexpr int xsh = CS == 16 ? 2 : CS == 8 ? 1 : 0;      // log2(channel groups of x)
    const int yc4 = p.Cout_p >> 2;
    constexpr int NX = (HH * HW * (CS / 4) + 255) / 256;
    constexpr int NY = (TH * 32 * NT * 4 + 255) / 256;
    const int nx_items = (HH * HW) << xsh, ny_items = TH * 32 * yc4;
    f32x4 rx[NX], ry[NY];
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = b % p.tiles_w;
        b /= p.tiles_w;
        const int ty = b % p.tiles_h;
        const int img = b / p.tiles_h;
        const int oy0 = ty * TH, ox0 = tx * 32;
        const float* Ximg = p.X + (long long)img * p.imgStrideX;
        const float* Yimg = p.DY + (long long)img * p.imgStrideY;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = t + 256 * i;
            const int c4 = idx & ((1 << xsh) - 1);
            const int pix = idx >> xsh;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < nx_items && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = *reinterpret_cast<const f32x4*>(Ximg + ((long long)gy * p.W + gx) * p.ldx + 4 * c4);
            rx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int idx = t + 256 * i;
            const int pix = (int)wdg_fastdiv_do((unsigned)idx, p.div_yc4);
            const int c4 = idx - pix * yc4;
            const int gy = oy0 + (pix >> 5), gx = ox0 + (pix & 31);
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < ny_items && gy < p.H && gx < p.W)
                v = *reinterpret_cast<const f32x4*>(Yimg + ((long long)gy * p.W + gx) * p.ldy + 4 * c4);
            ry[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = t + 256 * i;
            if (idx < nx_items) *reinterpret_cast<f32x4*>(&xs[(idx >> xsh) * CS + 4 * (idx & ((1 << xsh) - 1))]) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int idx = t + 256 * i;
            const int pix = (int)wdg_fastdiv_do((unsigned)idx, p.div_yc4);
            const int c4 = idx - pix * yc4;
            if (idx < ny_items) *reinterpret_cast<f32x4*>(&ys[pix * COS + 4 * c4]) = ry[i];
        }
    };

    if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();   // previous tile consumed (and the initial fills done)
        store_tile();
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x);
        constexpr int UNR = TM * NT >= 9 ? 2 : 8;   // bound the live fragment registers of the wide variants
#pragma unroll
        for (int ry_ = 0; ry_ < RPW; ++ry_) {
#pragma unroll UNR
            for (int s = 0; s < 8; ++s) {
                // compile-time offsets: every address below is lane base + immediate
                const WdgFragT<NT> bf = wdg_lds_frag_t<NT>(&ys[boff + (ry_ * 32 + 4 * s) * COS]);
                float af[TM];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) af[tm] = xs[aoff[tm] + (ry_ * HW + 4 * s) * CS];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int b2 = 0; b2 < NT; ++b2)
                        acc[tm][b2] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm], bf.v[b2], acc[tm][b2], 0, 0, 0);
            }
        }
    }
    // ---- block reduction of the four waves, then one slab entry per block
    __syncthreads();
    float* red = smem_t;   // [4][TM*16][COS]  (fits: checked on the host)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int b2 = 0; b2 < NT; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * TM * 16 + tm * 16 + 4 * lg + r) * COS + NT * li + b2] = acc[tm][b2][r];
    __syncthreads();
    float* dst = p.partial + (long long)blockIdx.x * TM * 16 * COS;
    for (int idx = t; idx < TM * 16 * COS; idx += 256)
        dst[idx] = (red[idx] + red[TM * 16 * COS + idx]) + (red[2 * TM * 16 * COS + idx] + red[3 * TM * 16 * COS + idx]);
}

// slab [nblocks][R_p][COS] -> dW[tap][ci][co] (+ dbias[co] from the ones-row), fixed summation order
__global__ void __launch_bounds__(256) wdg_wgrad_thin_reduce_kernel(const WdgWgradThin p, int R_p, int COS) {
    __shared__ float red[256];
    const int per = R_p * COS;
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (int base = blockIdx.x * 16; base < per; base += gridDim.x * 16) {
        const int idx = base + el;
        float v = 0.f;
        if (idx < per)
            for (int s = sl; s < p.nblocks; s += 16) v += p.partial[(long long)s * per + idx];
        red[threadIdx.x] = v;
        __syncthreads();
        if (sl == 0 && idx < per) {
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tsum += red[k * 16 + el];
            const int row = idx / COS, co = idx - row * COS;
            if (co < p.Cout) {
                float* d = nullptr;
                bool acc = p.accumulate != 0;
                if (row < 9 * p.Cin) d = p.dW + (long long)row * p.Cout + co;       // row = tap*Cin + ci
                else if (row == p.bias_row && p.dbias) { d = p.dbias + co; acc = true; }
                if (d) {
                    if (acc) tsum += *d;
                    *d = tsum;
                }
            }
        }
        __syncthreads();
    }
}

static int thin_tm(int Cin) { return (9 * Cin + 15) / 16; }
static bool thin_tm_built(int tm) { return tm == 1 || tm == 2 || tm == 3 || tm == 4 || tm == 5 || tm == 9; }
static int thin_nt(int Cout) { return Cout <= 16 ? 1 : Cout <= 32 ? 2 : Cout <= 64 ? 4 : 0; }
static int g_thin_enable = 1;
void wdg_wgrad_thin_enable(int v) { g_thin_enable = v != 0; }

int wdg_wgrad_thin_eligible(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    if (!g_thin_enable) return 0;
    if (g.stride != 1 || g.kh != 3 || g.kw != 3 || g.pad_h != 1 || g.pad_w != 1 || g.Cin > 16) return 0;
    if ((long long)g.n_img * g.Ho * g.Wo < 65536) return 0;
    return thin_tm_built(thin_tm(g.Cin)) && thin_nt(g.Cout) != 0;
}
// can the bias gradient ride along (a spare row in the last row tile)?
int wdg_wgrad_thin_has_bias_row(const wdg_conv_plan* pl) { return wdg_wgrad_thin_eligible(pl) && (9 * pl->g.Cin) % 16 != 0; }

static int thin_blocks(const wdg_conv_plan* pl, int nt) {
    const wdg_conv_geom& g = pl->g;
    const int th = nt == 4 ? 4 : 8;
    const long long ntiles = (long long)g.n_img * ((g.Ho + th - 1) / th) * ((g.Wo + 31) / 32);
    return (int)std::min<long long>(ntiles, (long long)pl->cus * 4);
}
size_t wdg_wgrad_thin_ws_bytes(const wdg_conv_plan* pl) {
    if (!wdg_wgrad_thin_eligible(pl)) return 0;
    const int nt = thin_nt(pl->g.Cout), tm = thin_tm(pl->g.Cin);
    return (size_t)thin_blocks(pl, nt) * tm * 16 * nt * 16 * sizeof(float);
}

int wdg_wgrad_thin_launch(const wdg_conv_plan* pl, const float* x, const float* dy, float* dw, float* dbias,
                          int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const wdg_conv_geom& g = pl->g;
    const int nt = thin_nt(g.Cout), tm = thin_tm(g.Cin);
    const int th = nt == 4 ? 4 : 8;
    WdgWgradThin p;
    memset(&p, 0, sizeof(p));
    p.X = x; p.DY = dy; p.dW = dw; p.dbias = dbias;
    p.imgStrideX = g.img_stride_x; p.imgStrideY = g.img_stride_y;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldx = g.ldx; p.ldy = g.ldy;
    p.Cin = g.Cin; p.CS = pl->Cin_p; p.Cout = g.Cout; p.Cout_p = pl->Cout_p;
    p.bias_row = (dbias && wdg_wgrad_thin_has_bias_row(pl)) ? 9 * g.Cin : -1;
    p.tiles_h = (g.Ho + th - 1) / th;
    p.tiles_w = (g.Wo + 31) / 32;
    p.ntiles = g.n_img * p.tiles_h * p.tiles_w;
    p.nblocks = thin_blocks(pl, nt);
    p.accumulate = accumulate;
    p.div_yc4 = wdg_fastdiv_make((unsigned)(pl->Cout_p / 4));
    const size_t need = (size_t)p.nblocks * tm * 16 * nt * 16 * sizeof(float);
    if (!ws || ws_bytes < need) {
        wdg_set_error("wgrad_thin: workspace too small (%zu < %zu)", ws_bytes, need);
        return WDG_ERR_WORKSPACE;
    }
    p.partial = (float*)ws;
    const size_t stage = ((size_t)(th + 2) * 34 * p.CS * (p.CS < 16 ? 2 : 1) + (size_t)th * 32 * nt * 16) * sizeof(float);
    const size_t red = (size_t)4 * tm * 16 * nt * 16 * sizeof(float);
    const size_t lds = std::max(stage, red);
    dim3 grid(p.nblocks), block(256);
    bool launched = false;
#define WDG_THIN_CASE(TM_, NT_, CS_)                                                                          \
    if (tm == TM_ && nt == NT_ && p.CS == CS_) {                                                              \
        static bool attr = false;                                                                             \
        if (!attr && lds > 48 * 1024) {                                                                       \
            WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_wgrad_thin_kernel<TM_, NT_, CS_>),  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));              \
            attr = true;                                                                                      \
        }                                                                                                     \
        hipLaunchKernelGGL((wdg_wgrad_thin_kernel<TM_, NT_, CS_>), grid, block, lds, st, p);                  \
        launched = true;                                                                                      \
    }
#define WDG_THIN_ROW(TM_, CS_) WDG_THIN_CASE(TM_, 1, CS_) WDG_THIN_CASE(TM_, 2, CS_) WDG_THIN_CASE(TM_, 4, CS_)
    WDG_THIN_ROW(1, 4) WDG_THIN_ROW(2, 4) WDG_THIN_ROW(3, 4) WDG_THIN_ROW(3, 8) WDG_THIN_ROW(4, 8) WDG_THIN_ROW(5, 8)
    WDG_THIN_ROW(9, 16)
#undef WDG_THIN_ROW
#undef WDG_THIN_CASE
    if (!launched) {
        wdg_set_error("wgrad_thin: no kernel for tm=%d nt=%d cs=%d", tm, nt, p.CS);
        return WDG_ERR_ARG;
    }
    WDG_LAUNCH_CHECK();
    const int per = tm * 16 * nt * 16;
    hipLaunchKernelGGL(wdg_wgrad_thin_reduce_kernel, dim3(std::min((per + 15) / 16, 1024)), block, 0, st, p, tm * 16, nt * 16);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
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
