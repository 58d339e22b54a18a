// convlstm16.hip — the recurrent steps of the discriminator's 16-feature ConvLSTM2D (models.py:101) at n_timesteps > 1, fp32.
//
// One launch per timestep and direction, ~500 per train step at the shipped sequence length (T = 24), each with little work
// (1.4 GFLOP, 57 / 95 MB at batch 8) on a map that gives every CU two or three tiles which all run in the same phase at the same
// time: such a launch is a chain of memory round trips.  The general halo-tile kernel (conv_halo.hip) needed 33 / 44 us per
// forward / backward step against ~12 / 20 us of HBM time.  These kernels do the same arithmetic with everything that shapes
// the chain fixed at compile time (3 x 3 taps, 16 features, 4 x 32-pixel tiles) and the requests ordered by first use:
//   * the weights come in the exact LDS layout (wdg_convlstm16_pack: a straight coalesced 36.9 KB copy per workgroup instead of
//     2,304 scattered 16-byte gathers in three dependent batches),
//   * input halo and weights are requested first; the tile's own operands (the input part of the gates, c_{t-1} / the cell
//     backward's seven inputs), which are consumed in the epilogue only, go out behind them and arrive under the MFMAs,
//   * forward: the two 16-pixel fragments of a wave are finished one after the other, so the stores of the first drain under
//     the MFMAs of the second; backward: the next 16-channel slice of dgates is in flight while the current one is multiplied.
// MFMA mapping as in conv_halo.hip (operands swapped: rows = output channels, columns = 16 consecutive pixels of an image row):
// accumulator register r of lane (li = lane & 15, lg = lane >> 4) is output channel 16 b + 4 lg + r of pixel li — in the forward
// step column tile b is gate b, so a lane holds i, f, c~, o of four features of its pixel and updates the cell in registers.
#include "conv_plan.h"
#include <algorithm>
#include <cstring>

namespace {
constexpr int L_TH = 4, L_TW = 32;                 // output tile
constexpr int L_HH = L_TH + 2, L_HW = L_TW + 2;    // input halo
constexpr int L_NPR = L_HH * L_HW;                 // 204 halo pixels
constexpr int L_NPIX = 208;                        // ... padded to a multiple of 16 (conflict-free fragment reads)
constexpr int L_WQ = 9 * 4 * 64;                   // float4 slots of the staged weights (forward: [tap][kg][64]; backward: [ck][tap][kg][16])
constexpr size_t L_LDS = (size_t)(4 * L_NPIX + L_WQ) * sizeof(f32x4);   // 50.2 KB -> three workgroups per CU

struct WdgLstm16 {
    const float* A;          // forward: h_{t-1}; backward: dgates_t (64 channels)
    const f32x4* Wl;         // weights in LDS layout
    long long imgStrideA;
    int ldA, n_img, H, W, tiles_h, tiles_w;
    int ldc, ldh;
    // forward
    float* gates;            // [pixel][64]: in = input part of the pre-activations, out = complete pre-activations
    const float* c_prev;
    float* c_out;
    float* h_out;
    // backward
    float* dh_prev;          // [pixel][ld_dh]: in = gradient from the layers above, out = complete gradient of h_{t-1}
    long long imgStrideDh;
    int ld_dh;
    const float* gates_t;
    const float* c_cur;
    const float* dc_in;
    float* dgates_out;
    float* dc_out;
};

__device__ __forceinline__ float l_hs(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float l_hsg(float x) { const float v = 0.2f * x + 0.5f; return (v >= 0.f && v <= 1.f) ? 0.2f : 0.f; }

// the four halo slots of a thread (816 = 4 kg x 204 pixels over 256 threads): LDS slot and the pixel's offset in the image (-1: padding)
struct HaloSlots {
    int lds[4];
    int off[4];     // pixel index within the image, -1 outside
};
__device__ __forceinline__ HaloSlots l_halo_slots(int t, int hy0, int hx0, int H, int W) {
    HaloSlots s;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = u * 256 + t;
        const int kg = idx / L_NPR, pix = idx - kg * L_NPR;
        const int hy = pix / L_HW, hx = pix - hy * L_HW;
        const int gy = hy0 + hy, gx = hx0 + hx;
        const bool in = idx < 4 * L_NPR;
        // (slots beyond the halo go to a padding slot no fragment reads — 204..207 of a plane: an `if (slot >= 0)` around the
        // LDS store makes the compiler sink the slot's LOAD into that branch, behind everything else and with a full wait)
        s.lds[u] = in ? kg * L_NPIX + pix : 4 * L_NPIX - 1;
        s.off[u] = (in && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (gy * W + gx) * 4 + kg : -1;   // (pixel, kg) packed
    }
    return s;
}

// byte offset of a halo slot's 16 bytes within the image, 0x80000000 (beyond the descriptor's 2 GiB) for padding — by arithmetic
// on the sign, not a select (which comes out as a branch around the load)
__device__ __forceinline__ unsigned l_halo_byte_off(int off, int ld) {
    const unsigned neg = (unsigned)(off >> 31);
    return ((unsigned)(((off >> 2) * ld + (off & 3) * 4) * 4) & ~neg) | (neg & 0x80000000u);
}

__global__ void __launch_bounds__(256, 3) wdg_lstm16_fwd_kernel(const WdgLstm16 p) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    f32x4* lds_a = smem;                  // [4 kg][208 pixels]
    f32x4* lds_w = smem + 4 * L_NPIX;     // [9 taps][4 kg][64 gate columns]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h, img = bid / p.tiles_h;
    const int oy0 = ty * L_TH, ox0 = tx * L_TW;
    const float* Aimg = p.A + (long long)img * p.imgStrideA;

    // ---- requests in the order of first use: halo of h_{t-1}, weights, then the tile's own operands.  ALL of them branch-free
    // (buffer loads; padding and the ragged edge get bit 31 of their offset set arithmetically -> out of the descriptor's
    // range -> zeros): as `if (inside) v = load` / `inside ? load : 0` the compiler emitted exec-masked blocks with two full
    // s_waitcnt vmcnt(0) between them — three round trips in sequence where this chain of launches can afford one.
    const HaloSlots hs = l_halo_slots(t, oy0 - 1, ox0 - 1, p.H, p.W);
    const long long pimg = (long long)img * p.H * p.W;
    const wdg_srd srdA = wdg_make_srd(Aimg), srdG = wdg_make_srd(p.gates + pimg * 64), srdC = wdg_make_srd(p.c_prev + pimg * p.ldc);
    f32x4 hv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        hv[u] = wdg_buffer_load_f32x4(srdA, l_halo_byte_off(hs.off[u], p.ldA));
    f32x4 wv[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) wv[u] = p.Wl[u * 256 + t];
    const int oy = oy0 + wave;
    f32x4 old[2][4], cprev[2];
    bool ok[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ox = ox0 + a * 16 + li;
        ok[a] = oy < p.H && ox < p.W;
        const unsigned bad = (unsigned)((p.H - 1 - oy) | (p.W - 1 - ox)) & 0x80000000u;
        const int pl = oy * p.W + ox;
#pragma unroll
        for (int b = 0; b < 4; ++b) old[a][b] = wdg_buffer_load_f32x4(srdG, (unsigned)((pl * 64 + b * 16 + 4 * lg) * 4) | bad);
        cprev[a] = wdg_buffer_load_f32x4(srdC, (unsigned)((pl * p.ldc + 4 * lg) * 4) | bad);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) lds_w[u * 256 + t] = wv[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        lds_a[hs.lds[u]] = hv[u];
    __syncthreads();

#pragma unroll
    for (int a = 0; a < 2; ++a) {
        f32x4 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int th = tap / 3, tw = tap % 3;
            const f32x4 af = lds_a[lg * L_NPIX + (wave + th) * L_HW + a * 16 + li + tw];
            f32x4 bf[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = lds_w[(tap * 4 + lg) * 64 + b * 16 + li];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][j], af[j], acc[b], 0, 0, 0);
        }
        // ---- this fragment's epilogue: complete pre-activations (kept for the backward pass), cell update (Keras hard_sigmoid /
        // tanh: c = f c_prev + i c~, h = o tanh(c) — the arithmetic of wdg_lstm_fwd, pointwise.hip)
        if (ok[a]) {
            const long long pix = pimg + (long long)oy * p.W + ox0 + a * 16 + li;
            f32x4 v[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                v[b] = acc[b] + old[a][b];
                *reinterpret_cast<f32x4*>(p.gates + pix * 64 + b * 16 + 4 * lg) = v[b];
            }
            f32x4 cn, hn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cn[r] = l_hs(v[1][r]) * cprev[a][r] + l_hs(v[0][r]) * wdg_tanh(v[2][r]);
                hn[r] = l_hs(v[3][r]) * wdg_tanh(cn[r]);
            }
            *reinterpret_cast<f32x4*>(p.c_out + pix * p.ldc + 4 * lg) = cn;
            *reinterpret_cast<f32x4*>(p.h_out + pix * p.ldh + 4 * lg) = hn;
        }
    }
}

// backward step: dh_{t-1} += conv_transpose(dgates_t, W_h) — now the complete gradient of h_{t-1} — then the cell backward of
// timestep t-1 on it (gates_{t-1}, c_{t-2} (NULL at t-1 = 0), c_{t-1}, dc flowing in from t -> dgates_{t-1}, dc flowing on to t-2)
__global__ void __launch_bounds__(256, 3) wdg_lstm16_bwd_kernel(const WdgLstm16 p) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    f32x4* lds_a = smem;                  // [4 kg][208 pixels] of the current 16-channel slice of dgates_t
    f32x4* lds_w = smem + 4 * L_NPIX;     // [4 slices][9 taps][4 kg][16 features]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h, img = bid / p.tiles_h;
    const int oy0 = ty * L_TH, ox0 = tx * L_TW;
    const float* Aimg = p.A + (long long)img * p.imgStrideA;

    // (every request branch-free, see the forward kernel: here the compiler had put a full wait between the request of the
    // next dgates slice and the MFMAs that were meant to cover it)
    const HaloSlots hs = l_halo_slots(t, oy0 - 1, ox0 - 1, p.H, p.W);
    const long long pimg = (long long)img * p.H * p.W;
    const wdg_srd srdA = wdg_make_srd(Aimg), srdD = wdg_make_srd(p.dh_prev + (long long)img * p.imgStrideDh),
                  srdG = wdg_make_srd(p.gates_t + pimg * 64), srdCp = wdg_make_srd((p.c_prev ? p.c_prev : p.c_cur) + pimg * p.ldc),
                  srdCc = wdg_make_srd(p.c_cur + pimg * p.ldc), srdDc = wdg_make_srd(p.dc_in + pimg * p.ldc);
    unsigned hoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        hoff[u] = l_halo_byte_off(hs.off[u], p.ldA);          // (+ 64 ck below: padding stays at 0x80000000 + 64 ck, out of range)
    f32x4 hv[4];
    auto halo_request = [&](int ck) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) hv[u] = wdg_buffer_load_f32x4(srdA, hoff[u] + ck * 64);
    };
    halo_request(0);
    f32x4 wv[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) wv[u] = p.Wl[u * 256 + t];
    const int oy = oy0 + wave;
    f32x4 old[2], bin[2][7];
    bool ok[2];
    const unsigned no_cprev = p.c_prev ? 0u : 0x80000000u;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ox = ox0 + a * 16 + li;
        ok[a] = oy < p.H && ox < p.W;
        const unsigned bad = (unsigned)((p.H - 1 - oy) | (p.W - 1 - ox)) & 0x80000000u;
        const int pl = oy * p.W + ox;
        old[a] = wdg_buffer_load_f32x4(srdD, (unsigned)((pl * p.ld_dh + 4 * lg) * 4) | bad);
#pragma unroll
        for (int q = 0; q < 4; ++q) bin[a][q] = wdg_buffer_load_f32x4(srdG, (unsigned)((pl * 64 + q * 16 + 4 * lg) * 4) | bad);
        const unsigned co = (unsigned)((pl * p.ldc + 4 * lg) * 4) | bad;
        bin[a][4] = wdg_buffer_load_f32x4(srdCp, co | no_cprev);
        bin[a][5] = wdg_buffer_load_f32x4(srdCc, co);
        bin[a][6] = wdg_buffer_load_f32x4(srdDc, co);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) lds_w[u * 256 + t] = wv[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        lds_a[hs.lds[u]] = hv[u];
    __syncthreads();

    f32x4 acc[2];
    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ck = 0; ck < 4; ++ck) {
        if (ck < 3) halo_request(ck + 1);          // in flight under this slice's MFMAs
        __builtin_amdgcn_sched_barrier(0);         // (... which the scheduler otherwise moves behind most of them)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // data gradient: tap (th, tw) reads dgates at (y + 1 - th, x + 1 - tw) -> halo row wave + 2 - th, column + 2 - tw
            const int th = tap / 3, tw = tap % 3;
            const f32x4 bf = lds_w[((ck * 9 + tap) * 4 + lg) * 16 + li];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 af = lds_a[lg * L_NPIX + (wave + 2 - th) * L_HW + a * 16 + li + 2 - tw];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[j], acc[a], 0, 0, 0);
            }
        }
        if (ck < 3) {
            __syncthreads();                       // every wave is done with this slice
#pragma unroll
            for (int u = 0; u < 4; ++u)
                lds_a[hs.lds[u]] = hv[u];
            __syncthreads();
        }
    }

    // ---- epilogue: complete dh_{t-1}, then the cell backward (the arithmetic of wdg_lstm_bwd, pointwise.hip)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (!ok[a]) continue;
        const long long pl = (long long)oy * p.W + ox0 + a * 16 + li, pix = pimg + pl;
        const f32x4 dhv = acc[a] + old[a];
        *reinterpret_cast<f32x4*>(p.dh_prev + (long long)img * p.imgStrideDh + pl * p.ld_dh + 4 * lg) = dhv;
        const f32x4 xi = bin[a][0], xf = bin[a][1], xc = bin[a][2], xo = bin[a][3], cp = bin[a][4], cc = bin[a][5], dci = bin[a][6];
        f32x4 di, df, dcc, dob, dcp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gi = l_hs(xi[r]), gf = l_hs(xf[r]), gc = wdg_tanh(xc[r]), go = l_hs(xo[r]);
            const float tc = wdg_tanh(cc[r]);
            const float dc = dhv[r] * go * (1.f - tc * tc) + dci[r];
            di[r] = dc * gc * l_hsg(xi[r]);
            df[r] = dc * cp[r] * l_hsg(xf[r]);
            dcc[r] = dc * gi * (1.f - gc * gc);
            dob[r] = dhv[r] * tc * l_hsg(xo[r]);
            dcp[r] = dc * gf;
        }
        float* dg = p.dgates_out + pix * 64 + 4 * lg;
        *reinterpret_cast<f32x4*>(dg) = di;
        *reinterpret_cast<f32x4*>(dg + 16) = df;
        *reinterpret_cast<f32x4*>(dg + 32) = dcc;
        *reinterpret_cast<f32x4*>(dg + 48) = dob;
        if (p.dc_out) *reinterpret_cast<f32x4*>(p.dc_out + pix * p.ldc + 4 * lg) = dcp;
    }
}

// w: the recurrent kernel [3][3][16][64] (HWIO).  wl_fwd [tap][kg][n = gate column 0..63][j] = w[tap][4 kg + j][n];
// wl_bwd [ck][tap][kg][n = feature 0..15][j] = w[tap][n][16 ck + 4 kg + j]
__global__ void __launch_bounds__(256) wdg_lstm16_pack_kernel(const float* __restrict__ w, float* __restrict__ wl_fwd, float* __restrict__ wl_bwd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 9 * 16 * 64) return;
    {
        const int j = i & 3, n = (i >> 2) & 63, kg = (i >> 8) & 3, tap = i >> 10;
        wl_fwd[i] = w[(tap * 16 + 4 * kg + j) * 64 + n];
    }
    {
        const int j = i & 3, n = (i >> 2) & 15, kg = (i >> 6) & 3, r = i >> 8;
        const int tap = r % 9, ck = r / 9;
        wl_bwd[i] = w[(tap * 16 + n) * 64 + 16 * ck + 4 * kg + j];
    }
}

int g_lstm16_step = 1;      // wdg_set_tuning("lstm16_step", 0/1): 0 = the halo-tile kernel's cell epilogues (conv_halo.hip)
bool lstm16_geom(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    return g_lstm16_step && g.kh == 3 && g.kw == 3 && g.stride == 1 && g.pad_h == 1 && g.pad_w == 1 && g.Cin == 16 && g.Cout == 64 &&
           g.H == g.Ho && g.W == g.Wo && g.ldy == 64 && g.img_stride_y == (int64_t)g.H * g.W * 64 && g.ldx % 4 == 0 &&
           (long long)g.H * g.W * std::max(g.ldx, 64) * 4 < (1LL << 31);      // (per-image buffer descriptors: 2 GiB)
}
bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr, const void* e = nullptr,
               const void* f = nullptr, const void* g = nullptr, const void* h = nullptr) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f | (uintptr_t)g | (uintptr_t)h) & 15) == 0;
}
int lds_opt_in() {
    static bool done = false;
    if (!done) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_lstm16_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_LDS));
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_lstm16_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_LDS));
        done = true;
    }
    return WDG_OK;
}
}  // namespace

void wdg_lstm16_set_step(int v) { g_lstm16_step = v != 0; }

extern "C" int wdg_convlstm16_supported(const wdg_conv_plan* pl) { return pl && lstm16_geom(pl) ? 1 : 0; }

extern "C" int wdg_convlstm16_pack(const float* w_hwio, float* wl_fwd, float* wl_bwd, wdg_stream stream) {
    WDG_CHECK_ARG(w_hwio && wl_fwd && wl_bwd && aligned16(wl_fwd, wl_bwd), "bad argument");
    hipLaunchKernelGGL(wdg_lstm16_pack_kernel, dim3(36), dim3(256), 0, (hipStream_t)stream, w_hwio, wl_fwd, wl_bwd);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_convlstm16_step(const wdg_conv_plan* pl, const float* h_prev, const float* wl_fwd, float* gates, const float* c_prev,
                                   float* c_out, int ldc, float* h_out, int ldh, wdg_stream stream) {
    WDG_CHECK_ARG(pl && h_prev && wl_fwd && gates && c_prev && c_out && h_out && lstm16_geom(pl), "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= 16 && ldh >= 16 && ldc % 4 == 0 && ldh % 4 == 0 && aligned16(h_prev, wl_fwd, gates, c_prev, c_out, h_out),
                  "bad strides / alignment");
    WDG_CHECK_ARG((long long)pl->g.H * pl->g.W * ldc * 4 < (1LL << 31), "an image's cell state must stay below 2 GiB");
    if (int rc = lds_opt_in()) return rc;
    const wdg_conv_geom& g = pl->g;
    WdgLstm16 p;
    memset(&p, 0, sizeof(p));
    p.A = h_prev; p.Wl = reinterpret_cast<const f32x4*>(wl_fwd); p.imgStrideA = g.img_stride_x; p.ldA = g.ldx;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.tiles_h = (g.H + L_TH - 1) / L_TH; p.tiles_w = (g.W + L_TW - 1) / L_TW;
    p.ldc = ldc; p.ldh = ldh; p.gates = gates; p.c_prev = c_prev; p.c_out = c_out; p.h_out = h_out;
    hipLaunchKernelGGL(wdg_lstm16_fwd_kernel, dim3((unsigned)((long long)g.n_img * p.tiles_h * p.tiles_w)), dim3(256), L_LDS,
                       (hipStream_t)stream, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_convlstm16_bwd_step(const wdg_conv_plan* pl, const float* dgates_next, const float* wl_bwd, float* dh_prev,
                                       const float* gates_t, const float* c_prev, const float* c_cur, const float* dc_in,
                                       float* dgates_out, float* dc_out, int ldc, wdg_stream stream) {
    WDG_CHECK_ARG(pl && dgates_next && wl_bwd && dh_prev && gates_t && c_cur && dc_in && dgates_out && lstm16_geom(pl),
                  "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= 16 && ldc % 4 == 0 && aligned16(dgates_next, wl_bwd, dh_prev, gates_t, c_prev, c_cur, dc_in, dgates_out) &&
                      aligned16(dc_out), "bad strides / alignment");
    WDG_CHECK_ARG((long long)pl->g.H * pl->g.W * ldc * 4 < (1LL << 31), "an image's cell state must stay below 2 GiB");
    if (int rc = lds_opt_in()) return rc;
    const wdg_conv_geom& g = pl->g;
    WdgLstm16 p;
    memset(&p, 0, sizeof(p));
    p.A = dgates_next; p.Wl = reinterpret_cast<const f32x4*>(wl_bwd); p.imgStrideA = g.img_stride_y; p.ldA = g.ldy;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.tiles_h = (g.H + L_TH - 1) / L_TH; p.tiles_w = (g.W + L_TW - 1) / L_TW;
    p.ldc = ldc; p.dh_prev = dh_prev; p.imgStrideDh = g.img_stride_x; p.ld_dh = g.ldx;
    p.gates_t = gates_t; p.c_prev = c_prev; p.c_cur = c_cur; p.dc_in = dc_in; p.dgates_out = dgates_out; p.dc_out = dc_out;
    hipLaunchKernelGGL(wdg_lstm16_bwd_kernel, dim3((unsigned)((long long)g.n_img * p.tiles_h * p.tiles_w)), dim3(256), L_LDS,
                       (hipStream_t)stream, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
