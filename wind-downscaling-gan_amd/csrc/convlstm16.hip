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
#include "convlstm2.h"
#include <algorithm>
#include <cstring>
#include <mutex>

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

// backward step: dh_{t-1} += conv_transpose(dgates_t, W_h) — now the complete gradient of h_{t-1} — then the cell backward of
// timestep t-1 on it (gates_{t-1}, c_{t-2} (NULL at t-1 = 0), c_{t-1}, dc flowing in from t -> dgates_{t-1}, dc flowing on to t-2)
__global__ void __launch_bounds__(256, 3) wdg_lstm16_fwd_kernel(const WdgLstm16 p) {
    int bid = blockIdx.x;
#include "convlstm16_fwd_body.h"
}
__global__ void __launch_bounds__(256, 3) wdg_lstm16_bwd_kernel(const WdgLstm16 p) {
    int bid = blockIdx.x;
#include "convlstm16_bwd_body.h"
}
// One launch per timestep for BOTH recurrent layers of the discriminator (models.py:93 and :101 are independent chains of T - 1
// dependent steps): workgroups [0, n16) run the 16-feature step, the rest the two-feature layer's pixel-per-thread step
// (convlstm2.h) — its ~7-10 us launches were launch latency, here they fill the quarter of the chip's workgroup slots the 16-feature
// step leaves free.
__global__ void __launch_bounds__(256, 3) wdg_lstm16_cl2_fwd_kernel(const WdgLstm16 p, const WdgCl2F q, int n16) {
    if ((int)blockIdx.x >= n16) {
        wdg_convlstm2_fwd_body((long long)((int)blockIdx.x - n16) * 256 + threadIdx.x, q.h_prev, q.ldx, q.imgStrideX, q.wF, q.gates, q.c_prev,
                               q.c_out, q.ldc, q.h_out, q.ldh, q.n_img, q.H, q.W);
        return;
    }
    int bid = blockIdx.x;
#include "convlstm16_fwd_body.h"
}
__global__ void __launch_bounds__(256, 3) wdg_lstm16_cl2_bwd_kernel(const WdgLstm16 p, const WdgCl2B q, int n16) {
    if ((int)blockIdx.x >= n16) {
        wdg_convlstm2_bwd_body((long long)((int)blockIdx.x - n16) * 256 + threadIdx.x, q.dg_next, q.wD, q.dh_prev, q.ldx, q.imgStrideX, q.gates_t,
                               q.c_prev, q.c_cur, q.dc_in, q.dgates_out, q.dc_out, q.ldc, q.n_img, q.H, q.W);
        return;
    }
    int bid = blockIdx.x;
#include "convlstm16_bwd_body.h"
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
    // per device (the attribute belongs to the function ON a device) and safe against a concurrent first call from another host thread
    static std::mutex mtx;
    static bool done_dev[64] = {};
    int dev = 0;
    WDG_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mtx);
    bool& done = done_dev[dev & 63];
    if (!done) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_lstm16_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_LDS));
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_lstm16_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_LDS));
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_lstm16_cl2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_LDS));
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_lstm16_cl2_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_LDS));
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

// ---- both recurrent layers in one launch per timestep.  pl2 / the *2 arguments: the two-feature layer as for wdg_convlstm_step /
// wdg_convlstm_bwd_step (wF2 = its packed forward weights, wD2 = its data-gradient weights).
extern "C" int wdg_convlstm16_pair_supported(const wdg_conv_plan* pl16, const wdg_conv_plan* pl2) {
    return pl16 && pl2 && lstm16_geom(pl16) && wdg_lstm2_geom(pl2) ? 1 : 0;
}
extern "C" int wdg_convlstm16_pair_step(const wdg_conv_plan* pl, const float* h_prev, const float* wl_fwd, float* gates, const float* c_prev,
                                        float* c_out, int ldc, float* h_out, int ldh, const wdg_conv_plan* pl2, const float* h_prev2,
                                        const float* wF2, float* gates2, const float* c_prev2, float* c_out2, int ldc2, float* h_out2,
                                        int ldh2, wdg_stream stream) {
    WDG_CHECK_ARG(wdg_convlstm16_pair_supported(pl, pl2) && h_prev && wl_fwd && gates && c_prev && c_out && h_out && h_prev2 && wF2 && gates2 &&
                      c_prev2 && c_out2 && h_out2, "not supported for these geometries");
    WDG_CHECK_ARG(ldc >= 16 && ldh >= 16 && ldc % 4 == 0 && ldh % 4 == 0 && aligned16(h_prev, wl_fwd, gates, c_prev, c_out, h_out) &&
                      aligned16(h_prev2, gates2) && ldc2 >= 2 && ldh2 >= 2, "bad strides / alignment");
    WDG_CHECK_ARG((long long)pl->g.H * pl->g.W * ldc * 4 < (1LL << 31), "an image's cell state must stay below 2 GiB");
    if (int rc = lds_opt_in()) return rc;
    const wdg_conv_geom& g = pl->g;
    const wdg_conv_geom& g2 = pl2->g;
    WdgLstm16 p;
    memset(&p, 0, sizeof(p));
    p.A = h_prev; p.Wl = reinterpret_cast<const f32x4*>(wl_fwd); p.imgStrideA = g.img_stride_x; p.ldA = g.ldx;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.tiles_h = (g.H + L_TH - 1) / L_TH; p.tiles_w = (g.W + L_TW - 1) / L_TW;
    p.ldc = ldc; p.ldh = ldh; p.gates = gates; p.c_prev = c_prev; p.c_out = c_out; p.h_out = h_out;
    WdgCl2F q;
    memset(&q, 0, sizeof(q));
    q.h_prev = h_prev2; q.ldx = g2.ldx; q.imgStrideX = g2.img_stride_x; q.wF = wF2; q.gates = gates2; q.c_prev = c_prev2; q.c_out = c_out2;
    q.ldc = ldc2; q.h_out = h_out2; q.ldh = ldh2; q.n_img = g2.n_img; q.H = g2.H; q.W = g2.W;
    const long long n16 = (long long)g.n_img * p.tiles_h * p.tiles_w, n2 = ((long long)g2.n_img * g2.H * g2.W + 255) / 256;
    WDG_CHECK_ARG(n16 + n2 < (1LL << 31), "grid too large");
    hipLaunchKernelGGL(wdg_lstm16_cl2_fwd_kernel, dim3((unsigned)(n16 + n2)), dim3(256), L_LDS, (hipStream_t)stream, p, q, (int)n16);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_convlstm16_pair_bwd_step(const wdg_conv_plan* pl, const float* dgates_next, const float* wl_bwd, float* dh_prev,
                                            const float* gates_t, const float* c_prev, const float* c_cur, const float* dc_in,
                                            float* dgates_out, float* dc_out, int ldc, const wdg_conv_plan* pl2, const float* dgates_next2,
                                            const float* wD2, float* dh_prev2, const float* gates_t2, const float* c_prev2,
                                            const float* c_cur2, const float* dc_in2, float* dgates_out2, float* dc_out2, int ldc2,
                                            wdg_stream stream) {
    WDG_CHECK_ARG(wdg_convlstm16_pair_supported(pl, pl2) && dgates_next && wl_bwd && dh_prev && gates_t && c_cur && dc_in && dgates_out &&
                      dgates_next2 && wD2 && dh_prev2 && gates_t2 && c_cur2 && dc_in2 && dgates_out2, "not supported for these geometries");
    WDG_CHECK_ARG(ldc >= 16 && ldc % 4 == 0 && aligned16(dgates_next, wl_bwd, dh_prev, gates_t, c_prev, c_cur, dc_in, dgates_out) &&
                      aligned16(dc_out, dgates_next2, dgates_out2) && ldc2 >= 2, "bad strides / alignment");
    WDG_CHECK_ARG((long long)pl->g.H * pl->g.W * ldc * 4 < (1LL << 31), "an image's cell state must stay below 2 GiB");
    if (int rc = lds_opt_in()) return rc;
    const wdg_conv_geom& g = pl->g;
    const wdg_conv_geom& g2 = pl2->g;
    WdgLstm16 p;
    memset(&p, 0, sizeof(p));
    p.A = dgates_next; p.Wl = reinterpret_cast<const f32x4*>(wl_bwd); p.imgStrideA = g.img_stride_y; p.ldA = g.ldy;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.tiles_h = (g.H + L_TH - 1) / L_TH; p.tiles_w = (g.W + L_TW - 1) / L_TW;
    p.ldc = ldc; p.dh_prev = dh_prev; p.imgStrideDh = g.img_stride_x; p.ld_dh = g.ldx;
    p.gates_t = gates_t; p.c_prev = c_prev; p.c_cur = c_cur; p.dc_in = dc_in; p.dgates_out = dgates_out; p.dc_out = dc_out;
    WdgCl2B q;
    memset(&q, 0, sizeof(q));
    q.dg_next = dgates_next2; q.wD = wD2; q.dh_prev = dh_prev2; q.ldx = g2.ldx; q.imgStrideX = g2.img_stride_x; q.gates_t = gates_t2;
    q.c_prev = c_prev2; q.c_cur = c_cur2; q.dc_in = dc_in2; q.dgates_out = dgates_out2; q.dc_out = dc_out2; q.ldc = ldc2;
    q.n_img = g2.n_img; q.H = g2.H; q.W = g2.W;
    const long long n16 = (long long)g.n_img * p.tiles_h * p.tiles_w, n2 = ((long long)g2.n_img * g2.H * g2.W + 255) / 256;
    WDG_CHECK_ARG(n16 + n2 < (1LL << 31), "grid too large");
    hipLaunchKernelGGL(wdg_lstm16_cl2_bwd_kernel, dim3((unsigned)(n16 + n2)), dim3(256), L_LDS, (hipStream_t)stream, p, q, (int)n16);
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
