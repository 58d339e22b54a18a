// convlstm1.hip — single-timestep ConvLSTM2D for few channels, fused and gate-recomputing.
//
// The discriminator starts with two ConvLSTM2D layers at FULL resolution on 2 and 5 input channels
// (/root/reference/src/downscaling/gan/models.py:93,101).  With n_timesteps = 1 the recurrence vanishes
// (h_0 = c_0 = 0): h = hs(o) * tanh(hs(i) * tanh(c~)), the forget gate and the recurrent kernel play no role.
// Unfused, the layer writes the gate pre-activations (64 channels for 16 features), re-reads them for the
// cell, stores them for the backward pass, and the backward pass writes and re-reads dgates: ~1.5 KB of HBM
// traffic per pixel for a layer whose input is 32 bytes per pixel.  Here
//   * forward:  x -> h in one kernel (plain fp32 FMAs with wave-uniform weights; for fp32 the vector unit has
//               the same peak as the MFMA unit and needs no padding of K = 9*Cin = 18 / 45 to the MFMA shape);
//   * backward: recomputes the gates from x, forms dgates, and produces dx from an LDS tile of dgates
//               (+ optionally the dense dgates tensor that the weight-gradient kernel consumes).
// Numerics are those of wdg_conv_fwd + wdg_lstm_fwd / wdg_lstm_bwd (same fp32 operations, different
// summation order).
#include "common.h"
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));
// two features per instruction: v_pk_fma_f32 (x broadcast to both halves, weights as an SGPR pair) doubles the
// fp32 FMA rate of the vector unit
__device__ __forceinline__ f32x2 cl_pk_fma(float x, f32x2 w, f32x2 acc) {
    return __builtin_elementwise_fma((f32x2){x, x}, w, acc);
}

constexpr int CL_TH = 4, CL_TW = 32;   // backward tile: 128 centre pixels, 2 threads (feature halves) per pixel

// tanh(x) = 1 - 2 / (1 + exp(2x)) in five instructions (v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma): the cell evaluates two per
// feature and pixel, and on the fp32 path the matrix instructions share the vector unit's multipliers, so every vector
// instruction of the cell is paid in full.  ABSOLUTE error <= ~2.5e-7 (one ulp each in exp and rcp around r = 0.5); the relative
// error grows as |x| -> 0 (the difference 1 - 2r cancels), which the cell does not care about: tanh enters h = o * tanh(c) and
// the factors (1 - tanh^2) of the backward pass by its value.  Saturates cleanly: exp -> inf gives 1, exp -> 0 gives -1.
// (The 12-instruction form it replaces — odd Taylor polynomial below |x| = 0.1, (1 - e) / (1 + e) above — held 5e-7 RELATIVE.)
__device__ __forceinline__ float cl_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // exp(2x) = 2^(2x log2 e)
    return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + e), 1.f);
}

__device__ __forceinline__ float cl_hsig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float cl_hsig_grad(float x) {
    const float v = 0.2f * x + 0.5f;
    return (v >= 0.f && v <= 1.f) ? 0.2f : 0.f;
}

struct WdgCl1 {
    const float* X;      // [n_img,H,W,ldx], CIN logical channels
    const float* Wx;     // HWIO [3][3][CIN][4F]
    const float* bias;   // [4F]
    const float* dH;     // [.., lddh] (backward)
    float* Hout;         // forward output [.., ldh]
    float* dG;           // optional dense dgates [P][4F] (forget-gate slots zero)
    float* dWpart;       // fused weight gradient (WG kernels): per-block partial [gridDim.x][RT*16][CT*16]
    float* dX;           // optional input gradient [.., lddx]
    long long imgStrideX, imgStrideH, imgStrideDH, imgStrideDX;
    int n_img, H, W, ldx, ldh, lddh, lddx;
    int accumulate_dx;
    int tiles_h, tiles_w;
    // second source of the LAST x2_n input channels (5-channel layer only): logical channel c >= CIN - x2_n is read from
    // X2[img][pixel][c - (CIN - x2_n)] instead of X.  The discriminator's low + high ConvLSTM (models.py:100-101) reads the
    // concatenation [low | high]: low is constant over a train step and lives in X, high changes with every pass and is read
    // in place from the tensor the caller holds (12 two-channel copies into the concatenation per step otherwise).
    const float* X2;
    long long imgStrideX2;
    int ldx2, x2_n;
};

// gate pre-activations (i, c~, o) of FH features starting at f0 for one pixel, x read through `load`
// Wx / bias are separate `const __restrict__` kernel arguments: only then does hipcc treat the (wave-uniform)
// weight reads as invariant and emit scalar loads (s_load_dwordx8) feeding v_fma SGPR operands.
template <int CIN, int F, int FH, bool PRELOAD, typename LoadX>
__device__ __forceinline__ void cl_gates(const float* __restrict__ Wx, const float* __restrict__ bias, int f0,
                                         LoadX load, float (&gi)[FH], float (&gc)[FH], float (&go)[FH]) {
    constexpr int C4 = (CIN + 3) / 4;
#pragma unroll
    for (int f = 0; f < FH; ++f) {
        gi[f] = bias[f0 + f];
        gc[f] = bias[2 * F + f0 + f];
        go[f] = bias[3 * F + f0 + f];
    }
    if (PRELOAD) {
        // x comes from global memory: rolled tap loop (full unrolling blows the SGPR budget: measured 5x slower)
        // with the next tap's loads issued before the current tap's FMA chain
        f32x4 nxt[C4];
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) nxt[c4] = load(0, 0, c4);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            f32x4 cur[C4];
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) cur[c4] = nxt[c4];
            if (tap < 8) {
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) nxt[c4] = load((tap + 1) / 3, (tap + 1) % 3, c4);
            }
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = 4 * c4 + j;
                    if (c < CIN) {
                        const float* w = Wx + (tap * CIN + c) * 4 * F + f0;
                        if constexpr (FH % 2 == 0) {
#pragma unroll
                            for (int f = 0; f < FH; f += 2) {
                                *(f32x2*)&gi[f] = cl_pk_fma(cur[c4][j], *(const f32x2*)&w[f], *(f32x2*)&gi[f]);
                                *(f32x2*)&gc[f] = cl_pk_fma(cur[c4][j], *(const f32x2*)&w[2 * F + f], *(f32x2*)&gc[f]);
                                *(f32x2*)&go[f] = cl_pk_fma(cur[c4][j], *(const f32x2*)&w[3 * F + f], *(f32x2*)&go[f]);
                            }
                        } else {
#pragma unroll
                            for (int f = 0; f < FH; ++f) {
                                gi[f] = fmaf(cur[c4][j], w[f], gi[f]);
                                gc[f] = fmaf(cur[c4][j], w[2 * F + f], gc[f]);
                                go[f] = fmaf(cur[c4][j], w[3 * F + f], go[f]);
                            }
                        }
                    }
                }
        }
        return;
    }
#pragma unroll 1   // x from LDS: keep the tap loop rolled (full unrolling hoists every weight and spills)
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            const f32x4 xv = load(tap / 3, tap % 3, c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * c4 + j;
                if (c < CIN) {
                    const float* w = Wx + (tap * CIN + c) * 4 * F + f0;   // wave-uniform address -> scalar loads
                    if constexpr (FH % 2 == 0) {
#pragma unroll
                        for (int f = 0; f < FH; f += 2) {
                            *(f32x2*)&gi[f] = cl_pk_fma(xv[j], *(const f32x2*)&w[f], *(f32x2*)&gi[f]);
                            *(f32x2*)&gc[f] = cl_pk_fma(xv[j], *(const f32x2*)&w[2 * F + f], *(f32x2*)&gc[f]);
                            *(f32x2*)&go[f] = cl_pk_fma(xv[j], *(const f32x2*)&w[3 * F + f], *(f32x2*)&go[f]);
                        }
                    } else {
#pragma unroll
                        for (int f = 0; f < FH; ++f) {
                            gi[f] = fmaf(xv[j], w[f], gi[f]);
                            gc[f] = fmaf(xv[j], w[2 * F + f], gc[f]);
                            go[f] = fmaf(xv[j], w[3 * F + f], go[f]);
                        }
                    }
                }
            }
        }
    }
}

// ---- forward: 128 pixels per block, thread = (pixel, feature half) -------------------------------------
template <int CIN, int F>
__global__ void __launch_bounds__(256) wdg_convlstm1_fwd_kernel(const WdgCl1 p, const float* __restrict__ Wx,
                                                                const float* __restrict__ bias) {
    constexpr int FH = F >= 2 ? F / 2 : 1;
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);   // waves 0,1 -> half 0; waves 2,3 -> half 1
    const int f0 = half * FH;
    const long long P = (long long)p.n_img * p.H * p.W;
    const long long pix = (long long)blockIdx.x * 128 + (threadIdx.x & 127);
    if (pix >= P || (F < 2 && half)) return;
    const int img = (int)(pix / ((long long)p.H * p.W));
    const int rem = (int)(pix - (long long)img * p.H * p.W);
    const int oy = rem / p.W, ox = rem - oy * p.W;
    const float* Ximg = p.X + (long long)img * p.imgStrideX;
    auto load = [&](int th, int tw, int c4) -> f32x4 {
        const int gy = oy + th - 1, gx = ox + tw - 1;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
            return *reinterpret_cast<const f32x4*>(Ximg + ((long long)gy * p.W + gx) * p.ldx + 4 * c4);
        return (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    __attribute__((aligned(8))) float gi[FH], gc[FH], go[FH];
    cl_gates<CIN, F, FH, true>(Wx, bias, f0, load, gi, gc, go);
    float* hp = p.Hout + (long long)img * p.imgStrideH + ((long long)oy * p.W + ox) * p.ldh + f0;
#pragma unroll
    for (int f = 0; f < FH; ++f) hp[f] = cl_hsig(go[f]) * cl_tanh(cl_hsig(gi[f]) * cl_tanh(gc[f]));
}

// ---- forward on the matrix pipe (5 -> 16): gates of 16 pixels per MFMA tile, weights resident in REGISTERS --------------------
// GEMM view per fragment of 16 consecutive pixels of a row: gates[n][pixel] = sum_k W[k][n] x[pixel][k], k = the 9 * 5 = 45
// flattened (tap, channel) pairs padded to 48 (12 steps of v_mfma_f32_16x16x4_f32), n = the 3 x 16 live gate columns (i, c~, o:
// with h_0 = c_0 = 0 the forget gate plays no role).  The weights are the MFMA's ROW operand: a lane (li = lane & 15,
// lq = lane >> 4) supplies W[k = 4 ks + lq][gate g][feature li] — 36 values that do not depend on the pixel, loaded once per
// persistent workgroup and kept in registers (the earlier LDS-resident form paid one ds_read per MFMA) — and ends up with gates
// i, c~, o of features 4 lq .. 4 lq + 3 of pixel li, so the cell runs on the accumulators and h leaves as one 16-byte store per
// lane.  x comes from a halo tile in LDS, one plane per channel (consecutive pixels = consecutive banks).  Per 16 pixels: 36
// MFMAs (1,152 matrix-pipe cycles) beside ~150 vector instructions of cell math from other waves — the two pipes overlap
// across the resident waves, which the all-VALU form (2,160 FMAs per pixel on the vector pipe alone) cannot.
constexpr int CLF_TH = 8, CLF_TW = 32;           // 256 pixels = 16 fragments per tile, 4 per wave
// PRE = true (n_timesteps > 1, models.py:101): the same pass leaves the input part of ALL FOUR gates' pre-activations
// (conv(x, kernel) + bias, [pixel][i | f | c~ | o]) for the recurrent steps (convlstm16.hip) instead of running the cell — the
// T-batched producer of a tensor of 64 floats per pixel (453 MB at batch 8, T = 24); the general kernels wrote it at 1.2-1.4 TB/s
// (halo tile 395 us, implicit GEMM 324 us), padding the 45-row reduction to 72 / 80.
// (launch bound: four persistent workgroups per CU are launched — at 132 registers only three were resident)
template <int CIN, int F, bool PRE = false>
__global__ void __launch_bounds__(256, PRE ? 1 : 4) wdg_convlstm1_fwd_mfma_kernel(const WdgCl1 p, const float* __restrict__ Wx,
                                                                     const float* __restrict__ bias) {
    static_assert(F == 16, "one 16-feature MFMA tile per gate");
    constexpr int NG = PRE ? 4 : 3;
    constexpr int KP = (9 * CIN + 3) / 4 * 4, KS = KP / 4;
    constexpr int XH = CLF_TH + 2, XW = CLF_TW + 2;
    constexpr int PLANE = XH * XW + 12;          // 352: consecutive channel planes start 0 mod 32 banks apart -> lq groups offset by tap shifts only
    __shared__ float xs[CIN * PLANE];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, lq = lane >> 4;
    // weights / bias / x offsets of this lane (tile-invariant)
    float wreg[KS][NG];
    int xoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + lq;
        int off = 0;
        if (k < 9 * CIN) {
            const int tap = k / CIN, c = k - tap * CIN;
            const int th = tap / 3, tw = tap - 3 * th;
            off = c * PLANE + th * XW + tw;
#pragma unroll
            for (int g = 0; g < NG; ++g) wreg[ks][g] = Wx[k * 4 * F + (PRE ? g : (g == 0 ? 0 : g + 1)) * F + li];
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) wreg[ks][g] = 0.f;     // padding rows: zero weights, any valid x address
        }
        xoff[ks] = off;
    }
    f32x4 bg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) bg[g][r] = bias[(PRE ? g : (g == 0 ? 0 : g + 1)) * F + 4 * lq + r];
    const int ntiles = p.n_img * p.tiles_h * p.tiles_w;
    // x halo of the NEXT tile in registers while this one is multiplied, requested branch-free (buffer loads, offset 0x80000000 =
    // beyond the descriptor for the padding and for slots beyond the halo): `if (inside) load` in the staging loop was two
    // dependent round trips per tile between two barriers
    // (exactly CIN values per slot — 16 bytes per full channel group, 4 bytes per leftover channel: the allocator recycles unused
    // lanes of a 16-byte destination at once and then has to wait for the load before it may overwrite them)
    constexpr int NXS = (XH * XW + 255) / 256;
    float xr[NXS][CIN];
    auto x_request = [&](int tile_) __attribute__((always_inline)) {
        int b_ = tile_;
        const int tx_ = b_ % p.tiles_w;
        b_ /= p.tiles_w;
        const int ty_ = b_ % p.tiles_h, img_ = b_ / p.tiles_h;
        const wdg_srd srdX = wdg_make_srd(p.X + (long long)img_ * p.imgStrideX);
        const wdg_srd srdX2 = wdg_make_srd(p.X2 ? p.X2 + (long long)img_ * p.imgStrideX2 : p.X);
        (void)srdX2;
#pragma unroll
        for (int s_ = 0; s_ < NXS; ++s_) {
            const int pix = t + 256 * s_;
            const int hy = pix / XW, hx = pix - hy * XW;
            const int gy = ty_ * CLF_TH - 1 + hy, gx = tx_ * CLF_TW - 1 + hx;
            const unsigned neg = (unsigned)((gy | (p.H - 1 - gy) | gx | (p.W - 1 - gx) | (XH * XW - 1 - pix)) >> 31);
            const unsigned off = ((unsigned)((gy * p.W + gx) * p.ldx * 4) & ~neg) | (neg & 0x80000000u);
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; ++c4) {
                const f32x4 q = wdg_buffer_load_f32x4(srdX, off + 16 * c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[s_][4 * c4 + j] = q[j];
            }
#pragma unroll
            for (int c = CIN / 4 * 4; c < CIN; ++c)
                xr[s_][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdX, (int)(off + 4 * c), 0, 0));
            if constexpr (CIN == 5) {
                // (branch-free: without a second source the two requests are out of range and their results unused)
                const unsigned off2 = p.X2 ? (((unsigned)((gy * p.W + gx) * p.ldx2 * 4) & ~neg) | (neg & 0x80000000u)) : 0x80000000u;
#pragma unroll
                for (int c = CIN - 2; c < CIN; ++c) {
                    const int rel = c - (CIN - p.x2_n);
                    const bool in2 = p.X2 != nullptr && rel >= 0;
                    const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdX2, in2 ? (int)(off2 + 4 * rel) : (int)0x80000000u, 0, 0));
                    xr[s_][c] = in2 ? v2 : xr[s_][c];
                }
            }
        }
    };
    if ((int)blockIdx.x < ntiles) x_request(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tx = b % p.tiles_w;
        b /= p.tiles_w;
        const int ty = b % p.tiles_h;
        const int img = b / p.tiles_h;
        const int oy0 = ty * CLF_TH, ox0 = tx * CLF_TW;
        // x halo -> channel planes (zero outside the image = the conv's zero padding; slots beyond the halo land in a plane's
        // 12 padding floats, which nothing reads)
#pragma unroll
        for (int s_ = 0; s_ < NXS; ++s_) {
            const int pix = min(t + 256 * s_, XH * XW);
#pragma unroll
            for (int c = 0; c < CIN; ++c) xs[c * PLANE + pix] = xr[s_][c];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) x_request(tile + gridDim.x);
        // 16 fragments (8 rows x 2 half rows of 16 pixels): wave wv takes rows 2 wv, 2 wv + 1.  Software-pipelined: the 36 MFMAs
        // of fragment fi + 1 stand before the cell arithmetic of fragment fi in program order, so the cell's instructions issue
        // while the matrix instructions drain instead of waiting for their own fragment's last one
        auto gates = [&](int fi, f32x4 (&ga)[NG]) {
            const int pbase = (2 * wv + (fi >> 1)) * XW + (fi & 1) * 16 + li;
#pragma unroll
            for (int g = 0; g < NG; ++g) ga[g] = bg[g];       // the accumulators start at the bias
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float xv = xs[pbase + xoff[ks]];
#pragma unroll
                for (int g = 0; g < NG; ++g) ga[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ks][g], xv, ga[g], 0, 0, 0);
            }
        };
        auto cell = [&](int fi, const f32x4 (&ga)[NG]) {
            const int gy = oy0 + 2 * wv + (fi >> 1), gx = ox0 + (fi & 1) * 16 + li;
            if constexpr (PRE) {
                if (gy < p.H && gx < p.W) {
                    float* dst = p.Hout + (long long)img * p.imgStrideH + ((long long)gy * p.W + gx) * p.ldh + 4 * lq;
#pragma unroll
                    for (int g = 0; g < NG; ++g) *reinterpret_cast<f32x4*>(dst + g * F) = ga[g];
                }
                return;
            }
            f32x4 h4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h4[r] = cl_hsig(ga[2][r]) * cl_tanh(cl_hsig(ga[0][r]) * cl_tanh(ga[1][r]));
            if (gy < p.H && gx < p.W)
                *reinterpret_cast<f32x4*>(p.Hout + (long long)img * p.imgStrideH + ((long long)gy * p.W + gx) * p.ldh + 4 * lq) = h4;
        };
        f32x4 g0[NG], g1[NG];
        gates(0, g0);
        gates(1, g1);
        cell(0, g0);
        gates(2, g0);
        cell(1, g1);
        gates(3, g1);
        cell(2, g0);
        cell(3, g1);
        __syncthreads();   // the next tile overwrites xs
    }
}

// ---- backward: 4x32 centre tile; dgates recomputed on the 6x34 halo into LDS, dx gathered from it --------
// WG = true additionally forms the layer's weight and bias gradient in the same pass: with x and dgates of the tile
// already in LDS, dW[(tap,ci)][gate] += sum_pixels x[p + tap - 1][ci] * dgates[p][gate] is 9 (2 for the 2-channel
// layer) 16x16 MFMA tiles over the tile's 128 pixels (rows = 9*CIN taps-by-channels + one constant-1 row for the
// bias gradient; columns = the 3F live gates; wave w takes the pixels of centre row w).  Blocks are persistent
// (grid-stride over tiles, accumulators stay in registers); each block leaves one partial that a second kernel sums
// in block order.  This replaces writing the dense 4F-channel dgates tensor (537 MB at the headline shape) and
// reading it back in a separate weight-gradient kernel.
// MF = true (F = 16 only): the gate recompute of stage 2 runs on the MATRIX pipe — rows = halo pixels (16 per tile), columns
// = the 3F = 48 live gate columns, k = the 9 * CIN = 45 (padded 48) flattened (tap, channel) pairs, weights resident in REGISTERS
// for the life of the persistent block (36 per lane: the MFMA's row operand does not depend on the pixel) — and the cell backward runs on the transposed accumulators (a lane holds gates
// i, c~, o of four features of one pixel).  The vector pipe keeps the cell math and the input-gradient stage; with three
// workgroups per CU in different stages the two pipes overlap (MI355X_MICROARCH.md: MFMA and VALU issue are separate).
// DXC0 > 0: the input gradient of channels [DXC0, CIN) only, written to dX[..., 0 : CIN - DXC0] (the discriminator needs the
// gradient of the two high-resolution channels of concat(low, high), models.py:100: the three low-resolution ones are data)
template <int CIN, int F, bool WG, bool MF = false, int DXC0 = 0>
__global__ void __launch_bounds__(256) wdg_convlstm1_bwd_kernel(const WdgCl1 p, const float* __restrict__ Wx,
                                                                const float* __restrict__ bias) {
    static_assert(!MF || F == 16, "the MFMA gate stage is written for 16 features");
    static_assert(DXC0 >= 0 && DXC0 < CIN, "channel range");
    constexpr int FH = F >= 2 ? F / 2 : 1;
    constexpr int C4 = (CIN + 3) / 4;
    constexpr int XH = CL_TH + 4, XW = CL_TW + 4;    // x halo (two 3x3 stages)
    constexpr int GH = CL_TH + 2, GW = CL_TW + 2;    // dgates halo
    constexpr int G3 = 3 * F + 1;                    // compact dgates [i | c~ | o] + 1 pad: an odd pixel stride keeps the per-pixel b32 accesses of a wave on distinct banks (3F = 48 was a 16-way conflict)
    constexpr int ROWS = 9 * CIN + 1, RT = (ROWS + 15) / 16, COLS = 3 * F, CT = (COLS + 15) / 16;
    constexpr int WN = RT * CT * 256;                // floats of one wave's accumulator tiles
    constexpr bool RED_IN_DGS = GH * GW * G3 >= 4 * WN;
    __shared__ __attribute__((aligned(16))) f32x4 xs[XH * XW * C4 + 1];     // (+ 1: where staging slots beyond the halo are written)
    __shared__ __attribute__((aligned(16))) float dgs[GH * GW * G3];
    __shared__ float dxp[128 * CIN];
    __shared__ float red_extra[(WG && !RED_IN_DGS) ? 4 * WN : 1];
    constexpr int KP = (9 * CIN + 3) / 4 * 4;        // flattened (tap, channel) rows padded to the MFMA k granule
    constexpr int NXS = (XH * XW * C4 + 255) / 256;  // x-halo staging slots per thread

    const int t = threadIdx.x;
    const int half = __builtin_amdgcn_readfirstlane(t >> 7);
    const int f0 = half * FH;
    const bool half_on = !(F < 2 && half);

    // fused weight gradient: tile-invariant operand addressing of this lane
    const int lane = t & 63, wv = t >> 6, li = lane & 15, lq = lane >> 4;
    f32x4 wacc[WG ? RT * CT : 1];
    int a_base[RT], a_mode[RT];
#pragma unroll
    for (int i = 0; i < (WG ? RT * CT : 1); ++i) wacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = rt * 16 + li;
        if (row < 9 * CIN) {
            const int tap = row / CIN, ci = row - tap * CIN;
            const int th = tap / 3, tw = tap - 3 * th;
            a_base[rt] = (((ci >> 2) * (XH * XW) + (1 + th) * XW + 1 + tw) << 2) + (ci & 3);
            a_mode[rt] = 0;
        } else {
            a_base[rt] = 0;
            a_mode[rt] = row == 9 * CIN ? 1 : 2;     // the constant-1 bias row / padding rows
        }
    }

    int xoff[MF ? KP / 4 : 1];
    float wreg[MF ? KP / 4 : 1][3];                        // MF: this lane's weights W[k = 4 ks + lq][gate][feature li], in registers
    if constexpr (MF) {
#pragma unroll
        for (int ks = 0; ks < KP / 4; ++ks) {
            const int k = 4 * ks + lq;
            int off = 0;                                   // padding rows: any valid address (their weights are zero)
#pragma unroll
            for (int g = 0; g < 3; ++g) wreg[ks][g] = 0.f;
            if (k < 9 * CIN) {
                const int tap = k / CIN, c = k - tap * CIN;
                const int th = tap / 3, tw = tap - 3 * th;
                off = (((c >> 2) * (XH * XW) + th * XW + tw) << 2) + (c & 3);
#pragma unroll
                for (int g = 0; g < 3; ++g) wreg[ks][g] = Wx[k * 4 * F + (g == 0 ? 0 : g + 1) * F + li];
            }
            xoff[ks] = off;
        }
    }
    const int ntiles = p.n_img * p.tiles_h * p.tiles_w;
    // The tile loop is a chain of round trips (x halo, dh, the read-modify-write of dx) with three workgroups per CU to hide them:
    // every request is branch-free (buffer loads, offset 0x80000000 = beyond the descriptor for padding — `if (inside) v = load`
    // compiles to an exec-masked block with its own full wait per load), the x halo of the NEXT tile travels during this tile's
    // arithmetic (so does its dh), the previous dx values are requested together.
    f32x4 xr[NXS];
    constexpr bool DHPRE = !MF && FH % 4 == 0;       // dh of this thread's work items of stage 2 (the scalar-gate path) likewise
    f32x4 dhr[DHPRE ? 2 : 1][DHPRE ? FH / 4 : 1];
    auto x_request = [&](int tile_) __attribute__((always_inline)) {
        int b_ = tile_;
        const int tx_ = b_ % p.tiles_w;
        b_ /= p.tiles_w;
        const int ty_ = b_ % p.tiles_h, img_ = b_ / p.tiles_h;
        const wdg_srd srdX = wdg_make_srd(p.X + (long long)img_ * p.imgStrideX);
        const wdg_srd srdX2 = wdg_make_srd(p.X2 ? p.X2 + (long long)img_ * p.imgStrideX2 : p.X);
        (void)srdX2;
        if constexpr (DHPRE) {
            const wdg_srd srdDH = wdg_make_srd(p.dH + (long long)img_ * p.imgStrideDH);
            const int n_items_ = p.dX ? GH * GW : CL_TH * CL_TW;
#pragma unroll
            for (int r_ = 0; r_ < 2; ++r_) {
                const int item = r_ * 128 + (t & 127);
                const int hy_ = p.dX ? item / GW : 1 + item / CL_TW;
                const int hx_ = p.dX ? item - (item / GW) * GW : 1 + item % CL_TW;
                const int gy = ty_ * CL_TH - 1 + hy_, gx = tx_ * CL_TW - 1 + hx_;
                const unsigned neg = (unsigned)((gy | (p.H - 1 - gy) | gx | (p.W - 1 - gx) | (n_items_ - 1 - item)) >> 31);
                const unsigned off = ((unsigned)(((gy * p.W + gx) * p.lddh + f0) * 4) & ~neg) | (neg & 0x80000000u);
#pragma unroll
                for (int q = 0; q < FH / 4; ++q) dhr[r_][q] = wdg_buffer_load_f32x4(srdDH, off + 16 * q);
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < NXS; ++s_) {
            const int idx = t + 256 * s_;
            const int c4 = idx % C4, pix = idx / C4;
            const int hy = pix / XW, hx = pix - hy * XW;
            const int gy = ty_ * CL_TH - 2 + hy, gx = tx_ * CL_TW - 2 + hx;
            const unsigned neg = (unsigned)((gy | (p.H - 1 - gy) | gx | (p.W - 1 - gx) | (XH * XW * C4 - 1 - idx)) >> 31);
            xr[s_] = wdg_buffer_load_f32x4(srdX, ((unsigned)(((gy * p.W + gx) * p.ldx + 4 * c4) * 4) & ~neg) | (neg & 0x80000000u));
            if constexpr (CIN == 5) {
                // second source (WdgCl1::X2): of this slot's four channels only channel 3 (group 0) / channel 4 (group 1) can come
                // from it — one more 4-byte request per slot, out of range (unused) without a second source
                const int csel = c4 == 0 ? 3 : 4;
                const int rel = csel - (CIN - p.x2_n);
                const bool in2 = p.X2 != nullptr && rel >= 0;
                const unsigned o2 = (((unsigned)(((gy * p.W + gx) * p.ldx2 + rel) * 4) & ~neg) | (neg & 0x80000000u));
                const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srdX2, in2 ? (int)o2 : (int)0x80000000u, 0, 0));
                xr[s_][3] = (in2 && c4 == 0) ? v2 : xr[s_][3];
                xr[s_][0] = (in2 && c4 == 1) ? v2 : xr[s_][0];
            }
        }
    };
    if ((int)blockIdx.x < ntiles) x_request(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % p.tiles_w;
    b /= p.tiles_w;
    const int ty = b % p.tiles_h;
    const int img = b / p.tiles_h;
    const int oy0 = ty * CL_TH, ox0 = tx * CL_TW;
    const float* DHimg = p.dH + (long long)img * p.imgStrideDH;

    // 1. x halo -> LDS (zero outside the image = the conv's zero padding)
#pragma unroll
    for (int s_ = 0; s_ < NXS; ++s_) {
        const int idx = t + 256 * s_;
        const int c4 = idx % C4, pix = idx / C4;
        // channel-group planes: a wave's 16-byte reads of consecutive pixels are contiguous
        xs[idx < XH * XW * C4 ? c4 * (XH * XW) + pix : XH * XW * C4] = xr[s_];
    }
    __syncthreads();
    if constexpr (MF) {
        // 2'. gates of the (GH x GW) halo pixels on the matrix pipe, 16 pixels per tile, wave w takes tiles w, w + 4, ...
        const float* xsf2 = reinterpret_cast<const float*>(xs);
        constexpr int NMT = (GH * GW + 15) / 16;
        for (int mt = wv; mt < NMT; mt += 4) {
            const int hp_ = mt * 16 + li;
            const bool hp_ok = hp_ < GH * GW;
            const int hy = hp_ok ? hp_ / GW : 0, hx = hp_ok ? hp_ - hy * GW : 0;
            const int pix0 = (hy * XW + hx) << 2;
            f32x4 ga[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) ga[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KP / 4; ++ks) {
                const float xv = xsf2[pix0 + xoff[ks]];
#pragma unroll
                for (int g = 0; g < 3; ++g) ga[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ks][g], xv, ga[g], 0, 0, 0);
            }
            // accumulator reg r of lane (li, lq): gate g of feature 4*lq + r of halo pixel li
            const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            const bool inside = hp_ok && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            f32x4 dh4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (inside) dh4 = *reinterpret_cast<const f32x4*>(DHimg + ((long long)gy * p.W + gx) * p.lddh + 4 * lq);
            if (hp_ok) {
                float* d = &dgs[hp_ * G3 + 4 * lq];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 4 * lq + r;
                    const float xi = ga[0][r] + bias[f], xc = ga[1][r] + bias[2 * F + f], xo = ga[2][r] + bias[3 * F + f];
                    const float si = cl_hsig(xi), tc_ = cl_tanh(xc), so = cl_hsig(xo);
                    const float th = cl_tanh(si * tc_);
                    const float dh = dh4[r];
                    const float dc = dh * so * (1.f - th * th);
                    d[r] = inside ? dc * tc_ * cl_hsig_grad(xi) : 0.f;
                    d[F + r] = inside ? dc * si * (1.f - tc_ * tc_) : 0.f;
                    d[2 * F + r] = inside ? dh * th * cl_hsig_grad(xo) : 0.f;
                }
            }
        }
    } else {
    // 2. dgates on the (GH x GW) halo: work item = (halo pixel, half); halves are wave-uniform.  Without an input gradient (the
    // weights-only pass of the critic update: 6 of the 10 launches of a train step) nothing reads the halo ring — the weight
    // gradient contracts over the centre pixels only — so the gates are recomputed for the 128 centre pixels: one round of
    // work items instead of two.
    const int n_items = p.dX ? GH * GW : CL_TH * CL_TW;
    for (int base = 0; base < n_items; base += 128) {
        const int item = base + (t & 127);
        const int hy_ = p.dX ? item / GW : 1 + item / CL_TW;
        const int hx_ = p.dX ? item - (item / GW) * GW : 1 + item % CL_TW;
        const int hp_ = hy_ * GW + hx_;
        if (item < n_items && half_on) {
            const int hy = hy_, hx = hx_;
            const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
            const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            float dgi[FH], dgc[FH], dgo[FH];
#pragma unroll
            for (int f = 0; f < FH; ++f) dgi[f] = dgc[f] = dgo[f] = 0.f;
            if (inside) {
                auto load = [&](int th, int tw, int c4) -> f32x4 { return xs[c4 * (XH * XW) + (hy + th) * XW + hx + tw]; };
                __attribute__((aligned(8))) float gi[FH], gc[FH], go[FH];
                cl_gates<CIN, F, FH, false>(Wx, bias, f0, load, gi, gc, go);
                const float* dhp = DHimg + ((long long)gy * p.W + gx) * p.lddh + f0;
                (void)dhp;
#pragma unroll
                for (int f = 0; f < FH; ++f) {
                    const float si = cl_hsig(gi[f]), tc_ = cl_tanh(gc[f]), so = cl_hsig(go[f]);
                    const float c = si * tc_;
                    const float th = cl_tanh(c);
                    float dh;
                    if constexpr (DHPRE) dh = dhr[base ? 1 : 0][f / 4][f % 4];
                    else dh = dhp[f];
                    const float dc = dh * so * (1.f - th * th);
                    dgi[f] = dc * tc_ * cl_hsig_grad(gi[f]);
                    dgc[f] = dc * si * (1.f - tc_ * tc_);
                    dgo[f] = dh * th * cl_hsig_grad(go[f]);
                }
                // dense dgates for the weight-gradient kernel: centre pixels only, gate order i,f,c,o (f = 0)
                if (p.dG && hy >= 1 && hy <= CL_TH && hx >= 1 && hx <= CL_TW) {
                    float* dg = p.dG + (((long long)img * p.H + gy) * p.W + gx) * 4 * F + f0;
#pragma unroll
                    for (int f = 0; f < FH; ++f) {
                        dg[f] = dgi[f];
                        dg[F + f] = 0.f;
                        dg[2 * F + f] = dgc[f];
                        dg[3 * F + f] = dgo[f];
                    }
                }
            }
            float* d = &dgs[hp_ * G3 + f0];
#pragma unroll
            for (int f = 0; f < FH; ++f) {
                d[f] = dgi[f];
                d[F + f] = dgc[f];
                d[2 * F + f] = dgo[f];
            }
        }
    }
    }
    __syncthreads();
    // (x halo and dh of the next tile: under the weight-gradient MFMAs and the dx stage — requested before stage 2 they would be
    // waited for at its first branch, the wait counter being in order)
    if (tile + (int)gridDim.x < ntiles) x_request(tile + gridDim.x);
    if constexpr (WG) {
        // 2b. weight / bias gradient of this tile: wave wv = centre row wv, 8 steps of 4 pixels
        const float* xsf = reinterpret_cast<const float*>(xs);
#pragma unroll 2
        for (int sstep = 0; sstep < 8; ++sstep) {
            const int px_ = 4 * sstep + lq;
            const int xo = (wv * XW + px_) << 2;
            const int go = ((wv + 1) * GW + px_ + 1) * G3;
            float av[RT], bv[CT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                av[rt] = a_mode[rt] == 0 ? xsf[a_base[rt] + xo] : (a_mode[rt] == 1 ? 1.f : 0.f);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bv[ct] = ct * 16 + li < COLS ? dgs[go + ct * 16 + li] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    wacc[rt * CT + ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[ct], wacc[rt * CT + ct], 0, 0, 0);
        }
    }
    if (p.dX) {
    // 3. dx[c] = sum_tap sum_g dgates[pixel + (1 - th, 1 - tw)][g] * Wx[tap][c][g]; each half sums its own gates
    const int cp = t & 127;
    const int py = cp >> 5, px = cp & 31;
    float dx[CIN];
#pragma unroll
    for (int c = DXC0; c < CIN; ++c) dx[c] = 0.f;
    if (half_on) {
        f32x2 dx2[CIN];   // packed partial sums over even / odd features
#pragma unroll
        for (int c = DXC0; c < CIN; ++c) dx2[c] = (f32x2){0.f, 0.f};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int th = tap / 3, tw = tap % 3;
            const float* dgp = &dgs[((py + 2 - th) * GW + px + 2 - tw) * G3 + f0];
            __attribute__((aligned(8))) float v[3][FH];
#pragma unroll
            for (int f = 0; f < FH; ++f) {
                v[0][f] = dgp[f];
                v[1][f] = dgp[F + f];
                v[2][f] = dgp[2 * F + f];
            }
#pragma unroll
            for (int c = DXC0; c < CIN; ++c) {
                const float* w = Wx + (tap * CIN + c) * 4 * F + f0;
                if constexpr (FH % 2 == 0) {
#pragma unroll
                    for (int f = 0; f < FH; f += 2) {
                        dx2[c] = __builtin_elementwise_fma(*(const f32x2*)&v[0][f], *(const f32x2*)&w[f], dx2[c]);
                        dx2[c] = __builtin_elementwise_fma(*(const f32x2*)&v[1][f], *(const f32x2*)&w[2 * F + f], dx2[c]);
                        dx2[c] = __builtin_elementwise_fma(*(const f32x2*)&v[2][f], *(const f32x2*)&w[3 * F + f], dx2[c]);
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < FH; ++f) {
                        dx[c] = fmaf(v[0][f], w[f], dx[c]);
                        dx[c] = fmaf(v[1][f], w[2 * F + f], dx[c]);
                        dx[c] = fmaf(v[2][f], w[3 * F + f], dx[c]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = DXC0; c < CIN; ++c) dx[c] += dx2[c][0] + dx2[c][1];
    }
    if (half == 1) {
#pragma unroll
        for (int c = DXC0; c < CIN; ++c) dxp[cp * CIN + c] = dx[c];
    }
    __syncthreads();
    if (half == 0) {
        const int gy = oy0 + py, gx = ox0 + px;
        float prev[CIN];
#pragma unroll
        for (int c = DXC0; c < CIN; ++c) prev[c] = 0.f;
        if (p.accumulate_dx) {          // (the previous values together, from a clamped = always valid address)
            const float* src = p.dX + (long long)img * p.imgStrideDX + ((long long)min(gy, p.H - 1) * p.W + min(gx, p.W - 1)) * p.lddx;
#pragma unroll
            for (int c = DXC0; c < CIN; ++c) prev[c] = src[c - DXC0];
        }
        if (gy < p.H && gx < p.W) {
            float* dst = p.dX + (long long)img * p.imgStrideDX + ((long long)gy * p.W + gx) * p.lddx;
#pragma unroll
            for (int c = DXC0; c < CIN; ++c) dst[c - DXC0] = dx[c] + (F >= 2 ? dxp[cp * CIN + c] : 0.f) + prev[c];
        }
    }
    }   // if (p.dX)
    if (gridDim.x < (unsigned)ntiles) __syncthreads();   // persistent blocks: the next tile overwrites xs / dgs / dxp
    }   // tile loop
    if constexpr (WG) {
        // block partial: accumulator reg r of lane (li, lq) is C[row 4*lq + r][col li]; sum the four waves in wave order
        float* red = RED_IN_DGS ? dgs : red_extra;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RT * CT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wv * WN + i * 256 + (4 * lq + r) * 16 + li] = wacc[i][r];
        __syncthreads();
        float* dst = p.dWpart + (long long)blockIdx.x * (RT * 16) * (CT * 16);
        for (int idx = t; idx < WN; idx += 256) {
            const float v = (red[idx] + red[WN + idx]) + (red[2 * WN + idx] + red[3 * WN + idx]);
            const int i = idx >> 8, row = (idx >> 4) & 15, col = idx & 15;
            dst[((i / CT) * 16 + row) * (CT * 16) + (i % CT) * 16 + col] = v;
        }
    }
}

// second stage of the fused weight gradient: block partials summed in block order -> dW [3][3][CIN][4F] and dbias [4F]
// (+=; the forget-gate columns receive nothing: with h_0 = c_0 = 0 that gate has no influence at T = 1)
template <int CIN, int F>
__global__ void __launch_bounds__(256) wdg_convlstm1_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks,
                                                                         float* __restrict__ dW, float* __restrict__ dbias) {
    constexpr int ROWS = 9 * CIN + 1, RT = (ROWS + 15) / 16, COLS = 3 * F, CT = (COLS + 15) / 16;
    // one wave per output element: lane l sums the partials of blocks l, l + 64, ... (fixed order), then a wave sum
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= ROWS * COLS) return;
    const int row = idx / COLS, col = idx - row * COLS;
    const float* src = part + row * (CT * 16) + col;
    float v = 0.f;
    for (int b = lane; b < nblocks; b += 64) v += src[(long long)b * (RT * 16) * (CT * 16)];
    v = wdg_wave_sum(v);
    if (lane) return;
    const int gate = col / F, f = col - gate * F;
    const int gcol = (gate == 0 ? 0 : gate + 1) * F + f;          // compact [i | c~ | o] -> dense [i | f | c~ | o]
    if (row < 9 * CIN)
        dW[row * 4 * F + gcol] += v;
    else
        dbias[gcol] += v;
}

// ---- host -------------------------------------------------------------------------------------------------
static int g_cl1_mfma = 0;     // wdg_set_tuning("convlstm1_mfma", bit 0): gate recompute of the 5 -> 16 backward on the matrix pipe
static int g_cl1_fwd_mfma = 1; // bit 1 of the same knob CLEARS it: 5 -> 16 forward on the matrix pipe with register-resident weights
void wdg_convlstm1_set_mfma(int v) { g_cl1_mfma = (v & 1) != 0; g_cl1_fwd_mfma = (v & 2) == 0; }
static int cl1_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        return prop.multiProcessorCount;
    return 256;
}
extern "C" int wdg_convlstm1_supported(int cin, int F) { return (cin == 2 && F == 2) || (cin == 5 && F == 16); }

// second source of the last x2_n input channels (WdgCl1::X2): the 5-channel layer on its matrix-pipe forward / its fused backward
extern "C" int wdg_convlstm1_x2_supported(int cin, int F, int x2_n) { return cin == 5 && F == 16 && x2_n >= 1 && x2_n <= 2 && g_cl1_fwd_mfma; }

extern "C" int wdg_convlstm1_fwd(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                                 float* h, int ldh, int64_t img_stride_h, int n_img, int H, int W, int cin, int F,
                                 wdg_stream stream) {
    return wdg_convlstm1_fwd_x2(x, ldx, img_stride_x, nullptr, 0, 0, 0, wx, bias, h, ldh, img_stride_h, n_img, H, W, cin, F, stream);
}

extern "C" int wdg_convlstm1_fwd_x2(const float* x, int ldx, int64_t img_stride_x, const float* x2, int ldx2, int64_t img_stride_x2,
                                    int x2_n, const float* wx, const float* bias, float* h, int ldh, int64_t img_stride_h, int n_img,
                                    int H, int W, int cin, int F, wdg_stream stream) {
    WDG_CHECK_ARG(x && wx && bias && h, "null argument");
    WDG_CHECK_ARG(wdg_convlstm1_supported(cin, F), "unsupported (cin, F)");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ldx % 4 == 0 && ldx >= wdg_round_up(cin, 4), "x alignment / ld");
    WDG_CHECK_ARG(!x2 || (wdg_convlstm1_x2_supported(cin, F, x2_n) && ldx2 >= x2_n && ldh % 4 == 0 && ((uintptr_t)h & 15) == 0 && img_stride_h % 4 == 0),
                  "second input source: 5 -> 16 layer on the matrix-pipe forward only");
    WdgCl1 p;
    memset(&p, 0, sizeof(p));
    p.X2 = x2; p.ldx2 = ldx2; p.imgStrideX2 = img_stride_x2; p.x2_n = x2 ? x2_n : 0;
    p.X = x; p.Wx = wx; p.bias = bias; p.Hout = h;
    p.imgStrideX = img_stride_x; p.imgStrideH = img_stride_h;
    p.n_img = n_img; p.H = H; p.W = W; p.ldx = ldx; p.ldh = ldh;
    const long long P = (long long)n_img * H * W;
    dim3 grid((unsigned)((P + 127) / 128)), block(256);
    if (cin == 2)
        hipLaunchKernelGGL((wdg_convlstm1_fwd_kernel<2, 2>), grid, block, 0, (hipStream_t)stream, p, wx, bias);
    else if (g_cl1_fwd_mfma && ldh % 4 == 0 && ((uintptr_t)h & 15) == 0 && img_stride_h % 4 == 0) {
        p.tiles_h = (H + CLF_TH - 1) / CLF_TH;
        p.tiles_w = (W + CLF_TW - 1) / CLF_TW;
        const long long ntiles = (long long)n_img * p.tiles_h * p.tiles_w;
        dim3 pgrid((unsigned)std::min<long long>(ntiles, (long long)cl1_cus() * 4));     // persistent: weights loaded once per workgroup
        hipLaunchKernelGGL((wdg_convlstm1_fwd_mfma_kernel<5, 16>), pgrid, block, 0, (hipStream_t)stream, p, wx, bias);
    } else
        hipLaunchKernelGGL((wdg_convlstm1_fwd_kernel<5, 16>), grid, block, 0, (hipStream_t)stream, p, wx, bias);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// Input part of the gate pre-activations of the 5 -> 16-feature ConvLSTM2D for all timesteps (n_timesteps > 1, models.py:101):
// gates[n, y, x, i | f | c~ | o] = conv(x, kernel)[...] + bias.  wx: the kernel [3][3][5][64] (HWIO) as stored.
// ---- the 2 -> 2-feature layer at n_timesteps > 1 (models.py:93): input part of the gates for ALL timesteps and its data gradient on
// the vector unit, one pixel per thread.  8 output columns from 2 channels x 9 taps (144 FMAs, weights wave-uniform = scalar
// operands): 48 bytes of HBM traffic per pixel.  The general halo-tile kernel (16-column MFMA tiles, K padded to 36) took 139 us for
// the 1.77 M pixels of batch 8 x T 24 x 96^2 — ten times the 85 MB it moves.
__global__ void __launch_bounds__(256) wdg_convlstm2_gates_x_kernel(const float* __restrict__ X, int ldx, long long imgStrideX,
                                                                    const float* __restrict__ Wx, const float* __restrict__ bias,
                                                                    float* __restrict__ gates, int n_img, int H, int W) {
    const long long P = (long long)n_img * H * W;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= P) return;
    const int img = (int)(pix / ((long long)H * W));
    const int rem = (int)(pix - (long long)img * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    const float* Ximg = X + (long long)img * imgStrideX;
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = bias[n];
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int gy = oy + tap / 3 - 1, gx = ox + tap % 3 - 1;
        float x0 = 0.f, x1 = 0.f;
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            const float* q = Ximg + ((long long)gy * W + gx) * ldx;
            x0 = q[0];
            x1 = q[1];
        }
        const float* w = Wx + tap * 16;                 // [tap][c][8]
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = fmaf(x1, w[8 + n], fmaf(x0, w[n], acc[n]));
    }
    f32x4* g = reinterpret_cast<f32x4*>(gates + pix * 8);
    g[0] = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    g[1] = (f32x4){acc[4], acc[5], acc[6], acc[7]};
}

// dx[p][c] (+)= sum_tap sum_n dgates[p + (1 - th, 1 - tw)][n] * Wx[tap][c][n]   (the transposed convolution of the above)
__global__ void __launch_bounds__(256) wdg_convlstm2_dx_kernel(const float* __restrict__ dG, const float* __restrict__ Wx,
                                                               float* __restrict__ dX, int lddx, long long imgStrideDX, int accumulate,
                                                               int n_img, int H, int W) {
    const long long P = (long long)n_img * H * W;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= P) return;
    const int img = (int)(pix / ((long long)H * W));
    const int rem = (int)(pix - (long long)img * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    const float* Gimg = dG + (long long)img * H * W * 8;
    float d0 = 0.f, d1 = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int gy = oy + 1 - tap / 3, gx = ox + 1 - tap % 3;
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            const f32x4* q = reinterpret_cast<const f32x4*>(Gimg + ((long long)gy * W + gx) * 8);
            const f32x4 a = q[0], b = q[1];
            const float* w = Wx + tap * 16;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                d0 = fmaf(a[n], w[n], d0);
                d0 = fmaf(b[n], w[4 + n], d0);
                d1 = fmaf(a[n], w[8 + n], d1);
                d1 = fmaf(b[n], w[12 + n], d1);
            }
        }
    }
    float* dst = dX + (long long)img * imgStrideDX + (long long)rem * lddx;
    if (accumulate) {
        d0 += dst[0];
        d1 += dst[1];
    }
    dst[0] = d0;
    dst[1] = d1;
}

static int g_cl2_thin = 1;      // wdg_set_tuning("lstm2_thin", 0/1): the one-pixel-per-thread kernels of the 2 -> 2-feature layer
void wdg_cl2_set_thin(int v) { g_cl2_thin = v != 0; }
extern "C" int wdg_convlstm_gates_x_supported(int cin, int F) { return (g_cl1_fwd_mfma && cin == 5 && F == 16) || (g_cl2_thin && cin == 2 && F == 2); }
extern "C" int wdg_convlstm_gates_x(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias, float* gates,
                                    int n_img, int H, int W, int cin, int F, wdg_stream stream) {
    WDG_CHECK_ARG(x && wx && bias && gates && wdg_convlstm_gates_x_supported(cin, F), "unsupported");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)gates & 15) == 0 && ldx % 4 == 0 && ldx >= wdg_round_up(cin, 4), "x / gates alignment, ld");
    if (cin == 2) {
        const long long P = (long long)n_img * H * W;
        hipLaunchKernelGGL(wdg_convlstm2_gates_x_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           (long long)img_stride_x, wx, bias, gates, n_img, H, W);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
    WdgCl1 p;
    memset(&p, 0, sizeof(p));
    p.X = x; p.Wx = wx; p.bias = bias; p.Hout = gates;
    p.imgStrideX = img_stride_x; p.imgStrideH = (long long)H * W * 4 * F;
    p.n_img = n_img; p.H = H; p.W = W; p.ldx = ldx; p.ldh = 4 * F;
    p.tiles_h = (H + CLF_TH - 1) / CLF_TH;
    p.tiles_w = (W + CLF_TW - 1) / CLF_TW;
    const long long ntiles = (long long)n_img * p.tiles_h * p.tiles_w;
    dim3 pgrid((unsigned)std::min<long long>(ntiles, (long long)cl1_cus() * 4)), block(256);
    hipLaunchKernelGGL((wdg_convlstm1_fwd_mfma_kernel<5, 16, true>), pgrid, block, 0, (hipStream_t)stream, p, wx, bias);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// Data gradient of wdg_convlstm_gates_x for the 2 -> 2-feature layer: dx[..., 0:2] (+)= conv_transpose(dgates, wx).
// dgates: dense [n][H][W][8]; wx: the kernel [3][3][2][8] (HWIO) as stored.
extern "C" int wdg_convlstm_gates_dx_supported(int cin, int F) { return g_cl2_thin && cin == 2 && F == 2; }
extern "C" int wdg_convlstm_gates_dx(const float* dgates, const float* wx, float* dx, int lddx, int64_t img_stride_dx, int accumulate,
                                     int n_img, int H, int W, int cin, int F, wdg_stream stream) {
    WDG_CHECK_ARG(dgates && wx && dx && wdg_convlstm_gates_dx_supported(cin, F), "unsupported");
    WDG_CHECK_ARG(((uintptr_t)dgates & 15) == 0 && lddx >= 2, "dgates alignment / dx stride");
    const long long P = (long long)n_img * H * W;
    hipLaunchKernelGGL(wdg_convlstm2_dx_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dgates, wx, dx, lddx,
                       (long long)img_stride_dx, accumulate, n_img, H, W);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

struct Cl1X2 {
    const float* x2;
    int ldx2;
    int64_t img_stride_x2;
    int x2_n;
};

static int cl1_bwd(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                   const float* dh, int lddh, int64_t img_stride_dh, float* dgates, float* dx, int lddx,
                   int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                   float* dw, float* dbias, void* ws, size_t ws_bytes, wdg_stream stream, const Cl1X2* x2 = nullptr, int dx_c0 = 0) {
    WDG_CHECK_ARG(x && wx && bias && dh, "null argument");
    WDG_CHECK_ARG(dx_c0 == 0 || (dx_c0 == 3 && cin == 5 && dx && !dw && !dgates), "dx_c0: 0, or 3 on the 5 -> 16 layer's input-gradient-only call");
    WDG_CHECK_ARG(wdg_convlstm1_supported(cin, F), "unsupported (cin, F)");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ldx % 4 == 0 && ldx >= wdg_round_up(cin, 4), "x alignment / ld");
    WDG_CHECK_ARG(!(x2 && x2->x2) || (wdg_convlstm1_x2_supported(cin, F, x2->x2_n) && x2->ldx2 >= x2->x2_n), "second input source: 5 -> 16 layer only");
    WdgCl1 p;
    memset(&p, 0, sizeof(p));
    if (x2 && x2->x2) { p.X2 = x2->x2; p.ldx2 = x2->ldx2; p.imgStrideX2 = x2->img_stride_x2; p.x2_n = x2->x2_n; }
    p.X = x; p.Wx = wx; p.bias = bias; p.dH = dh; p.dG = dgates; p.dX = dx;
    p.imgStrideX = img_stride_x; p.imgStrideDH = img_stride_dh; p.imgStrideDX = img_stride_dx;
    p.n_img = n_img; p.H = H; p.W = W; p.ldx = ldx; p.lddh = lddh; p.lddx = lddx;
    p.accumulate_dx = accumulate_dx;
    p.tiles_h = (H + CL_TH - 1) / CL_TH;
    p.tiles_w = (W + CL_TW - 1) / CL_TW;
    const long long ntiles = (long long)n_img * p.tiles_h * p.tiles_w;
    dim3 block(256);
    hipStream_t st = (hipStream_t)stream;
    if (!dw) {
        dim3 grid((unsigned)ntiles);
        if (cin == 2)
            hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<2, 2, false>), grid, block, 0, st, p, wx, bias);
        else if (dgates || !g_cl1_mfma) {
            if (dx_c0 == 3)
                hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<5, 16, false, false, 3>), grid, block, 0, st, p, wx, bias);
            else
                hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<5, 16, false>), grid, block, 0, st, p, wx, bias);
        } else {
            // persistent like the weight-gradient form: the LDS-resident weights are loaded once per workgroup
            dim3 pgrid((unsigned)std::min<long long>(ntiles, (long long)cl1_cus() * 2));   // (three fit — 52 KB of LDS, <= 168 registers — and measured slower: 594 vs 480 us with the weight gradient)
            if (dx_c0 == 3)
                hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<5, 16, false, true, 3>), pgrid, block, 0, st, p, wx, bias);
            else
                hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<5, 16, false, true>), pgrid, block, 0, st, p, wx, bias);
        }
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
    WDG_CHECK_ARG(dbias && ws, "fused weight gradient needs dbias and scratch");
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    const int nb = (int)std::min<long long>(ntiles, (long long)cus * ((cin == 5 && g_cl1_mfma) ? 2 : 3));
    const int rows = 9 * cin + 1, cols = 3 * F;
    const size_t slab = (size_t)((rows + 15) / 16 * 16) * ((cols + 15) / 16 * 16);
    if (ws_bytes < (size_t)nb * slab * sizeof(float)) {
        wdg_set_error("wdg_convlstm1_bwd_wgrad: scratch too small (%zu < %zu)", ws_bytes, (size_t)nb * slab * sizeof(float));
        return WDG_ERR_WORKSPACE;
    }
    p.dWpart = (float*)ws;
    dim3 grid((unsigned)nb), rgrid((unsigned)((rows * cols + 3) / 4));
    if (cin == 2) {
        hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<2, 2, true>), grid, block, 0, st, p, wx, bias);
        hipLaunchKernelGGL((wdg_convlstm1_wgrad_reduce_kernel<2, 2>), rgrid, block, 0, st, p.dWpart, nb, dw, dbias);
    } else {
        if (g_cl1_mfma)
            hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<5, 16, true, true>), grid, block, 0, st, p, wx, bias);
        else
            hipLaunchKernelGGL((wdg_convlstm1_bwd_kernel<5, 16, true>), grid, block, 0, st, p, wx, bias);
        hipLaunchKernelGGL((wdg_convlstm1_wgrad_reduce_kernel<5, 16>), rgrid, block, 0, st, p.dWpart, nb, dw, dbias);
    }
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_convlstm1_bwd(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                                 const float* dh, int lddh, int64_t img_stride_dh, float* dgates, float* dx, int lddx,
                                 int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                                 wdg_stream stream) {
    return cl1_bwd(x, ldx, img_stride_x, wx, bias, dh, lddh, img_stride_dh, dgates, dx, lddx, img_stride_dx, accumulate_dx,
                   n_img, H, W, cin, F, nullptr, nullptr, nullptr, 0, stream);
}

// Input gradient of the LAST cin - dx_c0 channels only: dx[..., 0 : cin - dx_c0] (+)= d(x[..., dx_c0 : cin]).  The discriminator's
// 5 -> 16 layer reads concat(low, high) (models.py:100) and only the two high-resolution channels' gradient is ever used — by the
// gradient penalty (ganbase.py:35) and by the generator step (:60): with dx = the buffer that holds the 2 -> 2 layer's input
// gradient and accumulate_dx = 1 the sum d(high) is complete after this call.  dx_c0: 0 or 3 (cin = 5).
extern "C" int wdg_convlstm1_bwd_dx_from(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                                         const float* dh, int lddh, int64_t img_stride_dh, float* dx, int lddx,
                                         int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                                         int dx_c0, wdg_stream stream) {
    WDG_CHECK_ARG(dx, "null dx");
    return cl1_bwd(x, ldx, img_stride_x, wx, bias, dh, lddh, img_stride_dh, nullptr, dx, lddx, img_stride_dx, accumulate_dx,
                   n_img, H, W, cin, F, nullptr, nullptr, nullptr, 0, stream, nullptr, dx_c0);
}

// As wdg_convlstm1_bwd, plus dw [3][3][cin][4F] += and dbias [4F] += formed in the same pass (no dgates tensor).
// ws: wdg_convlstm1_wgrad_ws_bytes(n_img, H, W, cin, F) bytes of scratch.
extern "C" size_t wdg_convlstm1_wgrad_ws_bytes(int n_img, int H, int W, int cin, int F) {
    const long long ntiles = (long long)n_img * ((H + CL_TH - 1) / CL_TH) * ((W + CL_TW - 1) / CL_TW);
    const size_t slab = (size_t)((9 * cin + 1 + 15) / 16 * 16) * ((3 * F + 15) / 16 * 16);
    return (size_t)std::min<long long>(ntiles, 4096) * slab * sizeof(float);
}

extern "C" int wdg_convlstm1_bwd_wgrad(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                                       const float* dh, int lddh, int64_t img_stride_dh, float* dx, int lddx,
                                       int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                                       float* dw, float* dbias, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(dw && dbias, "null gradient buffers");
    return cl1_bwd(x, ldx, img_stride_x, wx, bias, dh, lddh, img_stride_dh, nullptr, dx, lddx, img_stride_dx, accumulate_dx,
                   n_img, H, W, cin, F, dw, dbias, ws, ws_bytes, stream);
}

// wdg_convlstm1_bwd / wdg_convlstm1_bwd_wgrad with the last x2_n input channels read from a second tensor (WdgCl1::X2);
// dw == NULL: no weight gradient (dbias, ws unused).
extern "C" int wdg_convlstm1_bwd_x2(const float* x, int ldx, int64_t img_stride_x, const float* x2, int ldx2, int64_t img_stride_x2, int x2_n,
                                    const float* wx, const float* bias, const float* dh, int lddh, int64_t img_stride_dh, float* dx,
                                    int lddx, int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                                    float* dw, float* dbias, void* ws, size_t ws_bytes, wdg_stream stream) {
    const Cl1X2 s2 = {x2, ldx2, img_stride_x2, x2_n};
    WDG_CHECK_ARG(!dw || dbias, "null gradient buffers");
    return cl1_bwd(x, ldx, img_stride_x, wx, bias, dh, lddh, img_stride_dh, nullptr, dx, lddx, img_stride_dx, accumulate_dx,
                   n_img, H, W, cin, F, dw, dw ? dbias : nullptr, ws, ws_bytes, stream, &s2);
}

