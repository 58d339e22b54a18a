// upconv_fused_h16.hip — inference precision: UpSampling2D(2, 'bilinear') + Conv2DTranspose(5 x 5) of the generator
// (/root/reference/src/downscaling/gan/models.py:62-64) in ONE kernel, column form, the column tensor never leaving the CU.
//
// The column form (upconv_col.hip) is  z[r, (t, o)] = sum_c x[r, c] w[t][o][c]  on the low-resolution grid (a 1 x 1 GEMM with
// 25 * C columns) followed by  y[p, o] = act(bias + sum k_r(a) k_r(b) z[r, (t, o)])  over the (r, t, a, b) with
// 2 r - 1 + (a, b) + t - 2 = p.  As two launches, z — 400 floats per low-resolution pixel, 1.4 GB for a 16-tile group of the
// shipped generator — is written once and read once with a 2.25 x halo: both launches run at the memory system's rate
// (0.48 + 0.54 ms of the group's 3.5 ms; the 16-bit MFMA work of the GEMM is ~0.1 ms).  A 16-bit z was tried twice in
// round 3 and lost to its access granularity.  Here a PERSISTENT workgroup (one per CU: 135 KB of LDS; eight waves) walks over
// 8 x 8 low-resolution tiles (16 x 16 output pixels); per tile
//   * the 12 x 12 window of x (K = 160 channels) is loaded in one round trip of branch-free buffer loads and staged ONCE,
//     rounded to the operand format, as [k-octet][pixel] 16-byte slots (pixel index XOR-swizzled by the octet: conflict-free
//     staging stores and fragment reads); every wave then takes the pixel fragments it needs into registers for the whole tile
//     (they do not depend on the tap row);
//   * per tap row ty the weights' 80 x K slice streams into LDS (a straight copy of the layer's weight tensor viewed as
//     [25 * C][K], requested a tap row ahead), the 144 x 80 slice of z is formed by 45 MFMA tiles — six waves own 3 pixel tiles
//     x 2 column tiles, two waves the fifth column tile — and written to the LDS window the gather passes of upconv_col.hip read
//     (horizontal pass -> H, vertical pass -> the thread's output pixel);
//   * bias, LeakyReLU and the inference BatchNorm affine in the epilogue, as in wdg_upconv_gather.
// Same rounding points as the two-launch route (x and W rounded to nearest even, fp32 accumulation, fp32 z and interpolation);
// the GEMM is computed on the window, i.e. 2.25 x the multiply-adds — at 16-bit MFMA rates that is cheaper than z's traffic.
// Measured (16-tile group of the shipped generator, same box): 0.94 ms for the two launches, 0.74 ms fused (DESIGN 10.4 has the
// steps: what decided it were registers — the compiler hoisted ~130 registers of thread-index arithmetic out of the tile loop
// until that arithmetic was re-derived per iteration from an opaque value — and LDS operand re-reads, not HBM latency).
// -DFV_PROF=1 / -DFV_SKIP=bits are measurement builds (tools/prof_fused.py).
#include "common.h"
#include "h16.h"
#include <algorithm>
#include <type_traits>

#ifndef FV_SKIP
#define FV_SKIP 0                     // measurement builds: 1 no MFMA phase, 2 no horizontal pass, 4 no vertical pass, 8 no window loads
#endif
#ifndef FV_PROF
#define FV_PROF 0                     // measurement builds: per-phase shader-clock totals of workgroup 0 / wave 0 (wdg_fv_prof)
#endif
#if FV_PROF
__device__ unsigned long long fv_prof[8];
#define FV_MARK(k) do { const long long now_ = clock64(); pr[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define FV_MARK(k) do {} while (0)
#endif
namespace {
constexpr int F_TS = 8;                      // tile edge on the low-res grid
constexpr int F_ZW = F_TS + 4;               // window edge (12)
constexpr int F_NPX = F_ZW * F_ZW;           // 144 window pixels = 9 MFMA pixel tiles
constexpr int F_CQ = 4;                      // output channels / 4 (C = 16)
constexpr int F_PX = 5 * F_CQ + 1;           // float4 pitch of a window pixel's tap-row slice of z (20 used; 21: the MFMA epilogue's
                                             // 16 lanes of one register quad land in 16 different bank quads, at 20 only four)
constexpr int F_HP = F_CQ + 1;               // H pixel pitch (see wdg_upconv_gather_kernel)
constexpr int F_NCOL = 5 * 4 * F_CQ;         // 80 columns of z per tap row
constexpr int F_NT = 512;                    // threads: 133 KB of LDS leave ONE workgroup per CU — eight waves (two per SIMD) give its phases
                                             // (staging, MFMA tiles, the two gather passes, four barriers per tap row) something to overlap with
constexpr int F_NW = F_NT / 64;
constexpr int F_ZSLOTS = F_NPX * (5 * F_CQ + 1);   // 16-byte slots of a z buffer (144 pixels x pitch 21)
constexpr int F_SP = 24;                     // pixels / weight columns per staging pass

__device__ __forceinline__ float f_coef(int r, int a, int Hl) {
    const int q = 2 * r - 1 + a;                      // (selects, no early return: the callers sit in straight-line code)
    float c = (a == 0 || a == 3) ? 0.25f : 0.75f;
    c += (q == 0 || q == 2 * Hl - 1) ? 0.25f : 0.f;
    return (unsigned)q < (unsigned)(2 * Hl) ? c : 0.f;
}

template <int FMT, int K>
__global__ void __launch_bounds__(F_NT) wdg_upconv_fused_h16_kernel(const float* __restrict__ x, int ldx, long long isx,
                                                                   const wdg_h16<FMT>* __restrict__ w16, const float* __restrict__ bias,
                                                                   const float* __restrict__ affine, float* __restrict__ y, int ldy,
                                                                   long long isy, int Hl, int Wl, int tiles, int total, int act, float slope, int out16, int in16) {
    typedef wdg_h16x8<FMT> h16x8;
    static_assert(K % 32 == 0, "whole MFMA K-steps");
    constexpr int KO = K / 8, KS = K / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [x window / second z buffer | weights of a tap row | z | H]: once every wave holds its pixel fragments in registers the
    // window is dead, and its LDS becomes the buffer the MFMA phase of tap row ty + 1 writes while the gather of ty reads the other
    h16x8* Xs = reinterpret_cast<h16x8*>(smem);                       // [KO][144]
    f32x4* Zb = reinterpret_cast<f32x4*>(smem);                       // [144][21] (z of odd tap rows; aliases Xs)
    h16x8* Ws = Xs + F_ZSLOTS;                                        // [KO][80]
    f32x4* Za = reinterpret_cast<f32x4*>(Ws + KO * F_NCOL);           // [144][21] (z of even tap rows)
    f32x4* Hs = Za + F_ZSLOTS;                                        // [12 * 16][5]
    static_assert(KO * F_NPX <= F_ZSLOTS, "the window fits the z buffer it becomes");

    // Everything derived from the thread index is RE-derived at the top of each tile and each tap row from a value the compiler
    // cannot see through: left alone it hoists ~130 registers of loop-invariant addresses out of the two loops (the blocks
    // themselves use < 100), which is what stood between this kernel and keeping more operands in registers.
    int t = threadIdx.x, li, lq, wave, s_oct, s_p0, s_wy0, s_wx, qyl, qxl, og0, h_hq;
    bool s_on;
    const int tiles_x = (Wl + F_TS - 1) / F_TS;
    const wdg_srd srdW = wdg_make_srd(w16);

    // ---- staging geometry: F_SP = 24 pixels (or weight columns) x KO octets per pass over the first 480 threads — a thread keeps
    // ONE octet and ONE window column, pass u moves it two window rows down (24 = 2 x 12): nothing per-slot to keep in registers
    static_assert(KO * F_SP <= F_NT && F_NPX % F_SP == 0 && F_SP == 2 * F_ZW, "staging pattern");
    constexpr int OQ = F_CQ * 256 / F_NT;            // channel groups per thread (the threads beyond 256 take the upper groups)
    auto derive = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(t));
        li = t & 15, lq = (t >> 4) & 3, wave = t >> 6;
        s_oct = t % KO, s_p0 = t / KO;               // (s_p0 < F_SP for the staging threads)
        s_on = t < KO * F_SP;
        s_wy0 = s_p0 / F_ZW, s_wx = s_p0 - s_wy0 * F_ZW;
        qyl = (t & 255) >> 4, qxl = t & 15, og0 = (t >> 8) * OQ;      // this thread's output pixel within the 16 x 16 tile
        h_hq = (t / F_CQ) % (2 * F_TS);              // its output column in the horizontal pass
    };
    derive();
    // weights of a tap row: registers -> LDS one phase later
    constexpr int W_LD = (F_NCOL + F_SP - 1) / F_SP;
    u32x4 wr[W_LD];
    auto fetch_w = [&](int ty) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            const int c = s_p0 + F_SP * i;
            const unsigned bad = (unsigned)((KO * F_SP - 1 - t) | (F_NCOL - 1 - c)) & 0x80000000u;   // (arithmetic, see fetch_x)
            wr[i] = __builtin_amdgcn_raw_buffer_load_b128(srdW, (int)((((unsigned)(ty * F_NCOL + c) * K + s_oct * 8) << 1) | bad), 0, 0);
        }
    };
    auto store_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            const int c = s_p0 + F_SP * i;
            if (s_on && c < F_NCOL) Ws[s_oct * F_NCOL + (c ^ (s_oct & 7))] = __builtin_bit_cast(h16x8, wr[i]);
        }
    };
    // window of x: the whole window in ONE round trip of X_B slots per thread
    constexpr int X_B = F_NPX / F_SP;
    f32x4 xv[X_B][2];
    auto fetch_x = [&](int work) __attribute__((always_inline)) {
        const int n = work / tiles, tile = work - n * tiles;
        const int i0 = (tile / tiles_x) * F_TS, j0 = (tile % tiles_x) * F_TS;
        const int gx = j0 - 2 + s_wx, gy0 = i0 - 2 + s_wy0;
        // (descriptor per image; the padding gets bit 31 of its offset set ARITHMETICALLY — any of the range checks negative -> out
        // of the descriptor's range -> zeros.  Loads under `if (inside)`, and `inside ? off : OOB` selects alike, come out as
        // exec-masked blocks with an s_waitcnt each; this way the twelve loads are in flight together)
        const int colbad = gx | (Wl - 1 - gx) | (KO * F_SP - 1 - t);
        if (in16) {
            // x in the operand format already (rounded by its producers where this kernel would have): one request per slot
            const wdg_srd srdX = wdg_make_srd(reinterpret_cast<const wdg_h16<FMT>*>(x) + (long long)n * isx);
            const int off0 = ((gy0 * Wl + gx) * ldx + s_oct * 8) * 2, rs = 2 * Wl * ldx * 2;
#pragma unroll
            for (int u = 0; u < X_B; ++u) {
                const int gy = gy0 + 2 * u;
                xv[u][0] = wdg_buffer_load_f32x4(srdX, (unsigned)(off0 + u * rs) | ((unsigned)(colbad | gy | (Hl - 1 - gy)) & 0x80000000u));
            }
            return;
        }
        const wdg_srd srdX = wdg_make_srd(x + (long long)n * isx);
        const int off0 = ((gy0 * Wl + gx) * ldx + s_oct * 8) * 4, rs = 2 * Wl * ldx * 4;
#pragma unroll
        for (int u = 0; u < X_B; ++u) {
            const int gy = gy0 + 2 * u;
            const unsigned off = (unsigned)(off0 + u * rs) | ((unsigned)(colbad | gy | (Hl - 1 - gy)) & 0x80000000u);
            xv[u][0] = wdg_buffer_load_f32x4(srdX, off);
            xv[u][1] = wdg_buffer_load_f32x4(srdX, off + 16);
        }
    };
    auto store_x = [&]() __attribute__((always_inline)) {
        if (s_on) {
#pragma unroll
            for (int u = 0; u < X_B; ++u)
                Xs[s_oct * F_NPX + ((s_p0 + F_SP * u) ^ (s_oct & 7))] = in16 ? __builtin_bit_cast(h16x8, xv[u][0]) : wdg_pack_h16<FMT>(xv[u][0], xv[u][1]);
        }
    };
    static_assert(F_NW == 8 && F_NPX == 9 * 16, "3 groups of 3 pixel tiles x (2 + 2 + 1) column tiles over 6 + 2 waves");
    h16x8 bfr[5][KS];                                // the fragments of this wave's three (waves 0..5) or five (6, 7) pixel tiles, loaded once per tile
#if FV_PROF
    long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
    int work = blockIdx.x;
    if (work >= total) return;
    fetch_w(0);
    for (; work < total; work += gridDim.x) {
        const int n = work / tiles, tile = work - n * tiles;
        const int i0 = (tile / tiles_x) * F_TS, j0 = (tile % tiles_x) * F_TS;
        const int next = work + gridDim.x;
        derive();
        // (the window is requested where it is needed: requesting it a tile ahead — 48 registers across the whole tile — was
        // measured neutral: the kernel's phases are latency chains between barriers, not this round trip)
        if (!(FV_SKIP & 8)) fetch_x(work);
        store_x();                                   // (Xs is free: the previous tile's last MFMA phase ended behind a barrier)
        // horizontal-pass coefficients of this thread for this tile.  Its outputs i = t + F_NT k share channel group and output
        // column hq; tap column tx reads upsampled column q = 2 j0 + hq + 2 - tx, i.e. low-res columns rx = (q + 1 - b) / 2 for
        // the two b of q's parity, window column rx - (j0 - 2) (hz below: independent of the tile); the image border doubles /
        // drops taps
        float hc[10];
#pragma unroll
        for (int tx = 0; tx < 5; ++tx) {
            const int sx = h_hq + 3 - tx, q = 2 * j0 + sx - 1;
            const float edge = (q == 0 || q == 2 * Wl - 1) ? 0.25f : 0.f;
            const bool in = (unsigned)q < (unsigned)(2 * Wl);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int b = (sx & 1) + 2 * e;
                const bool on = (unsigned)(((sx - b) >> 1) + 2) < (unsigned)F_ZW;
                hc[2 * tx + e] = in && on ? ((b == 0 || b == 3) ? 0.25f : 0.75f) + edge : 0.f;
            }
        }
        FV_MARK(0);                                  // tile prologue: window load + staging, coefficients
        const int qy = 2 * i0 + qyl, qx = 2 * j0 + qxl;
        f32x4 acc[OQ];
#pragma unroll
        for (int o4 = 0; o4 < OQ; ++o4) acc[o4] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // fragment of k-step ks: octet 4 ks + lq, swizzle (4 ks + lq) & 7 = lq (even ks) or lq + 4 (odd ks)
        auto xfrag = [&](int pt, int ks) __attribute__((always_inline)) {
            return Xs[(ks * 4 + lq) * F_NPX + ((pt * 16 + li) ^ ((ks & 1) * 4 + lq))];
        };
        auto wfrag = [&](int ct, int ks) __attribute__((always_inline)) {
            return Ws[(ks * 4 + lq) * F_NCOL + ((ct * 16 + li) ^ ((ks & 1) * 4 + lq))];
        };
        // ---- 1. z slice of a tap row: 9 pixel tiles x 5 column tiles (= tx).  Operand fragment reads were 2.3 of the kernel's
        // 3.3 MB of LDS traffic per tile, so the tiles are blocked for register reuse: waves 0..5 own THREE pixel tiles (group
        // g = wave % 3) x TWO column tiles (pair wave / 3) and keep the pixel fragments — which do not depend on the tap row — in
        // registers for the whole tile: per tap row they read two column tiles' weights for six MFMA tiles.  The fifth column tile
        // goes to waves 6 (pixel tiles 0..4) and 7 (5..8), organised the same way.  Per tap row 14 fragment sets are read
        // instead of 90, and every wave's phase is one read - MFMA - store sequence.
        auto mfma_phase = [&](f32x4* Z) __attribute__((always_inline)) {
            auto zstore = [&](int pt, int ct, f32x4 c) __attribute__((always_inline)) {
                // register r of lane (li, lq): column ct * 16 + 4 lq + r (= output channel 4 lq + r of tap column tx = ct) of pixel li
                Z[(pt * 16 + li) * F_PX + ct * F_CQ + lq] = c;
            };
            if (FV_SKIP & 1) return;
            if (wave < 6) {
                const int g = wave % 3, ct0 = 2 * (wave / 3);
                f32x4 c[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) c[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                h16x8 a[2][KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a[0][ks] = wfrag(ct0, ks), a[1][ks] = wfrag(ct0 + 1, ks);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) c[i][j] = wdg_mfma16<FMT>(a[i][ks], bfr[j][ks], c[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) zstore(3 * g + j, ct0 + i, c[i][j]);
            } else {
                // fifth column tile: wave 6 pixel tiles 0..4, wave 7 pixel tiles 5..8 (five accumulators; wave 7's fifth is idle)
                const int p0 = wave == 6 ? 0 : 5;
                h16x8 a[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a[ks] = wfrag(4, ks);
                f32x4 c[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) c[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 5; ++j) c[j] = wdg_mfma16<FMT>(a[ks], bfr[j][ks], c[j]);
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (p0 + j < 9) zstore(p0 + j, 4, c[j]);
            }
        };
        // ---- 2. horizontal pass: H[ryl][qxl][o4] = sum_tx sum_{two (rx, b)} k_rx(b) z[ry, rx][(ty, tx), o4]
        auto h_pass = [&](const f32x4* Z) __attribute__((always_inline)) {
            int hz[10];
#pragma unroll
            for (int tx = 0; tx < 5; ++tx) {
                const int sx = h_hq + 3 - tx;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int rxl = ((sx - (sx & 1) - 2 * e) >> 1) + 2;
                    hz[2 * tx + e] = (unsigned)rxl < (unsigned)F_ZW ? rxl * F_PX + tx * F_CQ + (t % F_CQ) : 0;
                }
            }
            for (int i = t; i < ((FV_SKIP & 2) ? 0 : F_ZW * 2 * F_TS * F_CQ); i += F_NT) {
                const f32x4* zr = Z + (i / (F_CQ * 2 * F_TS)) * F_ZW * F_PX;
                f32x4 h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 10; ++j) h += hc[j] * zr[hz[j]];
                Hs[(i / F_CQ) * F_HP + (i % F_CQ)] = h;
            }
        };
        // ---- 3. vertical pass: the two (ry, a) pairs of a tap row
        auto v_pass = [&](int ty) __attribute__((always_inline)) {
            const int sy = qy + 3 - ty;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int a = (sy & 1) + 2 * e;
                const int ry = (sy - a) >> 1;
                const int ryl = ry - (i0 - 2);
                if (!(FV_SKIP & 4) && (unsigned)ryl < (unsigned)F_ZW) {
                    const float c = f_coef(ry, a, Hl);
#pragma unroll
                    for (int o4 = 0; o4 < OQ; ++o4) acc[o4] += c * Hs[(ryl * 2 * F_TS + qxl) * F_HP + og0 + o4];
                }
            }
        };

        // Software pipeline over the tap rows: the MFMA phase of row ty + 2 and the horizontal gather of row ty + 1 are
        // independent (two z buffers) and share one phase, the vertical gather of row ty shares the other with the staging of
        // the next weights — as five phases in sequence each was a latency chain between barriers at two waves per SIMD (MFMA
        // 27 %, horizontal gather 27 %, barriers 16 % of the kernel's clocks).  Two barriers per tap row:
        //   Z(ty): H of row ty complete; z of row ty + 1 complete; the weights' LDS free
        //   Y(ty): weights of row ty + 2 staged; every wave done with H (vertical gather of row ty)
        store_w();                                   // weights of tap row 0 (requested a tile ahead)
        FV_MARK(1);
        __syncthreads();
        FV_MARK(2);
        fetch_w(1);
        if (wave < 6) {
            const int g = wave % 3;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bfr[j][ks] = xfrag(3 * g + j, ks);
        } else {
            const int p0 = wave == 6 ? 0 : 5;
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bfr[j][ks] = xfrag(p0 + j < 9 ? p0 + j : 8, ks);
        }
        mfma_phase(Za);                              // row 0
        FV_MARK(3);
        __syncthreads();                             // (every wave holds its fragments: the window's LDS is the second z buffer now)
        store_w();                                   // weights of row 1
        __syncthreads();
        fetch_w(2);
        mfma_phase(Zb);                              // row 1
        h_pass(Za);                                  // row 0
        FV_MARK(4);
#pragma unroll 1
        for (int ty = 0; ty < 5; ++ty) {
            __syncthreads();                         // Z(ty)
            FV_MARK(5);
            derive();
            v_pass(ty);
            if (ty < 3) store_w();                   // weights of row ty + 2
            if (ty < 4) {
                __syncthreads();                     // Y(ty)
                if (ty < 3) {
                    if (ty + 3 < 5) fetch_w(ty + 3);
                    else if (next < total) fetch_w(0);
                    mfma_phase((ty & 1) ? Zb : Za);  // row ty + 2
                }
                h_pass((ty & 1) ? Za : Zb);          // row ty + 1
            }
            FV_MARK(6);
        }
        if (qy < 2 * Hl && qx < 2 * Wl) {
            f32x4 v[OQ];
#pragma unroll
            for (int o = 0; o < OQ; ++o) {
                const int o4 = og0 + o;
                v[o] = acc[o];
                if (bias) v[o] += *reinterpret_cast<const f32x4*>(bias + 4 * o4);
                if (act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[o][r] = wdg_lrelu(v[o][r], slope);
                }
                if (affine) v[o] = v[o] * *reinterpret_cast<const f32x4*>(affine + 4 * o4) + *reinterpret_cast<const f32x4*>(affine + 4 * F_CQ + 4 * o4);
            }
            if (out16) {
                // the thread's eight consecutive channels in the operand format (the only reader rounds to it anyway): 16 bytes
                static_assert(OQ == 2, "eight channels per thread");
                wdg_h16<FMT>* dst = reinterpret_cast<wdg_h16<FMT>*>(y) + (long long)n * isy + ((long long)qy * (2 * Wl) + qx) * ldy;
                *reinterpret_cast<h16x8*>(dst + 4 * og0) = wdg_pack_h16<FMT>(v[0], v[1]);
            } else {
                float* dst = y + (long long)n * isy + ((long long)qy * (2 * Wl) + qx) * ldy;
#pragma unroll
                for (int o = 0; o < OQ; ++o) *reinterpret_cast<f32x4*>(dst + 4 * (og0 + o)) = v[o];
            }
        }
        FV_MARK(7);                                  // epilogue
    }
#if FV_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int k = 0; k < 8; ++k) fv_prof[k] = pr[k];
#endif
}

template <int FMT, int K>
int fused_launch(int grid, size_t lds, hipStream_t st, const float* x, int ldx, long long isx, const void* w16, const float* bias,
                 const float* affine, float* y, int ldy, long long isy, int Hl, int Wl, int tiles, int total, int act, float slope, int out16, int in16) {
    static bool attr = false;
    if (!attr) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_upconv_fused_h16_kernel<FMT, K>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((wdg_upconv_fused_h16_kernel<FMT, K>), dim3(grid), dim3(F_NT), lds, st, x, ldx, isx,
                       reinterpret_cast<const wdg_h16<FMT>*>(w16), bias, affine, y, ldy, isy, Hl, Wl, tiles, total, act, slope, out16, in16);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
int fused_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}
}  // namespace

#if FV_PROF
extern "C" int wdg_fv_prof(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fv_prof), sizeof(fv_prof)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int wdg_upconv_fused_h16_supported(int Cin, int C) { return C == 16 && Cin == 160; }

// y [n, 2 Hl, 2 Wl, >= C] = affine(act(bias + convT5x5(bilinear_x2(x_low)))), x_low [n, Hl, Wl, ldx >= Cin] fp32 (rounded to the
// operand format while staged), w16 = the layer's weight tensor [25 * C][Cin] in bf16 (fmt 0) / fp16 (fmt 1).
// out16 != 0: y holds 16-bit elements of the operand format (ldy / img_stride_y in elements, ldy % 8 == 0) — for a reader that
// rounds to that format anyway (wdg_conv_thin16_fwd_h16): the same values, half the bytes.  in16 != 0: x_low likewise (written by
// wdg_conv_fwd_h16_act16 / wdg_conv_dgrad_h16_act16 with out16).
extern "C" int wdg_upconv_fused_h16(const void* x_low_, int ldx, int64_t img_stride_x, const void* w16, int fmt, const float* bias,
                                    const float* affine, void* y_, int ldy, int64_t img_stride_y, int n_img, int Hl, int Wl, int Cin,
                                    int C, int act, float slope, int out16, int in16, wdg_stream stream) {
    float* y = reinterpret_cast<float*>(y_);
    const float* x_low = reinterpret_cast<const float*>(x_low_);
    WDG_CHECK_ARG(!in16 || ldx % 8 == 0, "16-bit input: pixel stride a multiple of 8 elements");
    WDG_CHECK_ARG(!out16 || ldy % 8 == 0, "16-bit output: pixel stride a multiple of 8 elements");
    WDG_CHECK_ARG(x_low && w16 && y && (fmt == 0 || fmt == 1) && n_img > 0 && n_img < 65536 && Hl > 0 && Wl > 0, "bad argument");
    WDG_CHECK_ARG(wdg_upconv_fused_h16_supported(Cin, C), "unsupported channel counts (Cin 160, C 16)");
    WDG_CHECK_ARG(ldx % 4 == 0 && ldx >= Cin && ldy % 4 == 0 && ldy >= C, "pixel strides");
    WDG_CHECK_ARG(((uintptr_t)x_low & 15) == 0 && ((uintptr_t)w16 & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
                  ((uintptr_t)affine & 15) == 0, "x / w / y / bias / affine must be 16-byte aligned");
    constexpr int K = 160;
    const size_t lds = ((size_t)2 * F_ZSLOTS + (size_t)(K / 8) * F_NCOL + (size_t)F_ZW * 2 * F_TS * F_HP) * 16;   // 137,728 B
    const int tiles = ((Hl + F_TS - 1) / F_TS) * ((Wl + F_TS - 1) / F_TS);
    WDG_CHECK_ARG((long long)tiles * n_img < (1ll << 31), "too many tiles");
    WDG_CHECK_ARG((long long)Hl * Wl * ldx * 4 < (1ll << 31), "an image must stay below 2 GiB (buffer descriptor per image)");
    const int total = tiles * n_img;
    const int grid = std::min(total, fused_cus());  // persistent: one workgroup per CU (LDS), each prefetching its next tile's window
    hipStream_t st = (hipStream_t)stream;
    if (fmt == 0)
        return fused_launch<0, K>(grid, lds, st, x_low, ldx, img_stride_x, w16, bias, affine, y, ldy, img_stride_y, Hl, Wl, tiles, total, act, slope, out16, in16);
    return fused_launch<1, K>(grid, lds, st, x_low, ldx, img_stride_x, w16, bias, affine, y, ldy, img_stride_y, Hl, Wl, tiles, total, act, slope, out16, in16);
}
