// conv_patch_h16.hip — inference-precision forward convolution with the input PATCH of an output tile resident in LDS
// (bf16 / fp16 operands on v_mfma_f32_16x16x32, fp32 accumulation, fp32 activations in memory).
//
// Why: the 16-bit MFMA finishes a 128x128x64 K-step in ~0.2 us, so conv_igemm_bf16.hip — which gathers every tap's
// operand rows from L1 / L2 again (kh*kw / stride^2 = 9-16 reads of each input value per workgroup) — is bound by the
// L2 -> CU path (~8-10 of its ~17 TB/s, profiles/r02j_pmc_infer_bf16.txt), not by the matrix cores.  Here a workgroup
// owns a TH x TW patch of output pixels of ONE image x BN output channels:
//   * the (TH-1)*s+kh x (TW-1)*s+kw input patch (a chunk of its channels) is read ONCE, rounded to the 16-bit format and
//     stored in LDS as [channel group of 8][row][column parity][column / stride] 16-byte slots: the 16 pixels of an MFMA
//     fragment read consecutive (or bank-disjoint) slots for every tap, so the tap loop is ds_read_b128 + MFMA only;
//   * the 16-bit weights [Cout][taps][Cin_p] stream through a double-buffered LDS stage of two K-steps (2 x 32 reduction
//     indices x BN rows), fetched one stage ahead into registers;
//   * the MFMA takes the weights as its row operand, so a lane ends up with 4 consecutive channels of one pixel and
//     the epilogue (bias, LeakyReLU, inference BatchNorm affine) stores 16 bytes per lane.
// Same rounding point as conv_igemm_bf16.hip (activations rounded to nearest even while staging), so both satisfy the
// same parity tests (tests/test_bf16_gpu.py); the dispatcher in conv_igemm_bf16.hip picks this kernel when the output map
// divides into one of the two tile shapes below and falls back to the gather kernel otherwise.
//
// Fragment shapes (16 pixels each): 1 x 16 (maps whose width divides by 16) or 4 x 4 (width divides by 24: the 24 x 24 maps
// of the shipped 96-pixel generator).  ds_read_b128 serves lanes {0-3, 12-15, 20-27} in one LDS cycle, i.e. pixels 0-3 and
// 12-15 of one channel group with pixels 4-11 of the next: conflict-free when the channel-group pitch is a multiple of
// 16 slots and, for 4 x 4 fragments, the slot distance of two patch rows is 4 or 12 modulo 16 (the planner pads for it).
#include "conv_plan.h"
#include "h16.h"
#include <algorithm>
#include <cstring>

struct WdgPatchH16 {
    const float* A;
    const void* B;
    float* Out;
    const float* bias;
    const float* affine;       // optional [2*Ncols]: scale | shift applied after the activation
    long long imgStrideA, imgStrideO;
    int H, W, ldA, Ho, Wo, ldO;
    int Ncols, ldB, Cin_p;
    int kh, kw, sshift;        // stride = 1 << sshift
    int pad_h, pad_w;
    int act, accumulate;
    float slope;
    int in16;                  // the input is stored in the 16-bit operand format (A points to 16-bit elements; ldA / imgStrideA in
                               // elements): the producer rounded the activation where this kernel would have — same bits, half the bytes
    int out16;                 // the output is stored in the 16-bit operand format (Out points to 16-bit elements; ldO / imgStrideO in
                               // elements): the column GEMM of the upsample layer, whose only reader is the bilinear gather
    int shufC;                 // > 0: transposed k x k stride-k convolution as a 1 x 1 GEMM with k * k * shufC columns — column n is tap
                               // n / shufC (ky = tap / shufK, kx = tap % shufK), channel n % shufC of output pixel (k y + ky, k x + kx): the
                               // epilogue scatters ("pixel shuffle"); bias / affine are indexed by the channel.  Wo here = low-res width
    int shufK;
    int gate_F;                // > 0: ConvLSTM gate columns interleaved — column n is gate n & 3 of feature n >> 2, i.e. weight row
                               // and bias index (n & 3) * gate_F + (n >> 2); a lane's 4 accumulator registers are then i, f, c~, o
    const float* gates_x;      // LSTM step (template LSTM): input part of the gates, interleaved columns [pixel][4 * gate_F]
    const float* c_prev;       // previous cell state [pixel][ldc] or NULL (zero)
    float* c_out;              // new cell state; Out receives h
    int ldc;
    void* h16_out;             // LSTM step: optional copy of h in the 16-bit operand format [pixel][ldh16] (what the next step and the next
    int ldh16;                 // layer would round h to while staging it: the same bits, half the bytes; Out may then be NULL)
    int lstm_vec;              // LSTM step, NT = 4: whole column tiles, 16-byte aligned c / h rows -> the transposed 16-byte epilogue
    int mt;                    // fragments per wave (host side: picks the instantiation)
    int fw_shift, tfx;         // fragment width 1 << fw_shift (16 or 4), fragments per fragment-row of the tile
    int TH, TW, PH, PW, PWs, pitch;   // tile, patch, columns per parity plane, slots per channel-group plane
    int CK8, nchunk, kcn;      // channel groups per chunk, chunks, K-steps per tap and chunk
    int flat, nent;            // flat: one chunk whose group count is not a multiple of 4 (24 or 40 channels) — the K-steps walk the
                               // flattened (tap, channel group) list, 4 entries each, instead of padding every tap to 4 groups
    int npad, ntab;            // weight stages per chunk rounded up to the pipeline depth; entries of each K-step table
    wdg_fastdiv div_kw, div_kcn;
    int tiles_x, tiles_y, tiles_n, ntn_blk;   // tiles_n counts workgroups along the channels, each doing ntn_blk channel tiles
    wdg_fastdiv div_tn, div_tx, div_ty, div_ck, div_pw;
};

#ifndef WDG_PATCH_PROF
#define WDG_PATCH_PROF 0              // measurement builds: shader-clock totals per phase, summed over the workgroups (wdg_patch_prof)
#endif
#if WDG_PATCH_PROF
__device__ unsigned long long patch_prof[8];
#define PP_MARK(k) do { __builtin_amdgcn_s_waitcnt(0xc07f); const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pp[k] += now_ - pp_last; pp_last = now_; } while (0)
#else
#define PP_MARK(k) do {} while (0)
#endif
// MT fragments of 16 pixels per wave (2 waves along the pixels), NT 16-channel tiles per wave (2 waves along the channels)
// compile-time experiment knobs (tools/build_variants.sh)
#ifndef WDG_PATCH_LB2
#define WDG_PATCH_LB2 1
#endif
#ifndef WDG_PATCH_HG
#define WDG_PATCH_HG ((MT * NT <= 16) ? 2 : 1)
#endif
#ifndef WDG_PATCH_DEPTH
#define WDG_PATCH_DEPTH (LSTM ? 3 : 2)             // weight stages in flight (register sets): deep for the latency-bound recurrent step
#endif
#if WDG_PATCH_LB2
#define WDG_PATCH_BOUNDS __launch_bounds__(256, 2)
#else
#define WDG_PATCH_BOUNDS __launch_bounds__(256)
#endif
// 4 x 4 transpose between a lane's four registers and the four lanes li, li + 16, li + 32, li + 48 (one lane per quarter lq):
// afterwards register j of quarter lq holds what register lq of quarter j held.  v_permlane32_swap exchanges the upper half of
// its first operand with the lower half of its second (register bit 1 <-> lane bit 5), v_permlane16_swap the odd 16-lane rows of
// the first with the even rows of the second (register bit 0 <-> lane bit 4).
__device__ __forceinline__ void wdg_tr4(float (&v)[4]) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    unsigned u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = __builtin_bit_cast(unsigned, v[j]);
    u32x2 r = __builtin_amdgcn_permlane32_swap(u[0], u[2], false, false); u[0] = r[0]; u[2] = r[1];
    r = __builtin_amdgcn_permlane32_swap(u[1], u[3], false, false); u[1] = r[0]; u[3] = r[1];
    r = __builtin_amdgcn_permlane16_swap(u[0], u[1], false, false); u[0] = r[0]; u[1] = r[1];
    r = __builtin_amdgcn_permlane16_swap(u[2], u[3], false, false); u[2] = r[0]; u[3] = r[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(float, u[j]);
}

template <int FMT, int MT, int NT, bool NLOOP, int DBG = 0, int LSTM = 0>
__global__ void WDG_PATCH_BOUNDS wdg_conv_patch_h16_kernel(const WdgPatchH16 p) {
    typedef wdg_h16x8<FMT> h16x8;
    constexpr int BN = 2 * NT * 16;
    constexpr int B_LOADS = 8 * BN / 256;          // 16-byte weight slots per thread and stage (2 K-steps)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    h16x8* ldsB = reinterpret_cast<h16x8*>(smem_raw);              // [2 stages][8 planes][BN]
    h16x8* ldsP = ldsB + 2 * 8 * BN;                                // [CK8][pitch] patch

#if WDG_PATCH_PROF
    long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pp_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;
    const int s = 1 << p.sshift;

    // ---- block -> (image, tile row, tile column, channel tile); channel tiles of one pixel tile run back to back (shared patch)
    int bid;
    {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
        bid = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + i;     // XCD x walks a contiguous tile range
    }
    int rest = (int)wdg_fastdiv_do((unsigned)bid, p.div_tn);
    const int tn = bid - rest * p.tiles_n;
    int r2 = (int)wdg_fastdiv_do((unsigned)rest, p.div_tx);
    const int tx = rest - r2 * p.tiles_x;
    const int img = (int)wdg_fastdiv_do((unsigned)r2, p.div_ty);
    const int ty = r2 - img * p.tiles_y;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    const int iy0 = oy0 * s - p.pad_h, ix0 = ox0 * s - p.pad_w;

    const wdg_srd srdA = p.in16 ? wdg_make_srd(reinterpret_cast<const wdg_h16<FMT>*>(p.A) + (long long)img * p.imgStrideA)
                                : wdg_make_srd(p.A + (long long)img * p.imgStrideA);
    const wdg_srd srdB = wdg_make_srd(p.B);          // (2 GiB window: byte offsets with bit 31 set are out of range and return zeros)

    // ---- per-lane fragment bases (LDS slots) and output pixels
    const int FW = 1 << p.fw_shift, FH = 16 >> p.fw_shift;
    const int fy = li >> p.fw_shift, fx = li & (FW - 1);
    int fbase[MT], opix[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int f = wm * MT + a;
        const int fr = f / p.tfx, fc = f - fr * p.tfx;
        const int oyl = fr * FH + fy, oxl = fc * FW + fx;
        fbase[a] = (oyl << (2 * p.sshift)) * p.PWs + oxl;
        opix[a] = p.shufC ? ((oy0 + oyl) * p.shufK) * (p.Wo * p.shufK) + (ox0 + oxl) * p.shufK : (oy0 + oyl) * p.Wo + ox0 + oxl;
    }

    const int npatch = p.CK8 * p.PH * p.PW;
    const int dummy_slot = p.CK8 * p.pitch;         // one spare slot behind the patch takes the stores of the tail threads
    const int bj = t & 7;                           // this thread's weight slot of a stage: K-step bj >> 2, channel group bj & 3
    float* outImg = p.Out + (long long)img * p.imgStrideO;

    // ---- K-step tables, behind the patch in LDS: entry e = 4 * (K-step of the chunk) + (channel group q of the K-step's 32
    // reduction indices).  tabT: slot offset inside the patch of that (tap, channel group) — what a lane of quarter lq = q adds to
    // its fragment bases; tabK: element offset inside a weight row (without the chunk's first channel).  A K-step walks the taps
    // with kcn K-steps of 4 channel groups each, or (flat: 24 / 40 channels) the flattened (tap, channel group) list 4 entries
    // at a time.  Entries that do not exist (channel group past the chunk, K-step past the end — the table covers the stages the
    // pipeline requests beyond the last) read patch slot 0 against zero weights (offset out of the descriptor's range).
    int* tabT = reinterpret_cast<int*>(ldsP + dummy_slot + 1);
    int* tabK = tabT + p.ntab;
    float* cst = reinterpret_cast<float*>(tabK + p.ntab);          // [3][BN] epilogue constants of the current channel tile
    for (int e = t; e < p.ntab; e += 256) {
        const int ks = e >> 2, q = e & 3;
        int tap, g8, koff;
        bool ok;
        if (p.flat) {
            tap = (int)wdg_fastdiv_do((unsigned)e, p.div_ck);
            g8 = e - tap * p.CK8;
            ok = e < p.nent;
            koff = 8 * e;
        } else {
            tap = (int)wdg_fastdiv_do((unsigned)ks, p.div_kcn);
            g8 = (ks - tap * p.kcn) * 4 + q;
            ok = tap < p.kh * p.kw && g8 < p.CK8;
            koff = tap * p.Cin_p + g8 * 8;
        }
        const int ky = (int)wdg_fastdiv_do((unsigned)tap, p.div_kw), kx = tap - ky * p.kw;
        tabT[e] = ok ? (((ky << p.sshift) + (kx & (s - 1))) * p.PWs) + (kx >> p.sshift) + g8 * p.pitch : 0;
        tabK[e] = (ok && !(DBG & 1)) ? koff : 0x40000000;
    }

    PP_MARK(0);                                      // index arithmetic + K-step tables
    // ---- channel tiles of this workgroup: one, or (ntn_blk > 1: a patch that holds every channel, shallow reductions) several
    // against the same resident patch
    // (NLOOP is a template parameter: the single-tile kernels keep their register allocation)
    for (int tni = 0; tni < (NLOOP ? p.ntn_blk : 1); ++tni) {
    const int n0 = NLOOP ? tni * BN : tn * BN;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- this thread's weight slots of a stage: j = (K-step of the stage, channel group q), column n
    unsigned b_off[B_LOADS];   // byte offset of weight row n (columns past the layer's last: row 0 — their accumulators are never stored)
    int b_slot[B_LOADS];       // LDS slot inside a stage
#pragma unroll
    for (int r = 0; r < B_LOADS; ++r) {
        const int n = (t >> 3) + 32 * r;
        const int ng = n0 + n;
        const int wrow = p.gate_F ? (ng & 3) * p.gate_F + (ng >> 2) : ng;
        b_off[r] = (ng < p.Ncols) ? (unsigned)(wrow * p.ldB) << 1 : 0u;
        b_slot[r] = bj * BN + (n ^ bj);
    }

    // LSTM step: the epilogue's inputs (input part of the gates, previous cell state), requested in ONE batch
    f32x4 gx[LSTM ? NT : 1][LSTM ? MT : 1];
    float cp[LSTM ? NT : 1][LSTM ? MT : 1];
    bool gx_loaded = false;
    auto load_cell_inputs = [&]() {
        if constexpr (LSTM) {
            const int nw0_ = n0 + wn * (BN / 2);
            if constexpr (NT == 4) {
                if (p.lstm_vec) {
                    // the quarter's four consecutive features of the previous cell state in ONE 16-byte request per fragment, handed to
                    // the lanes that hold those features' gates by the lane transpose (instead of NT 4-byte requests)
#pragma unroll
                    for (int a = 0; a < MT; ++a) {
                        const long long pix = (long long)img * p.Ho * p.Wo + opix[a];
                        float c4[4] = {0.f, 0.f, 0.f, 0.f};
                        if (p.c_prev && !(DBG & 64)) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(p.c_prev + pix * p.ldc + (nw0_ >> 2) + 4 * lq);
#pragma unroll
                            for (int j = 0; j < 4; ++j) c4[j] = q[j];
                        }
                        wdg_tr4(c4);
#pragma unroll
                        for (int b = 0; b < NT; ++b) {
                            cp[b][a] = c4[b];
                            gx[b][a] = (DBG & 64) ? (f32x4){0.1f, 0.2f, 0.3f, 0.4f}
                                                  : *reinterpret_cast<const f32x4*>(p.gates_x + pix * (4 * p.gate_F) + nw0_ + b * 16 + 4 * lq);
                        }
                    }
                    gx_loaded = true;
                    return;
                }
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = nw0_ + b * 16 + 4 * lq;
                const int f = n >> 2;
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const long long pix = (long long)img * p.Ho * p.Wo + opix[a];
                    if constexpr (DBG & 64) { gx[b][a] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f}; cp[b][a] = 0.5f; continue; }
                    const bool on = n < p.Ncols;
                    gx[b][a] = on ? *reinterpret_cast<const f32x4*>(p.gates_x + pix * (4 * p.gate_F) + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    cp[b][a] = (on && p.c_prev) ? p.c_prev[pix * p.ldc + f] : 0.f;
                }
            }
            gx_loaded = true;
        }
    };
    for (int ck = 0; ck < p.nchunk; ++ck) {
        __syncthreads();                             // every wave is done with the previous chunk's patch, weight stages and epilogue tiles
                                                     // (first chunk: the K-step tables are complete)
        if constexpr (!LSTM) {
            if (ck == 0) {
                // ---- epilogue constants of this channel tile -> LDS: bias | scale | shift per column (columns past the layer's last:
                // 0 | 1 | 0), published by the barrier behind the patch.  The epilogue used to fetch them from global memory per column
                // tile — twelve scalar loads behind validity branches, each with its own wait, four tiles in sequence: most of the
                // 14,000 clocks a workgroup of the first layer spent between its last MFMA and its last store
                // (profiles/r06f_patch_phases.txt).
                const int naff = p.shufC ? p.shufC : p.Ncols;
                for (int c = t; c < BN; c += 256) {
                    const int n = n0 + c;
                    const bool on = n < p.Ncols;
                    const int nb = p.gate_F ? (n & 3) * p.gate_F + (n >> 2) : p.shufC ? n % p.shufC : n;
                    cst[c] = (on && p.bias) ? p.bias[nb] : 0.f;
                    cst[BN + c] = (on && p.affine) ? p.affine[nb] : 1.f;
                    cst[2 * BN + c] = (on && p.affine) ? p.affine[naff + nb] : 0.f;
                }
            }
        }
        // ---- patch chunk: global fp32 -> 16-bit -> LDS, PU slots (2 x 16-byte loads each) per thread in flight
        constexpr int PU = (MT * NT >= 24) ? 8 : (WDG_PATCH_DEPTH > 2) ? 6 : 10;   // (the 6 x 4 tile has no registers to spare; nor a deep weight pipeline)
        if (p.in16) {
            // 16-bit activations: a slot is ONE 16-byte request and goes to LDS as it is
            for (int base = (!NLOOP || tni == 0) ? 0 : npatch; base < npatch; base += PU * 256) {
                f32x4 v[PU];
                int slot[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int idx = base + u * 256 + t;
                    const int pix = (int)wdg_fastdiv_do((unsigned)idx, p.div_ck);
                    const int c = idx - pix * p.CK8;
                    const int y = (int)wdg_fastdiv_do((unsigned)pix, p.div_pw);
                    const int x = pix - y * p.PW;
                    const int gy = iy0 + y, gx = ix0 + x;
                    const bool ok = idx < npatch && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && !(DBG & 2);
                    v[u] = wdg_buffer_load_f32x4(srdA, ok ? (unsigned)((gy * p.W + gx) * p.ldA + (ck * p.CK8 + c) * 8) << 1 : WDG_SRD_OOB);
                    slot[u] = idx < npatch ? c * p.pitch + (((y << p.sshift) + (x & (s - 1))) * p.PWs) + (x >> p.sshift) : dummy_slot;
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) ldsP[slot[u]] = __builtin_bit_cast(h16x8, v[u]);
            }
        } else
        for (int base = (!NLOOP || tni == 0) ? 0 : npatch; base < npatch; base += PU * 256) {
            f32x4 v[PU][2];
            int slot[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int idx = base + u * 256 + t;
                const int pix = (int)wdg_fastdiv_do((unsigned)idx, p.div_ck);
                const int c = idx - pix * p.CK8;
                const int y = (int)wdg_fastdiv_do((unsigned)pix, p.div_pw);
                const int x = pix - y * p.PW;
                const int gy = iy0 + y, gx = ix0 + x;
                const bool ok = idx < npatch && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && !(DBG & 2);
                const unsigned off = ok ? (unsigned)((gy * p.W + gx) * p.ldA + (ck * p.CK8 + c) * 8) << 2 : WDG_SRD_OOB;
                v[u][0] = wdg_buffer_load_f32x4(srdA, off);
                v[u][1] = wdg_buffer_load_f32x4(srdA, ok ? off + 16u : WDG_SRD_OOB);
                slot[u] = idx < npatch ? c * p.pitch + (((y << p.sshift) + (x & (s - 1))) * p.PWs) + (x >> p.sshift) : dummy_slot;
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) ldsP[slot[u]] = wdg_pack_h16<FMT>(v[u][0], v[u][1]);
        }

        PP_MARK(1);                                  // chunk barrier + patch: requests, conversion, LDS stores
        // ---- weight stages, D register sets deep: stage s + D is requested while stage s is computed from LDS, and stage s + 1 —
        // requested D - 1 compute stages ago — moves from its registers into the other LDS buffer behind the MFMAs; one barrier
        // per stage.  The body is branch-free and the same for every layer shape: what a K-step reads — the patch slot offset of its
        // (tap, channel group) for the lane's quarter lq, the weight-row offset for the thread's slot bj — comes from the two LDS
        // tables built above (one ds_read each, requested a stage ahead), not from counters with carries, divisions and
        // validity selects in the loop (round 5: ~80 vector + ~25 scalar instructions and a dozen branches per K-step of 16
        // MFMAs — SQ_INSTS_VALU 105 M against 21 M MFMAs on the generator's first layer, profiles/r06a_pmc_infer_bf16.txt; the
        // branches also cost the compiler its exact count of outstanding loads, hence s_waitcnt vmcnt(0) in front of every
        // stage).  The stage count is rounded up to a multiple of D: table entries past the end hold an out-of-range weight
        // offset (zeros, no memory traffic) and patch slot 0.
        constexpr int D = WDG_PATCH_DEPTH;
        const unsigned kb0 = (unsigned)(ck * p.CK8) * 16u;          // this chunk's first channel group, in bytes of a weight row
        u32x4 rb[D][B_LOADS];
        auto ldk = [&](int f) { return ((unsigned)tabK[8 * f + bj] << 1) + kb0; };
        auto fetch_stage = [&](u32x4 (&r_)[B_LOADS], unsigned kq) {
#pragma unroll
            for (int r = 0; r < B_LOADS; ++r) r_[r] = __builtin_amdgcn_raw_buffer_load_b128(srdB, (int)(b_off[r] + kq), 0, 0);
        };
        auto store_stage = [&](const u32x4 (&r_)[B_LOADS], int buf) {
            h16x8* sB = ldsB + buf * 8 * BN;
#pragma unroll
            for (int r = 0; r < B_LOADS; ++r) sB[b_slot[r]] = __builtin_bit_cast(h16x8, r_[r]);
        };
        auto compute_stage = [&](int buf, int off0, int off1) {
            const h16x8* sB = ldsB + buf * 8 * BN;
            // both K-steps' fragments are requested before the first MFMA where the registers allow it (the 6 x 4 tile: one at a time)
            constexpr int HG = WDG_PATCH_HG;
#pragma unroll
            for (int h0 = 0; h0 < 2; h0 += HG) {
                h16x8 af[HG][MT], bf[HG][NT];
                if constexpr (!(DBG & 8)) {
#pragma unroll
                    for (int hh = 0; hh < HG; ++hh) {
                        const int h = h0 + hh;
                        const int pl = h * 4 + lq;
#pragma unroll
                        for (int b = 0; b < NT; ++b) bf[hh][b] = sB[pl * BN + ((wn * (BN / 2) + b * 16 + li) ^ pl)];
#pragma unroll
                        for (int a = 0; a < MT; ++a) af[hh][a] = ldsP[fbase[a] + (h ? off1 : off0)];
                    }
                    // every fragment read of the group stands BEFORE its first MFMA: left to itself the compiler sinks the pixel
                    // fragments to their uses through one register quad (read, s_waitcnt lgkmcnt(0), four MFMAs, read, ...), six
                    // exposed LDS round trips per stage with one wave per SIMD to hide them — the recurrent step spent ~0.9 us per
                    // stage on 0.16 us of matrix work (profiles/r04l_step16_skeletons.txt, r04m_chain_floor.txt)
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int hh = 0; hh < HG; ++hh) {
#pragma unroll
                        for (int b = 0; b < NT; ++b) bf[hh][b] = __builtin_bit_cast(h16x8, (u32x4){(unsigned)off0, (unsigned)b, (unsigned)hh, 1u});
#pragma unroll
                        for (int a = 0; a < MT; ++a) af[hh][a] = __builtin_bit_cast(h16x8, (u32x4){(unsigned)off1, (unsigned)a, (unsigned)hh, 2u});
                    }
                }
                if constexpr (!(DBG & 4)) {
#pragma unroll
                    for (int hh = 0; hh < HG; ++hh)
#pragma unroll
                        for (int a = 0; a < MT; ++a)
#pragma unroll
                            for (int b = 0; b < NT; ++b)
                                acc[a][b] = wdg_mfma16<FMT>(bf[hh][b], af[hh][a], acc[a][b]);
                } else {
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            acc[a][b][0] += (float)bf[0][b][0] + (float)af[HG - 1][a][1];
                }
            }
        };
        const bool bar = !(DBG & 32);
#pragma unroll
        for (int i = 0; i < D; ++i) fetch_stage(rb[i], ldk(i));
        store_stage(rb[0], 0);
        unsigned kq = ldk(D);                                       // weight offset of the next stage to request
        int t0 = tabT[lq], t1 = tabT[4 + lq];                       // patch offsets of the next stage to compute (its two K-steps)
        __syncthreads();
        PP_MARK(2);                                  // first weight stages requested, stage 0 in LDS, patch visible
        for (int st = 0; st < p.npad; st += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int sc = st + u;
                fetch_stage(rb[u], kq);                             // stage sc + D (rb[u]'s stage sc is in LDS already)
                kq = ldk(sc + D + 1);
                const int o0 = t0, o1 = t1;
                t0 = tabT[8 * (sc + 1) + lq];
                t1 = tabT[8 * (sc + 1) + 4 + lq];
                compute_stage((D & 1) ? (sc & 1) : (u & 1), o0, o1);
                store_stage(rb[(u + 1) % D], (D & 1) ? ((sc + 1) & 1) : ((u + 1) & 1));
                if (bar) __syncthreads();
            }
        }
        PP_MARK(3);                                  // the stage loop
    }

    // ---- epilogue.  Lane (li, lq) holds channels 4*lq .. 4*lq+3 of pixel li of every fragment and stores them directly,
    // 16 bytes per lane.  (A transpose through LDS to whole-pixel stores was measured SLOWER: 654 vs 487 us on the 400-column
    // GEMM, profiles/r02n_perf_patch_epilogue.txt.)
    const int nw0 = n0 + wn * (BN / 2);
    if constexpr (LSTM) {
        // ConvLSTM2D cell (gan/models.py:45; Keras hard_sigmoid / tanh, gate order i, f, c, o) on the accumulators: register r of
        // a lane is gate r of feature (n >> 2) of its pixel.  z = recurrent conv + input part; c = f * c_prev + i * tanh(c~);
        // h = o * tanh(c).  Same arithmetic as wdg_lstm_fwd (pointwise.hip) behind an accumulating convolution.
        // EVERY input of the wave's tiles (input part of the gates, previous cell state: MT * NT 16-byte + 4-byte loads) in one
        // memory round trip, then the arithmetic and the stores.  Read per column tile, each tile's loads waited behind
        // the previous tile's stores (possible aliases): NT dependent round trips — 12 of the step's 35 us on their own
        // (profiles/r04m_chain_floor.txt: 14.7 us for a step without its K loop against 3.1 us without the epilogue's memory traffic)
        if (!gx_loaded) load_cell_inputs();
        bool vec_done = false;
        if constexpr (NT == 4) {
            if (p.lstm_vec) {
                // a lane's four column tiles hold features f0 + 4 b + lq of its pixel (b = tile): after the lane transpose quarter lq
                // holds the four CONSECUTIVE features f0 + 4 lq .. + 3 — c and h leave as 16-byte stores (24 four-byte stores per lane
                // before), h optionally also / only in the 16-bit operand format for its readers
                const int f0 = (nw0 >> 2) + 4 * lq;
                wdg_h16<FMT>* h16 = reinterpret_cast<wdg_h16<FMT>*>(p.h16_out);
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const long long pix = (long long)img * p.Ho * p.Wo + opix[a];
                    float cn[4], hv[4];
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        const f32x4 z = acc[a][b] + gx[b][a];
                        const float gi = fminf(fmaxf(__builtin_fmaf(0.2f, z[0], 0.5f), 0.f), 1.f);
                        const float gf = fminf(fmaxf(__builtin_fmaf(0.2f, z[1], 0.5f), 0.f), 1.f);
                        const float gc = wdg_tanh(z[2]);
                        const float go = fminf(fmaxf(__builtin_fmaf(0.2f, z[3], 0.5f), 0.f), 1.f);
                        cn[b] = __builtin_fmaf(gi, gc, gf * cp[b][a]);       // (explicit: the same rounding as the per-tile form below)
                        hv[b] = go * wdg_tanh(cn[b]);
                    }
                    wdg_tr4(cn);
                    wdg_tr4(hv);
                    if constexpr (DBG & 16) { if (cn[0] == 123.456f) p.c_out[pix * p.ldc + f0] = cn[0]; continue; }
                    *reinterpret_cast<f32x4*>(p.c_out + pix * p.ldc + f0) = (f32x4){cn[0], cn[1], cn[2], cn[3]};
                    if (p.Out) *reinterpret_cast<f32x4*>(outImg + (long long)opix[a] * p.ldO + f0) = (f32x4){hv[0], hv[1], hv[2], hv[3]};
                    if (h16) {
                        typedef wdg_h16<FMT> h16x4 __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<h16x4*>(h16 + pix * p.ldh16 + f0) =
                            (h16x4){(wdg_h16<FMT>)hv[0], (wdg_h16<FMT>)hv[1], (wdg_h16<FMT>)hv[2], (wdg_h16<FMT>)hv[3]};
                    }
                }
                vec_done = true;
            }
        }
#pragma unroll
        for (int b = 0; b < (vec_done ? 0 : NT); ++b) {
            const int n = nw0 + b * 16 + 4 * lq;
            if (n >= p.Ncols) continue;
            const int f = n >> 2;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const long long pix = (long long)img * p.Ho * p.Wo + opix[a];
                const f32x4 z = acc[a][b] + gx[b][a];
                const float gi = fminf(fmaxf(__builtin_fmaf(0.2f, z[0], 0.5f), 0.f), 1.f);
                const float gf = fminf(fmaxf(__builtin_fmaf(0.2f, z[1], 0.5f), 0.f), 1.f);
                const float gc = wdg_tanh(z[2]);
                const float go = fminf(fmaxf(__builtin_fmaf(0.2f, z[3], 0.5f), 0.f), 1.f);
                const float cn = __builtin_fmaf(gi, gc, gf * cp[b][a]);
                if constexpr (DBG & 16) { if (cn == 123.456f) p.c_out[pix * p.ldc + f] = cn; continue; }
                p.c_out[pix * p.ldc + f] = cn;
                outImg[(long long)opix[a] * p.ldO + f] = go * wdg_tanh(cn);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = nw0 + b * 16 + 4 * lq;
            if (n >= p.Ncols || (DBG & 16)) continue;
            const bool full = n + 3 < p.Ncols;
            // scattered output (transposed k x k stride-k layer): this quad's tap -> pixel offset, channel
            const int stap = p.shufC ? n / p.shufC : 0;
            const int nch = p.shufC ? n - stap * p.shufC : n;                       // (shufC % 4 == 0: a quad stays inside one tap)
            const int spix = p.shufC ? (stap / p.shufK) * (p.Wo * p.shufK) + stap % p.shufK : 0;
            const int nl = n - n0;                                                  // column inside the workgroup's channel tile
            const f32x4 bias4 = *reinterpret_cast<const f32x4*>(cst + nl), sc4 = *reinterpret_cast<const f32x4*>(cst + BN + nl),
                        sh4 = *reinterpret_cast<const f32x4*>(cst + 2 * BN + nl);
            if (p.out16) {
                // 16-bit result (column count a multiple of 16; no accumulate).  A lane's four channels are 8 bytes — stores of
                // that width ran at 0.6x the rate and made the 16-bit z slower than the fp32 one.  So the column tiles go in pairs:
                // the two lanes that share a pixel's eight consecutive channels (lq even / odd) swap one quad each, the even one
                // stores tile b, the odd one tile b + 1, 16 bytes per lane.  Bias / activation / affine as on the fp32 route (the
                // constants of tile b + 1 are loaded here as well); with a scattering epilogue (shufC % 32 == 0) a pair stays
                // inside one tap.
                static_assert(NT % 2 == 0, "column tiles in pairs");
                if (b & 1) continue;
                typedef wdg_h16<FMT> h16x4 __attribute__((ext_vector_type(4)));
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                wdg_h16<FMT>* out16 = reinterpret_cast<wdg_h16<FMT>*>(p.Out) + (long long)img * p.imgStrideO;
                const bool odd = lq & 1;
                const int nst = nw0 + (b + (odd ? 1 : 0)) * 16 + 8 * (lq >> 1);
                const int nst_ch = p.shufC ? nst - stap * p.shufC : nst;
                const f32x4 bias4b = *reinterpret_cast<const f32x4*>(cst + nl + 16), sc4b = *reinterpret_cast<const f32x4*>(cst + BN + nl + 16),
                            sh4b = *reinterpret_cast<const f32x4*>(cst + 2 * BN + nl + 16);
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    f32x4 v0 = acc[a][b] + bias4, v1 = acc[a][b + 1] + bias4b;
                    if (p.act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = wdg_lrelu(v0[r], p.slope); v1[r] = wdg_lrelu(v1[r], p.slope); }
                    }
                    if (p.affine) { v0 = v0 * sc4 + sh4; v1 = v1 * sc4b + sh4b; }
                    const u32x2 t0 = __builtin_bit_cast(u32x2, (h16x4){(wdg_h16<FMT>)v0[0], (wdg_h16<FMT>)v0[1], (wdg_h16<FMT>)v0[2], (wdg_h16<FMT>)v0[3]});
                    const u32x2 t1 = __builtin_bit_cast(u32x2, (h16x4){(wdg_h16<FMT>)v1[0], (wdg_h16<FMT>)v1[1], (wdg_h16<FMT>)v1[2], (wdg_h16<FMT>)v1[3]});
                    const u32x2 mine = odd ? t1 : t0, give = odd ? t0 : t1;
                    u32x2 got;
                    got[0] = (unsigned)__shfl_xor((int)give[0], 16, 64);
                    got[1] = (unsigned)__shfl_xor((int)give[1], 16, 64);
                    const u32x4 o = odd ? (u32x4){got[0], got[1], mine[0], mine[1]} : (u32x4){mine[0], mine[1], got[0], got[1]};
                    if (nst < p.Ncols) *reinterpret_cast<u32x4*>(out16 + (long long)(opix[a] + spix) * p.ldO + nst_ch) = o;
                }
                continue;
            }
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                float* dst = outImg + (long long)(opix[a] + spix) * p.ldO + nch;
                f32x4 v = acc[a][b] + bias4;
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], p.slope);
                }
                if (p.affine) v = v * sc4 + sh4;
                if constexpr (DBG & 128) { if (v[0] == 123.456f) *reinterpret_cast<f32x4*>(dst) = v; continue; }   // (timing: the arithmetic without the store traffic)
                if (full) {
                    if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.Ncols) dst[r] = p.accumulate ? dst[r] + v[r] : v[r];
                }
            }
        }
    }
    }   // channel tiles
#if WDG_PATCH_PROF
    PP_MARK(4);                                      // epilogue: constants, activation, stores issued
    __builtin_amdgcn_s_waitcnt(0x0070);
    PP_MARK(5);                                      // ... and completed (vmcnt(0))
    if (t == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&patch_prof[k], (unsigned long long)pp[k]);
        atomicAdd(&patch_prof[7], 1ull);
    }
#endif
}
#if WDG_PATCH_PROF
extern "C" int wdg_patch_prof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(patch_prof), sizeof(patch_prof)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(patch_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
static int g_patch_h16 = 1;
static int g_patch_dbg = 0;   // timing experiments (builds with -DWDG_PATCH_EXPERIMENTS): template DBG bit 0 no weight fetch, 1 no patch
                              // loads, 2 no MFMA, 3 no fragment reads, 4 no output stores, 5 no stage barriers
void wdg_patch_h16_set_dbg(int v) { g_patch_dbg = v; }
static int g_patch_flat = 1;             // flattened (tap, channel group) K-steps for 24- / 40-channel layers (tuning key patch_flat)
void wdg_patch_h16_set_flat(int v) { g_patch_flat = v; }
static int g_patch_nloop = 1;            // a workgroup walks all channel tiles of its pixel tile when the patch holds every channel
void wdg_patch_h16_set_nloop(int v) { g_patch_nloop = v; }
static int g_patch_budget = 44 * 1024;     // LDS bytes of a patch chunk: with the 32 KiB of weight stages two workgroups share a CU
void wdg_patch_h16_set(int v) { g_patch_h16 = v; }
void wdg_patch_h16_set_budget(int kib) { g_patch_budget = kib * 1024; }

struct WdgPatchCfg {
    int fw_shift, tfx, TH, TW, MT;
};

// what the kernel needs of a convolution: the forward conv of a plan, or the 1 x 1 transposed conv (= the column GEMM of the
// column-form upsample + transposed-conv layer, upconv_col.hip) whose operand is dy and whose result is dx
struct WdgPatchView {
    int n_img, H, W, ldA, Ho, Wo, ldO;
    long long imgStrideA, imgStrideO;
    int K_p, Ncols;            // padded reduction channels per tap, output channels
    int kh, kw, stride, pad_h, pad_w, cus;
};
static WdgPatchView patch_view(const wdg_conv_plan* pl, bool transposed1x1) {
    const wdg_conv_geom& g = pl->g;
    WdgPatchView v;
    if (!transposed1x1)
        v = {g.n_img, g.H, g.W, g.ldx, g.Ho, g.Wo, g.ldy, g.img_stride_x, g.img_stride_y, pl->Cin_p, g.Cout,
             g.kh, g.kw, g.stride, g.pad_h, g.pad_w, pl->cus};
    else
        v = {g.n_img, g.Ho, g.Wo, g.ldy, g.H, g.W, g.ldx, g.img_stride_y, g.img_stride_x, pl->Cout_p, g.Cin,
             1, 1, 1, 0, 0, pl->cus};
    return v;
}
// ... of a transposed k x k stride-k (non-overlapping) layer: a 1 x 1 GEMM on the low-resolution grid with k * k * Cin columns,
// scattered by the epilogue (WdgPatchH16::shufC)
static WdgPatchView patch_view_shuffle(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    WdgPatchView v = {g.n_img, g.Ho, g.Wo, g.ldy, g.Ho, g.Wo, g.ldx, g.img_stride_y, g.img_stride_x, pl->Cout_p, g.kh * g.kw * g.Cin,
                      1, 1, 1, 0, 0, pl->cus};
    return v;
}

// candidate tile shapes for the output map (most pixels per tile first); returns their number
static int patch_shapes(const WdgPatchView& g, WdgPatchCfg* c) {
    int n = 0;
    if (g.Ho % 8 == 0 && g.Wo % 16 == 0) c[n++] = {4, 1, 8, 16, 4};      // 8 fragments of 1 x 16
    if (g.Ho % 8 == 0 && g.Wo % 24 == 0) c[n++] = {2, 6, 8, 24, 6};      // 12 fragments of 4 x 4
    if (g.Ho % 4 == 0 && g.Wo % 24 == 0) c[n++] = {2, 6, 4, 24, 3};      // 6 fragments of 4 x 4 (strided layers with many channels: smaller patch)
    return n;
}

static int g_patch_lstm_small = 1;       // the recurrent step on the smallest tile shape (tuning key patch_lstm_small)
static int g_patch_lstm_bn = 0;          // recurrent step: channel tile forced to 64 / 128 (0 = the general rule); patch_lstm_small values >= 64
void wdg_patch_h16_set_lstm_small(int v) {
    if (v >= 64) g_patch_lstm_bn = v;
    else if (v == 4 || v == 8) return;       // (rounds 4-5: selected a specialised deep-pipeline instantiation; every instantiation has that loop now)
    else { g_patch_lstm_small = (v & 1) != 0; if (v & 2) g_patch_lstm_bn = 0; }
}
// small_first: the candidate shapes from the smallest — the per-timestep recurrent convolution has few pixel tiles (48 of 8 x 24 on
// the shipped 24 x 24 map x 16 tiles -> 384 workgroups, 1.5 per CU: half the CUs carry two); 4 x 24 gives every CU three
static bool patch_plan(const WdgPatchView& g, WdgPatchH16& p, bool small_first = false) {
    if (!g_patch_h16 || (g.stride != 1 && g.stride != 2) || g.K_p % 8 || g.Ncols < 32 || g.ldO % 4) return false;
    WdgPatchCfg cand[3];
    const int ncand = patch_shapes(g, cand);
    const int s = g.stride;
    const int C8 = g.K_p / 8;
    for (int cj = 0; cj < ncand; ++cj) {
        const int ci = small_first ? ncand - 1 - cj : cj;
        const WdgPatchCfg& c = cand[ci];
        const int PH = (c.TH - 1) * s + g.kh, PW = (c.TW - 1) * s + g.kw;
        int PWs = (PW + s - 1) / s;
        if (c.fw_shift == 2)
            while (((s * s * PWs) & 15) != 4 && ((s * s * PWs) & 15) != 12) ++PWs;    // rows of a 4 x 4 fragment on disjoint banks
        const int pitch = wdg_round_up(PH * s * PWs, 16);
        int CK8 = 0;
        if ((long long)C8 * pitch * 16 <= g_patch_budget) CK8 = C8;
        else
            for (int k = 4; k < C8; k += 4)
                if (C8 % k == 0 && (long long)k * pitch * 16 <= g_patch_budget) CK8 = k;
        if (!CK8) continue;
        p.sshift = s == 2 ? 1 : 0;
        p.fw_shift = c.fw_shift; p.tfx = c.tfx; p.TH = c.TH; p.TW = c.TW; p.mt = c.MT;
        p.PH = PH; p.PW = PW; p.PWs = PWs; p.pitch = pitch;
        p.CK8 = CK8; p.nchunk = C8 / CK8; p.kcn = (CK8 + 3) / 4;
        p.tiles_x = g.Wo / c.TW; p.tiles_y = g.Ho / c.TH;
        return true;
    }
    return false;
}

int wdg_patch_h16_eligible(const wdg_conv_plan* pl) {
    WdgPatchH16 p;
    memset(&p, 0, sizeof(p));
    return patch_plan(patch_view(pl, false), p) ? 1 : 0;
}

// ... of the transposed 1 x 1 view (x side = result, y side = operand)
int wdg_patch_h16_eligible_t(const wdg_conv_plan* pl) {
    WdgPatchH16 p;
    memset(&p, 0, sizeof(p));
    return patch_plan(patch_view(pl, true), p) ? 1 : 0;
}

// ... of a transposed k x k stride-k layer as one scattering GEMM
int wdg_patch_h16_eligible_s(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    if (g.kh != g.kw || g.stride != g.kh || g.kh < 2 || g.pad_h || g.pad_w || g.Cin % 4 || g.H != g.Ho * g.kh || g.W != g.Wo * g.kw) return 0;
    WdgPatchH16 p;
    memset(&p, 0, sizeof(p));
    return patch_plan(patch_view_shuffle(pl), p) ? 1 : 0;
}

template <int FMT, int MT, int NT, bool NLOOP, int DBG = 0, int LSTM = 0>
static int patch_launch(const WdgPatchH16& p, int blocks, size_t lds, hipStream_t st) {
    static size_t lds_set = 0;
    if (lds > lds_set) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_conv_patch_h16_kernel<FMT, MT, NT, NLOOP, DBG, LSTM>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL((wdg_conv_patch_h16_kernel<FMT, MT, NT, NLOOP, DBG, LSTM>), dim3(blocks), dim3(256), lds, st, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// returns WDG_OK after launching, or 1 when the geometry is not eligible (caller falls back to the gather kernel).
// transposed1x1: x is dy, y is dx and w16 the data-gradient packing [Cin][Cout_p] of a 1 x 1, stride-1 plan.
int wdg_patch_h16_launch(const wdg_conv_plan* pl, int transposed1x1, const float* x, const void* w16, const float* bias,
                         const float* affine, float* y, int act, float slope, int accumulate, int fmt, hipStream_t st,
                         const WdgPatchGates* gx, int out16, int in16) {
    if (out16 && (gx || accumulate)) return 1;
    const bool shuffle = transposed1x1 == 2;
    if (transposed1x1 == 1 && (pl->g.kh != 1 || pl->g.kw != 1 || pl->g.stride != 1 || pl->g.pad_h || pl->g.pad_w)) return 1;
    if (shuffle && (pl->g.kh != pl->g.kw || pl->g.stride != pl->g.kh || pl->g.kh < 2 || pl->g.pad_h || pl->g.pad_w || pl->g.Cin % 4 ||
                    pl->g.H != pl->g.Ho * pl->g.kh || pl->g.W != pl->g.Wo * pl->g.kw || gx))
        return 1;
    const WdgPatchView g = shuffle ? patch_view_shuffle(pl) : patch_view(pl, transposed1x1 != 0);
    WdgPatchH16 p;
    memset(&p, 0, sizeof(p));
    if (!patch_plan(g, p, g_patch_lstm_small && gx && gx->c_out)) return 1;
    if (out16 && (g.Ncols % 16 || g.ldO % 8)) return 1;              // whole 16-column tiles in pairs of lanes, 16-byte stores
    if (out16 && shuffle && pl->g.Cin % 32) return 1;                // ... a pair inside one tap
    if (in16 && g.ldA % 8) return 1;
    p.A = x; p.B = w16; p.Out = y; p.bias = bias; p.affine = affine; p.out16 = out16; p.in16 = in16;
    p.imgStrideA = g.imgStrideA; p.imgStrideO = g.imgStrideO;
    p.H = g.H; p.W = g.W; p.ldA = g.ldA; p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldO;
    p.Ncols = g.Ncols; p.ldB = g.kh * g.kw * g.K_p; p.Cin_p = g.K_p;
    p.kh = g.kh; p.kw = g.kw; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
    p.act = act; p.slope = slope; p.accumulate = accumulate;
    if (shuffle) { p.shufC = pl->g.Cin; p.shufK = pl->g.kh; }
    const bool lstm = gx && gx->c_out;
    if (gx) {
        // ConvLSTM gate columns interleaved (x-part: plain epilogue, columns written interleaved; step: cell update in the epilogue)
        if (gx->F <= 0 || g.Ncols != 4 * gx->F || transposed1x1) return 1;
        p.gate_F = gx->F;
        if (lstm) {
            p.gates_x = gx->gates_x; p.c_prev = gx->c_prev; p.c_out = gx->c_out; p.ldc = gx->ldc;
            p.Out = gx->h_out; p.ldO = gx->ldh; p.imgStrideO = (long long)g.Ho * g.Wo * gx->ldh;
            p.h16_out = gx->h16_out; p.ldh16 = gx->ldh16;
            if (!gx->h_out && !gx->h16_out) return 1;
            if (gx->skip_k) p.nchunk = 0;            // t = 0: h_{-1} = 0, the recurrent convolution contributes nothing
        }
    }
    if ((long long)g.H * g.W * g.ldA * 4 >= (1LL << 31) || (long long)g.Ncols * p.ldB * 2 >= (1LL << 31)) return 1;
    const int MT = p.mt;
    // 64-channel tiles when the map is small (the per-timestep recurrent convolution) or the layer is narrow
    const long long tiles_px = (long long)g.n_img * p.tiles_x * p.tiles_y;
    const bool narrow = g.Ncols <= 64 || tiles_px * ((g.Ncols + 127) / 128) < (long long)g.cus * 3 / 2;
    const bool h16_state = lstm && (gx->h16_out || !gx->h_out);       // (written by the transposed epilogue of the 128-column tile only)
    const int BN = h16_state ? 128 : (lstm && g_patch_lstm_bn) ? g_patch_lstm_bn : narrow ? 64 : 128;
    p.tiles_n = (g.Ncols + BN - 1) / BN;
    if (lstm) {
        const bool al16 = !((uintptr_t)gx->c_out & 15) && !((uintptr_t)gx->c_prev & 15) && !((uintptr_t)gx->h_out & 15) && !((uintptr_t)gx->h16_out & 7) &&
                          !((uintptr_t)gx->gates_x & 15);
        p.lstm_vec = BN == 128 && g.Ncols % BN == 0 && gx->ldc % 4 == 0 && (!gx->h_out || gx->ldh % 4 == 0) && (!gx->h16_out || gx->ldh16 % 4 == 0) && al16;
        if (!p.lstm_vec && (gx->h16_out || !gx->h_out)) return 1;        // (the 16-bit copy of h is written by the transposed epilogue only)
    }
    p.ntn_blk = 1;
    if (g_patch_nloop && !lstm && p.nchunk == 1 && p.tiles_n > 1 && tiles_px >= 4LL * g.cus) { p.ntn_blk = p.tiles_n; p.tiles_n = 1; }
    p.div_tn = wdg_fastdiv_make((unsigned)p.tiles_n);
    p.div_tx = wdg_fastdiv_make((unsigned)p.tiles_x);
    p.div_ty = wdg_fastdiv_make((unsigned)p.tiles_y);
    p.div_ck = wdg_fastdiv_make((unsigned)p.CK8);
    p.div_kw = wdg_fastdiv_make((unsigned)g.kw);
    p.div_kcn = wdg_fastdiv_make((unsigned)p.kcn);
    p.nent = g.kh * g.kw * p.CK8;
    p.flat = g_patch_flat && p.nchunk == 1 && (p.CK8 & 3) != 0 && p.Cin_p == p.CK8 * 8;
    p.div_pw = wdg_fastdiv_make((unsigned)p.PW);
    const long long blocks = tiles_px * p.tiles_n;
    if (blocks <= 0 || blocks >= (1LL << 31)) return 1;
    {
        const int depth = lstm ? 3 : 2;                                   // WDG_PATCH_DEPTH of the instantiation launched below
        const int nks = p.flat ? (p.nent + 3) / 4 : g.kh * g.kw * p.kcn;  // K-steps per chunk
        const int nstage = (nks + 1) / 2;
        p.npad = (nstage + depth - 1) / depth * depth;
        p.ntab = (p.npad + depth + 1) * 8;                                // the pipeline requests stages up to npad + depth
    }
    const size_t lds = (size_t)2 * 8 * BN * 16 + ((size_t)p.CK8 * p.pitch + 1) * 16 + (size_t)2 * p.ntab * 4 + (size_t)3 * BN * 4;   // weight stages + patch + K-step tables + epilogue constants
    if (lds > 160 * 1024) return 1;
#ifdef WDG_PATCH_EXPERIMENTS
    // timing experiments (wrong results by design): bf16, 1 x 16 fragments, 128 channels per tile only
#define WDG_PATCH_DBG_CASE(D) if (g_patch_dbg == D && fmt == 0 && MT == 4 && BN == 128) return patch_launch<0, 4, 4, false, D>(p, (int)blocks, lds, st)
    WDG_PATCH_DBG_CASE(1); WDG_PATCH_DBG_CASE(2); WDG_PATCH_DBG_CASE(4); WDG_PATCH_DBG_CASE(8); WDG_PATCH_DBG_CASE(12);
    WDG_PATCH_DBG_CASE(16); WDG_PATCH_DBG_CASE(32); WDG_PATCH_DBG_CASE(3); WDG_PATCH_DBG_CASE(15); WDG_PATCH_DBG_CASE(128); WDG_PATCH_DBG_CASE(131);
#undef WDG_PATCH_DBG_CASE
    // ... and of the recurrent step (bf16, 4 x 24 tiles, 128 channels per tile); bit 6: no gate / cell-state loads in the epilogue
#define WDG_PATCH_DBG_LSTM(D) if (lstm && g_patch_dbg == D && fmt == 0 && MT == 3 && BN == 128) return patch_launch<0, 3, 4, false, D, 1>(p, (int)blocks, lds, st)
    WDG_PATCH_DBG_LSTM(1); WDG_PATCH_DBG_LSTM(2); WDG_PATCH_DBG_LSTM(4); WDG_PATCH_DBG_LSTM(16); WDG_PATCH_DBG_LSTM(32); WDG_PATCH_DBG_LSTM(64);
    WDG_PATCH_DBG_LSTM(3); WDG_PATCH_DBG_LSTM(80); WDG_PATCH_DBG_LSTM(83); WDG_PATCH_DBG_LSTM(87); WDG_PATCH_DBG_LSTM(119);
#undef WDG_PATCH_DBG_LSTM
#endif
#define WDG_PATCH_CASE(F, M, N)                                                                        \
    if (fmt == F && MT == M && BN == 32 * N)                                                           \
        return lstm ? patch_launch<F, M, N, false, 0, 1>(p, (int)blocks, lds, st)                     \
                    : p.ntn_blk > 1 ? patch_launch<F, M, N, true>(p, (int)blocks, lds, st) : patch_launch<F, M, N, false>(p, (int)blocks, lds, st)
    WDG_PATCH_CASE(0, 4, 4); WDG_PATCH_CASE(0, 4, 2); WDG_PATCH_CASE(0, 6, 4); WDG_PATCH_CASE(0, 6, 2); WDG_PATCH_CASE(0, 3, 4); WDG_PATCH_CASE(0, 3, 2);
    WDG_PATCH_CASE(1, 4, 4); WDG_PATCH_CASE(1, 4, 2); WDG_PATCH_CASE(1, 6, 4); WDG_PATCH_CASE(1, 6, 2); WDG_PATCH_CASE(1, 3, 4); WDG_PATCH_CASE(1, 3, 2);
#undef WDG_PATCH_CASE
    return 1;
}
