// conv_igemm.hip — im2col-free implicit-GEMM convolution on v_mfma_f32_16x16x4_f32 (gfx950).
//
// One kernel template serves the forward convolution and (phase-decomposed) data gradient /
// transposed convolution; a second one serves the weight gradient.  Replaces the TF conv kernels
// behind every kl.Conv2D / kl.Conv2DTranspose / kl.ConvLSTM2D call of
// /root/reference/src/downscaling/gan/models.py:33-140.
//
// GEMM view:   Out[m][n] = sum_k A[m][k] * B[n][k]
//   m : output pixel of one phase (img, pa, pb)           -> rows
//   n : output channel                                    -> cols
//   k : (tap, input-channel) flattened in groups of 4 channels ("k4 groups"); a per-plan device
//       table gives, for every k4 group, the A element offset relative to the pixel base, the tap
//       displacement (for the zero-padding test) and the B element offset.
//
// LDS tile layout: [kg][row] float4 slots, slot = kg*ROWS + (row ^ kg)  (kg = 0..7).
//   * a lane's float4 holds 4 consecutive k of one row, so one ds_read_b128 feeds 4 MFMAs;
//   * staging writes (8 consecutive lanes = 8 kg of one row = 128 contiguous global bytes) and
//     fragment reads (16 rows x 4 kg per wave instruction) are both bank-conflict free under the
//     XOR (guide: cdna_hip_programming.md section 2 / T2).
// MFMA operand map (16x16x4 f32): lane l supplies A[row l&15][k l>>4], B[k l>>4][col l&15];
// accumulator reg r of lane l is C[row (l>>4)*4 + r][col l&15].
#include "conv_plan.h"
#include <algorithm>

// Timing-only skeletons (measurement builds: -DWDG_KLOOP_EXP=<bits>, tools/build_exp.sh; the shipped library is built without it
// and the results of such a build are WRONG by construction).  bit 0: operand loads replaced by constants; bit 1: no LDS staging
// stores; bit 2: fragments not re-read from LDS inside the K loop (read once before it); bit 3: no barriers in the K loop.
#ifndef WDG_KLOOP_EXP
#define WDG_KLOOP_EXP 0
#endif

struct WdgIgemm {
    const float* A;
    const float* B;
    float* Out;
    const float* bias;
    const int4* ktab;  // {a_off, dh, dw, b_off}; b_off < 0 marks a padding entry
    float* partial;    // split-K slabs [phase][split][Mmax][Ncols]
    long long imgStrideA, imgStrideO;
    int n_img, H, W, ldA;  // A tensor
    int Ho, Wo, ldO;       // Out tensor
    int Ncols, ldB;
    int a_mul, o_mul;
    int act, accumulate;
    float slope;
    int splitk, k4_per_split;
    int Mmax, nphase;
    int xcd_swizzle;   // remap blockIdx.x so that each XCD (blocks b, b+8, ...) walks a contiguous range of tiles
    int n_fastest;     // 1: consecutive blocks walk the column tiles of ONE row tile (they share the A rows in L2; the whole B
                       // operand is small enough to stay resident); 0: consecutive blocks walk row tiles (share the B tile)
    int phase_in_x;    // > 0: blockIdx.x = tile * nphase + phase (the phases of one output tile run back to back on ONE XCD and
                       // share its L2 copy of the input rows); 0: phase = blockIdx.z
    // fused BatchNorm hooks of the epilogue (EPI template flag): per-channel sum / sum of squares of the written values
    // into one of `stats_rep` replica slabs [2][stats_C] (fp64 atomics; replicas spread the contention), or the
    // inference-mode normalisation  v * affine[n] + affine[Ncols_pad + n]  after the activation
    double* stats;
    int stats_C, stats_rep;
    const float* affine;
    int affine_ld;
    // fused LayerNormalization of the output rows (EPI 3; the split-K second stage has its own form): y = act(conv + bias)
    // goes to Out, z = (y - mean) * rstd * gamma + beta to Out2 (same view), (mean, rstd) per pixel to mean_rstd
    float* Out2;
    const float* ln_gamma;
    const float* ln_beta;
    float* mean_rstd;
    float ln_eps;
    // ConvLSTM2D recurrent step (EPI 4; models.py:45 at n_timesteps > 1): the B rows are gate-interleaved (column n = gate n & 3
    // of feature n >> 2), so a lane's four accumulator registers are i, f, c~, o of one feature of one pixel; Out is the gates slab
    // [pixel][4 F] in the standard order (it holds the input part, receives the pre-activation sums the backward pass reads),
    // the cell update runs on the accumulators and writes c_out / h_out
    int lstm_F, ldc, ldh;
    const float* c_prev;
    float* c_out;
    float* h_out;
    // LayerNormalization BACKWARD on the data gradient being written (EPI 5; models.py:97,105,116,125 differentiated): the output
    // columns [lnb_c0, lnb_c0 + lnb_C) are the gradient dz w.r.t. the OUTPUT of a LayerNormalization whose input was y = LeakyReLU(conv
    // + bias).  The epilogue holds dz for all channels of a pixel, so it writes dpre = rstd * (dz g - mean_c(dz g) - xh mean_c(dz g xh))
    // * lrelu'(y) in their place (what the standalone wdg_ln_bwd pass produced from a re-read of dz) and, when lnb_par is given,
    // accumulates the block's share of dgamma = sum dz xh, dbeta = sum dz, dbias = sum dpre into one of lnb_rep replica slabs
    // [3][lnb_C] (float atomics; wdg_ln_param_finish_kernel sums the replicas into the gradient vectors and clears them).
    const float* lnb_y;
    const float* lnb_stats;
    const float* lnb_gamma;
    float* lnb_par;
    long long lnb_imgStride;
    int lnb_ldy, lnb_c0, lnb_C, lnb_rep;
    float lnb_slope;
    WdgPhase ph[9];
};

// row index of a phase -> (image, pa, pb): linear order, or 2-D row tiles (WdgPhase::t2_w: rows are "virtual", tile * BM + row
// in tile; rows >= t2_rows of a tile are idle — wdg_row_valid)
template <int BM>
__device__ __forceinline__ void wdg_row_to_pixel(const WdgPhase& ph, int PaPb, int m, int& img, int& pa, int& pb) {
    if (ph.t2_w) {
        const int tile = m / BM, ml = m - tile * BM;
        img = (int)wdg_fastdiv_do((unsigned)tile, ph.div_t2_img);
        const int tr = tile - img * ph.t2_tpi;
        const int ty = (int)wdg_fastdiv_do((unsigned)tr, ph.div_t2_w);
        const int tx = tr - ty * ph.t2_tiles_w;
        const int ly = (int)wdg_fastdiv_do((unsigned)ml, ph.div_t2_ml);
        pa = ty * ph.t2_h + ly;
        pb = tx * ph.t2_w + (ml - ly * ph.t2_w);
    } else {
        img = (int)wdg_fastdiv_do((unsigned)m, ph.div_papb);
        const int rem = m - img * PaPb;
        pa = (int)wdg_fastdiv_do((unsigned)rem, ph.div_pb);
        pb = rem - pa * ph.Pb;
    }
}

template <int BM>
__device__ __forceinline__ bool wdg_row_valid(const WdgPhase& ph, int Mph, int m) {
    return m < Mph && (!ph.t2_w || (m & (BM - 1)) < ph.t2_rows);
}

// EPI: 0 plain epilogue, 1 = + BatchNorm batch statistics of the output (training-mode producer), 2 = + inference-mode
// BatchNorm affine.  Separate instantiations: the plain kernels keep their register allocation.
// KG = 2: the reduction is split over two groups of four waves INSIDE the workgroup (512 threads; each group runs the K loop
// on its half with its own LDS stage, the second hands its accumulators to the first through LDS, the first owns the epilogue).
// For launches with fewer tiles than CUs and a deep reduction whose epilogue needs the complete sums (the generator's recurrent
// step at batch 8: 144 tiles of 128 x 128, K = 1152): twice the waves on the tile's MFMAs, no workspace, no second kernel.
// (256 x 32 tile, default loop: 132 registers sat four above the four-waves-per-SIMD line — the bound makes the compiler fit 128)
// measurement builds: -DWDG_EARLY_LOADS=1 pins the next tile's operand requests in front of the K-step's MFMAs on the narrow tiles
// (BN <= 64) and relaxes the 256 x 32 tile's four-waves-per-SIMD bound (the requests' 24-40 destination registers are then live
// across the MFMA phase)
#ifndef WDG_EARLY_LOADS
#define WDG_EARLY_LOADS 0
#endif
// -DWDG_MFMA_PRIO=1: s_setprio(1) / (0) around every cluster of MFMAs of the K loops (cdna_hip_programming.md T5: a compiler effect —
// the cluster stays between the barriers it was written between)
#ifndef WDG_MFMA_PRIO
#define WDG_MFMA_PRIO 0
#endif
template <int BM, int BN, int WGM, int WGN, int PIPE, int EPI = 0, int KG = 1>
__global__ void __launch_bounds__(256 * KG, (BM == 256 && BN == 32 && PIPE == 3 && KG == 1 && !WDG_EARLY_LOADS) ? 4 : 1) wdg_igemm_kernel(const WdgIgemm p) {
    static_assert(KG == 1 || (KG == 2 && PIPE == 3 && (EPI == 0 || EPI == 4)), "in-workgroup split: rotated loop, barrier-free epilogues");
    static_assert(PIPE != 5 || (BN % 16 == 0 && BM % 32 == 0), "LDS-DMA pieces are 8 whole rows");
    constexpr int MT = BM / WGM / 16;
    constexpr int NT = BN / WGN / 16;
    constexpr int A_LOADS = BM / 32;
    constexpr int B_LOADS = (BN + 31) / 32;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(MT >= 1 && NT >= 1, "tile");

    // one LDS array (dynamic): [stage][A tile | B tile]; PIPE >= 1 uses two stages
    extern __shared__ __attribute__((aligned(16))) f32x4 lds_raw[];
    constexpr int STAGE = 8 * (BM + BN);
    const int kgrp = KG > 1 ? (int)(threadIdx.x >> 8) : 0;
    f32x4* const lds_all = lds_raw + (KG > 1 ? kgrp * STAGE : 0);

    const int t = threadIdx.x & 255;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int kg = t & 7;
    const int lrow = t >> 3;  // 0..31

    int bid = p.xcd_swizzle ? wdg_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    int phase_id = blockIdx.z;
    if (p.phase_in_x) {
        phase_id = bid % p.phase_in_x;
        bid /= p.phase_in_x;
    }
    const WdgPhase ph = p.ph[phase_id];
    const int tiles_m = (p.Mmax + BM - 1) / BM;
    int tm, tn;
    if (p.n_fastest) {
        const int tiles_n = (p.Ncols + BN - 1) / BN;
        tn = bid % tiles_n;
        tm = bid / tiles_n;
    } else {
        tm = bid % tiles_m;
        tn = bid / tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int PaPb = ph.Pa * ph.Pb;
    const int Mph = ph.t2_w ? p.n_img * ph.t2_tpi * BM : p.n_img * PaPb;     // (2-D tiles: virtual rows)
    if (m0 >= Mph) return;

    const int k4_begin = (KG > 1 ? kgrp : (int)blockIdx.y) * p.k4_per_split;
    int k4_end = k4_begin + p.k4_per_split;
    if (k4_end > ph.K4) k4_end = ph.K4;
    const int nk = k4_end > k4_begin ? (k4_end - k4_begin) >> 3 : 0;

    // ---- buffer descriptors (wave-uniform: kernel arguments and blockIdx only).  Every operand load is a
    // buffer_load_dwordx4 whose byte offset is pushed out of range for padding / out-of-image / tail lanes,
    // so the hardware range check returns the zeros and the load sequence has no branches.
    const int img0 = ph.t2_w ? (int)wdg_fastdiv_do((unsigned)(m0 / BM), ph.div_t2_img) : (int)wdg_fastdiv_do((unsigned)m0, ph.div_papb);
    const wdg_srd srdA = wdg_make_srd(p.A + (long long)img0 * p.imgStrideA);
    const wdg_srd srdB = wdg_make_srd(p.B);

    // ---- per-thread A row state (element offsets relative to image img0)
    int a_off[A_LOADS];
    int a_ih0[A_LOADS], a_iw0[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (wdg_row_valid<BM>(ph, Mph, m)) {
            int img, pa, pb;
            wdg_row_to_pixel<BM>(ph, PaPb, m, img, pa, pb);
            const int ih0 = pa * p.a_mul + ph.a_off_h;
            const int iw0 = pb * p.a_mul + ph.a_off_w;
            a_ih0[i] = ih0;
            a_iw0[i] = iw0;
            a_off[i] = (int)((long long)(img - img0) * p.imgStrideA) + (ih0 * p.W + iw0) * p.ldA;
        } else {
            a_ih0[i] = -(1 << 28);
            a_iw0[i] = -(1 << 28);
            a_off[i] = 0;
        }
    }
    // ---- per-thread B row state
    int b_off[B_LOADS];
    bool b_ok[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int nl = lrow + 32 * i;
        const int n = n0 + nl;
        b_ok[i] = (nl < BN) && (n < p.Ncols);
        b_off[i] = n * p.ldB;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[A_LOADS], rb[B_LOADS];
    const int4* tab = p.ktab + ph.tab_off + k4_begin + kg;

    // the table entry of the NEXT tile is fetched together with this tile's operands, so no load_tile ever
    // waits on its own table read
    int4 e_cur = nk > 0 ? tab[0] : (int4){0, 0, 0, -1};
    auto load_tile = [&](int kt) {
        const int4 e = e_cur;
        if constexpr (WDG_KLOOP_EXP & 1) {
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) ra[i] = (f32x4){(float)(e.x + i), 1.f, (float)kt, 2.f};
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) rb[i] = (f32x4){(float)(e.w + i), 1.f, (float)kt, 2.f};
            e_cur = tab[(kt + 1 < nk ? kt + 1 : kt) * 8];
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int ih = a_ih0[i] + e.y, iw = a_iw0[i] + e.z;
            const bool ok = ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && (e.w >= 0);
            ra[i] = wdg_buffer_load_f32x4(srdA, ok ? (unsigned)(a_off[i] + e.x) << 2 : WDG_SRD_OOB);
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const bool ok = b_ok[i] && (e.w >= 0);
            rb[i] = wdg_buffer_load_f32x4(srdB, ok ? (unsigned)(b_off[i] + e.w) << 2 : WDG_SRD_OOB);
        }
        e_cur = tab[(kt + 1 < nk ? kt + 1 : kt) * 8];
    };

    auto store_tile = [&](f32x4* ldsA, f32x4* ldsB) {
        if constexpr (WDG_KLOOP_EXP & 2) {
            asm volatile("" :: "v"(ra[0][0]), "v"(rb[0][0]));
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = lrow + 32 * i;
            ldsA[kg * BM + (row ^ kg)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int row = lrow + 32 * i;
            if (32 * (i + 1) <= BN || row < BN) ldsB[kg * BN + (row ^ kg)] = rb[i];
        }
    };
    auto read_frags = [&](const f32x4* ldsA, const f32x4* ldsB, int h, f32x4 (&af)[MT], f32x4 (&bf)[NT]) {
        const int kgr = 4 * h + (lane >> 4);
#pragma unroll
        for (int a = 0; a < MT; ++a) af[a] = ldsA[kgr * BM + ((wm * (BM / WGM) + a * 16 + (lane & 15)) ^ kgr)];
#pragma unroll
        for (int b = 0; b < NT; ++b) bf[b] = ldsB[kgr * BN + ((wn * (BN / WGN) + b * 16 + (lane & 15)) ^ kgr)];
    };
    auto mfma_block = [&](const f32x4 (&af)[MT], const f32x4 (&bf)[NT]) {
        if constexpr (WDG_MFMA_PRIO != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    // A = weights, B = pixels: the accumulator holds the TRANSPOSED tile — reg r of lane (i = lane & 15,
                    // q = lane >> 4) is output channel 4q + r of pixel i — so the epilogue writes 16 bytes per lane
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][j], af[a][j], acc[a][b], 0, 0, 0);
        if constexpr (WDG_MFMA_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (PIPE == 5) {
        // ---- operands travel global -> LDS directly (LDS-DMA: buffer_load_dwordx4 ... lds), two LDS stages, ONE barrier per K-step.
        // No staging registers (the rotated loop keeps 24-40 of them live between the request and the ds_write, which is why the
        // compiler sinks the requests behind the K-step's MFMAs), no ds_write instructions, and the requests of tile k + 1 are issued
        // at the top of step k: a whole MFMA phase ahead of the barrier that publishes them.
        // The DMA's LDS destination is lane-linear (wave base + 16 * lane), so the stage is pixel-major, slot(row, pos) = 8 * row + pos,
        // and the bank swizzle sits on the SOURCE side: slot (row, pos) holds the row's k4 group  pos ^ ((row >> 1) & 7).  A staging
        // thread (row = t >> 3, pos = t & 7) therefore owns the k4 group kgl = (t & 7) ^ ((t >> 4) & 7) — for every one of its rows,
        // 32 apart —, the 8 lanes of a row request a permutation of the row's 128 contiguous bytes (full lines), and a fragment
        // read (16 consecutive rows of one k4 group) touches 16 slots whose (row & 1, pos) pairs are all different: 64 banks.
        // Padding / tail lanes: offset beyond the descriptor -> the DMA writes zeros (probed: tools/probes/lds_dma_oob.hip).
        const int kgl = (t & 7) ^ ((t >> 4) & 7);
        const int4* tab5 = p.ktab + ph.tab_off + k4_begin + kgl;
        const int wv_u = __builtin_amdgcn_readfirstlane(wave);
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        int4 e5 = nk > 0 ? tab5[0] : (int4){0, 0, 0, -1};
        auto issue = [&](int kt, int stage) {
            const int4 e = e5;
            f32x4* const sA = lds_all + stage * STAGE;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int ih = a_ih0[i] + e.y, iw = a_iw0[i] + e.z;
                const bool ok = ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W) && (e.w >= 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (lds_ptr_t)(sA + 64 * wv_u + 256 * i), 16,
                                                         ok ? (int)((unsigned)(a_off[i] + e.x) << 2) : (int)WDG_SRD_OOB, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) {
                if (8 * wv_u + 32 * i < BN) {           // (wave-uniform: a wave's piece is 8 rows, BN is a multiple of 16)
                    const bool ok = b_ok[i] && (e.w >= 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (lds_ptr_t)(sA + 8 * BM + 64 * wv_u + 256 * i), 16,
                                                             ok ? (int)((unsigned)(b_off[i] + e.w) << 2) : (int)WDG_SRD_OOB, 0, 0, 0);
                }
            }
            e5 = tab5[(kt + 1 < nk ? kt + 1 : kt) * 8];
        };
        auto read_frags5 = [&](const f32x4* sA, int h, f32x4 (&af)[MT], f32x4 (&bf)[NT]) {
            const int kgr = 4 * h + (lane >> 4);
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int row = wm * (BM / WGM) + a * 16 + (lane & 15);
                af[a] = sA[8 * row + (kgr ^ ((row >> 1) & 7))];
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int row = wn * (BN / WGN) + b * 16 + (lane & 15);
                bf[b] = sA[8 * BM + 8 * row + (kgr ^ ((row >> 1) & 7))];
            }
        };
        if (nk > 0) issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                      // s_waitcnt vmcnt(0) + barrier: every wave's pieces of tile kt have landed, and
                                                  // every wave has finished reading the other stage (its reads of step kt - 1)
            if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
            const f32x4* sA = lds_all + (kt & 1) * STAGE;
            {
                f32x4 af[MT], bf[NT];
                read_frags5(sA, 0, af, bf);
                mfma_block(af, bf);
            }
            {
                f32x4 af[MT], bf[NT];
                read_frags5(sA, 1, af, bf);
                mfma_block(af, bf);
            }
        }
        __syncthreads();                          // (epilogues reuse the stages as scratch)
    } else if (PIPE == 0) {
        f32x4* ldsA = lds_all;
        f32x4* ldsB = lds_all + 8 * BM;
        if (nk > 0) load_tile(0);
        for (int kt = 0; kt < nk; ++kt) {
            store_tile(ldsA, ldsB);
            __syncthreads();
            if (kt + 1 < nk) load_tile(kt + 1);  // in flight under the MFMAs below
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 af[MT], bf[NT];
                read_frags(ldsA, ldsB, h, af, bf);
                mfma_block(af, bf);
            }
            __syncthreads();
        }
    } else if (PIPE == 3) {
        // rotated single-stage loop, ONE basic block per K-step: the operand loads of the next tile (address
        // arithmetic + 8 buffer loads) and its LDS stores sit in the same block as the MFMAs, so the scheduler can
        // interleave them with the matrix instructions instead of running them as separate phases
        f32x4* ldsA = lds_all;
        f32x4* ldsB = lds_all + 8 * BM;
        if (nk > 0) {
            load_tile(0);
            store_tile(ldsA, ldsB);
        }
        __syncthreads();
        if constexpr ((WDG_KLOOP_EXP & 4) != 0) {
            f32x4 af[MT], bf[NT];
            read_frags(ldsA, ldsB, 0, af, bf);
            for (int kt = 0; kt < nk; ++kt) {
                load_tile(kt + 1 < nk ? kt + 1 : kt);
                mfma_block(af, bf);
                if constexpr (!(WDG_KLOOP_EXP & 8)) __syncthreads();
                mfma_block(af, bf);
                store_tile(ldsA, ldsB);
                if constexpr (!(WDG_KLOOP_EXP & 8)) __syncthreads();
#pragma unroll
                for (int a = 0; a < MT; ++a) asm volatile("" : "+v"(af[a]));
            }
        } else
        for (int kt = 0; kt < nk; ++kt) {
            load_tile(kt + 1 < nk ? kt + 1 : kt);   // unconditional (the last one is redundant)
            if constexpr (WDG_EARLY_LOADS != 0 && BN <= 64) __builtin_amdgcn_sched_barrier(0);
            {
                f32x4 af[MT], bf[NT];
                read_frags(ldsA, ldsB, 0, af, bf);
                mfma_block(af, bf);
            }
            {
                f32x4 af[MT], bf[NT];
                read_frags(ldsA, ldsB, 1, af, bf);
                if constexpr (!(WDG_KLOOP_EXP & 8)) __syncthreads();   // every wave has read this tile: the stage may be overwritten
                mfma_block(af, bf);
            }
            store_tile(ldsA, ldsB);
            if constexpr (!(WDG_KLOOP_EXP & 8)) __syncthreads();
        }
    } else {
        // double-buffered LDS: the next tile is written into the other stage right after this wave's own
        // MFMAs, one barrier per K-step
        if (nk > 0) {
            load_tile(0);
            store_tile(lds_all, lds_all + 8 * BM);
            __syncthreads();
            if (nk > 1) load_tile(1);
        }
        for (int kt = 0; kt < nk; ++kt) {
            const f32x4* ldsA = lds_all + (kt & 1) * STAGE;
            const f32x4* ldsB = ldsA + 8 * BM;
            if (PIPE == 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 af[MT], bf[NT];
                    read_frags(ldsA, ldsB, h, af, bf);
                    mfma_block(af, bf);
                }
            } else {
                f32x4 af0[MT], bf0[NT], af1[MT], bf1[NT];
                read_frags(ldsA, ldsB, 0, af0, bf0);
                read_frags(ldsA, ldsB, 1, af1, bf1);   // second half's fragments land under the first half's MFMAs
                mfma_block(af0, bf0);
                mfma_block(af1, bf1);
            }
            if (kt + 1 < nk) {
                f32x4* nA = lds_all + ((kt + 1) & 1) * STAGE;
                store_tile(nA, nA + 8 * BM);
            }
            __syncthreads();
            if (kt + 2 < nk) load_tile(kt + 2);
        }
    }

    if constexpr (KG > 1) {
        // the second group's partial sums -> LDS (both stages are free now; slot = the same (tile, thread) in both groups)
        __syncthreads();
        if (kgrp == 1) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) lds_raw[(a * NT + b) * 256 + t] = acc[a][b];
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] += lds_raw[(a * NT + b) * 256 + t];
    }

    // ---- epilogue: one pixel per (row tile a), four consecutive output channels per (column tile b)
    const int q4 = 4 * (lane >> 4);
    const int NcP = (p.Ncols + 3) & ~3;
    if constexpr (EPI == 3) {
        // conv -> bias -> LeakyReLU -> LayerNormalization over the channels, the block owning complete rows (tiles_n == 1,
        // no split-K: checked by the host).  A row's channels sit in 4 * WGN lanes (the four lane >> 4 groups of WGN
        // waves): its two reductions (sum, centred sum of squares) go through LDS, the K loop's stage being free.
        // WGN == 1 (the 128 x 64 tile of the discriminator's first strided layer runs this epilogue as four waves of 32 rows x all
        // 64 columns): a row's channels sit in the four lane >> 4 groups of ONE wave, so both reductions are two xor-shuffles —
        // no LDS round trip, no barrier (the LDS form below cost that layer 56 us of its 431: 4 barriers + 16 partial reads)
        float* red = reinterpret_cast<float*>(lds_all);                    // [BM][4 * WGN]
        constexpr int RW = 4 * WGN;
        const int rsub = wn * 4 + (lane >> 4);
        float mean[MT], rstd[MT];
        const float invC = 1.f / (float)p.Ncols;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            float s_ = 0.f;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + wn * (BN / WGN) + b * 16 + q4;
                f32x4 v = acc[a][b];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (p.bias) v[r] += n + r < p.Ncols ? p.bias[n + r] : 0.f;
                    if (p.act) v[r] = wdg_lrelu(v[r], p.slope);
                    s_ += n + r < p.Ncols ? v[r] : 0.f;
                }
                acc[a][b] = v;
            }
            if constexpr (WGN == 1) {
                s_ += __shfl_xor(s_, 16, 64);
                s_ += __shfl_xor(s_, 32, 64);
                mean[a] = s_ * invC;
            } else {
                red[(wm * (BM / WGM) + a * 16 + (lane & 15)) * RW + rsub] = s_;
            }
        }
        if constexpr (WGN > 1) {
            __syncthreads();
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                float s_ = 0.f;
#pragma unroll
                for (int k = 0; k < RW; ++k) s_ += red[(wm * (BM / WGM) + a * 16 + (lane & 15)) * RW + k];
                mean[a] = s_ * invC;
            }
            __syncthreads();
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            float q_ = 0.f;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + wn * (BN / WGN) + b * 16 + q4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[a][b][r] - mean[a];
                    q_ += n + r < p.Ncols ? d * d : 0.f;
                }
            }
            if constexpr (WGN == 1) {
                q_ += __shfl_xor(q_, 16, 64);
                q_ += __shfl_xor(q_, 32, 64);
                rstd[a] = 1.f / sqrtf(q_ * invC + p.ln_eps);
            } else {
                red[(wm * (BM / WGM) + a * 16 + (lane & 15)) * RW + rsub] = q_;
            }
        }
        if constexpr (WGN > 1) {
            __syncthreads();
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                float q_ = 0.f;
#pragma unroll
                for (int k = 0; k < RW; ++k) q_ += red[(wm * (BM / WGM) + a * 16 + (lane & 15)) * RW + k];
                rstd[a] = 1.f / sqrtf(q_ * invC + p.ln_eps);
            }
        }
        // stores: pixel offsets once per row, gamma / beta once per column tile (column tile outermost)
        long long off[MT];
        bool rv[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = m0 + wm * (BM / WGM) + a * 16 + (lane & 15);
            rv[a] = wdg_row_valid<BM>(ph, Mph, m);
            off[a] = 0;
            if (!rv[a]) continue;
            int img, pa, pb;
            wdg_row_to_pixel<BM>(ph, PaPb, m, img, pa, pb);
            const int oh = pa * p.o_mul + ph.o_off_h;
            const int ow = pb * p.o_mul + ph.o_off_w;
            off[a] = (long long)img * p.imgStrideO + ((long long)oh * p.Wo + ow) * p.ldO;
            if (p.mean_rstd && rsub == 0) {
                // (forward launches only: one phase, so the output pixel index is img * PaPb + pa * Pb + pb)
                const long long pixi = (long long)img * PaPb + pa * ph.Pb + pb;
                p.mean_rstd[2 * pixi] = mean[a];
                p.mean_rstd[2 * pixi + 1] = rstd[a];
            }
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = n0 + wn * (BN / WGN) + b * 16 + q4;
            if (n >= NcP) continue;
            f32x4 gm, bt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gm[r] = n + r < p.Ncols ? p.ln_gamma[n + r] : 0.f;
                bt[r] = n + r < p.Ncols ? p.ln_beta[n + r] : 0.f;
            }
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                if (!rv[a]) continue;
                const f32x4 v = acc[a][b];
                f32x4 z;
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = (v[r] - mean[a]) * rstd[a] * gm[r] + bt[r];
                *reinterpret_cast<f32x4*>(p.Out + off[a] + n) = v;
                *reinterpret_cast<f32x4*>(p.Out2 + off[a] + n) = z;
            }
        }
        return;
    }
    if constexpr (EPI == 4) {
        // ConvLSTM cell on the accumulators (Keras hard_sigmoid / tanh, gate order i, f, c, o; same arithmetic as wdg_lstm_fwd
        // behind an accumulating convolution).  One phase, stride 1, no split-K (checked by the host).
        const int F = p.lstm_F;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = m0 + wm * (BM / WGM) + a * 16 + (lane & 15);
            if (!wdg_row_valid<BM>(ph, Mph, m)) continue;
            int img, pa, pb;
            wdg_row_to_pixel<BM>(ph, PaPb, m, img, pa, pb);
            const long long pix = (long long)img * PaPb + (long long)pa * ph.Pb + pb;
            float* g = p.Out + pix * (4 * F);
            // every column tile's inputs first (one memory round trip), then the arithmetic and the stores
            float gx[NT][4], cp[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + wn * (BN / WGN) + b * 16 + q4;
                const int f = n >> 2;
                const bool on = n < p.Ncols;
#pragma unroll
                for (int r = 0; r < 4; ++r) gx[b][r] = on ? g[r * F + f] : 0.f;
                cp[b] = on ? p.c_prev[pix * p.ldc + f] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + wn * (BN / WGN) + b * 16 + q4;
                if (n >= p.Ncols) continue;
                const int f = n >> 2;
                float z[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z[r] = acc[a][b][r] + gx[b][r];
                    g[r * F + f] = z[r];
                }
                const float gi = fminf(fmaxf(0.2f * z[0] + 0.5f, 0.f), 1.f);
                const float gf = fminf(fmaxf(0.2f * z[1] + 0.5f, 0.f), 1.f);
                const float gc = wdg_tanh(z[2]);
                const float go = fminf(fmaxf(0.2f * z[3] + 0.5f, 0.f), 1.f);
                const float cn = gi * gc + gf * cp[b];
                p.c_out[pix * p.ldc + f] = cn;
                p.h_out[pix * p.ldh + f] = go * wdg_tanh(cn);
            }
        }
        return;
    }
    if constexpr (EPI == 5 || EPI == 6) {
        // (EPI 6: the same without parameter gradients — the input-gradient-only passes — and without their 12 * NT accumulators)
        constexpr bool PAR = EPI == 5;
        // One wave owns complete rows (WGN == 1: checked at instantiation), a pixel's channels sit in the four lane >> 4 groups
        // of that wave: both reductions of the LayerNorm backward are in-lane sums + two xor-shuffles.  No bias / activation /
        // accumulate / split-K on this route (the host falls back to the two-launch form otherwise).
        static_assert(WGN == 1, "EPI 5 / 6: a row's channels in one wave");
        const int c0 = p.lnb_c0, C = p.lnb_C;
        const float invC = 1.f / (float)C;
        const int HoWo = p.Ho * p.Wo;
        // this lane's share of the parameter gradients, per column tile of the group (summed over its MT pixels)
        float* const red5 = reinterpret_cast<float*>(lds_all);            // [WGM][3][BN] (the K loop has ended behind a barrier: its stage is free)
        // (gamma and y are read once per reduction pass instead of being held across both: 32 registers less on the 64-column
        // tile, whose plain variant runs four waves per SIMD; the second read hits L1)
        auto gamma4 = [&](int b) -> f32x4 {
            const int n = n0 + b * 16 + q4;
            f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (n >= c0 && n < c0 + C) g = *reinterpret_cast<const f32x4*>(p.lnb_gamma + (n - c0));
            return g;
        };
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = m0 + wm * (BM / WGM) + a * 16 + (lane & 15);
            const bool rv = wdg_row_valid<BM>(ph, Mph, m);
            int img = 0, pa = 0, pb_ = 0;
            if (rv) wdg_row_to_pixel<BM>(ph, PaPb, m, img, pa, pb_);
            const int oh = pa * p.o_mul + ph.o_off_h;
            const int ow = pb_ * p.o_mul + ph.o_off_w;
            const long long pix = (long long)oh * p.Wo + ow;
            const long long off = (long long)img * p.imgStrideO + pix * p.ldO;
            const float* yrow = p.lnb_y + (long long)img * p.lnb_imgStride + pix * p.lnb_ldy;
            const float* st = p.lnb_stats + 2 * ((long long)img * HoWo + pix);
            float mean = 0.f, rstd = 0.f;
            if (rv) {
                mean = st[0];
                rstd = st[1];
            }
            auto y4 = [&](int b) -> f32x4 {
                const int n = n0 + b * 16 + q4;
                f32x4 y = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (rv && n >= c0 && n < c0 + C) y = *reinterpret_cast<const f32x4*>(yrow + (n - c0));
                return y;
            };
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const f32x4 yv = y4(b), gm = gamma4(b);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (yv[r] - mean) * rstd;
                    const float gg = acc[a][b][r] * gm[r];       // (gamma = 0 outside the group)
                    s1 += gg;
                    s2 += gg * xh;
                }
            }
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            s1 *= invC;
            s2 *= invC;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + b * 16 + q4;
                if (n >= NcP) continue;
                f32x4 v = acc[a][b];
                if (n >= c0 && n < c0 + C) {
                    const f32x4 yv = y4(b), gm = gamma4(b);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (yv[r] - mean) * rstd;
                        float d = rstd * (v[r] * gm[r] - s1 - xh * s2);
                        if (p.lnb_slope >= 0.f) d *= (yv[r] > 0.f ? 1.f : p.lnb_slope);
                        if constexpr (PAR) {
                            // this pixel's share of the parameter gradients: summed over the 16 pixels of the lane group at
                            // once (DPP row sums) and kept in LDS — 12 * NT live accumulators would cost the 64-column tile a wave
                            const float t0 = wdg_row16_sum(rv ? v[r] * xh : 0.f), t1 = wdg_row16_sum(rv ? v[r] : 0.f),
                                        t2 = wdg_row16_sum(rv ? d : 0.f);
                            if ((lane & 15) == 0) {
                                const int col = b * 16 + q4 + r;
                                float* rr = red5 + wm * 3 * BN + col;
                                rr[0] = (a ? rr[0] : 0.f) + t0;
                                rr[BN] = (a ? rr[BN] : 0.f) + t1;
                                rr[2 * BN] = (a ? rr[2 * BN] : 0.f) + t2;
                            }
                        }
                        v[r] = d;
                    }
                }
                if (rv) *reinterpret_cast<f32x4*>(p.Out + off + n) = v;
            }
        }
        if constexpr (!PAR) return;
        if (!p.lnb_par) return;
        // the WGM row-waves' sums meet here: one atomic per (block, channel, quantity) into a replica slab
        __syncthreads();
        float* slab = p.lnb_par + (size_t)(blockIdx.x % (unsigned)p.lnb_rep) * 3 * C;
        for (int idx = t; idx < 3 * BN; idx += 256) {
            const int which = idx / BN, col = idx - which * BN;
            const int n = n0 + col;
            if (n < c0 || n >= c0 + C) continue;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) v += red5[(w * 3 + which) * BN + col];
            atomicAdd(slab + which * C + (n - c0), v);
        }
        return;
    }
    // Column tile outermost, the MT pixels of a lane innermost: the per-pixel output offsets are formed once (MT values), the
    // bias / affine vectors of a column tile are loaded once, and the BatchNorm statistics need 8 accumulator registers at a
    // time instead of 8 * NT — with all NT tiles' sums live the EPI 1 variant of the 128 x 128 tile took 188 registers (two
    // waves per SIMD) against 168 (three) of the plain one.
    long long doff[MT];
    bool rv[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const int m = m0 + wm * (BM / WGM) + a * 16 + (lane & 15);
        rv[a] = wdg_row_valid<BM>(ph, Mph, m);
        doff[a] = 0;
        if (rv[a]) {
            if (p.splitk > 1) {
                doff[a] = (((long long)phase_id * p.splitk + blockIdx.y) * p.Mmax + m) * NcP;
            } else {
                int img, pa, pb;
                wdg_row_to_pixel<BM>(ph, PaPb, m, img, pa, pb);
                const int oh = pa * p.o_mul + ph.o_off_h;
                const int ow = pb * p.o_mul + ph.o_off_w;
                doff[a] = (long long)img * p.imgStrideO + ((long long)oh * p.Wo + ow) * p.ldO;
            }
        }
    }
    float* red = reinterpret_cast<float*>(lds_all);          // EPI 1: [WGM][BN][2] (the K loop has ended behind a barrier, its stages are free)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + wn * (BN / WGN) + b * 16 + q4;
        if (n >= NcP) continue;
        if (p.splitk > 1) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
                if (rv[a]) *reinterpret_cast<f32x4*>(p.partial + doff[a] + n) = acc[a][b];
            continue;
        }
        f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f}, sc4 = bias4, sh4 = bias4;
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[r] = n + r < p.Ncols ? p.bias[n + r] : 0.f;
        }
        if constexpr (EPI == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc4[r] = n + r < p.Ncols ? p.affine[n + r] : 1.f;
                sh4[r] = n + r < p.Ncols ? p.affine[p.affine_ld + n + r] : 0.f;
            }
        }
        f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            if (!rv[a]) continue;
            float* dst = p.Out + doff[a] + n;
            f32x4 v = acc[a][b] + bias4;
            if (p.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], p.slope);
            }
            if constexpr (EPI == 1) {
                // (pad channels Ncols..NcP-1 carry zeros: zero weights, no bias)
                s1 += v;
#pragma unroll
                for (int r = 0; r < 4; ++r) s2[r] = fmaf(v[r], v[r], s2[r]);
            }
            if constexpr (EPI == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(v[r], sc4[r], sh4[r]);
            }
            if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = v;   // channels Ncols .. round4(Ncols)-1 receive zeros (padding)
        }
        if constexpr (EPI == 1) {
            // per-channel partial sums of this block -> one replica slab.  Lanes that share lane >> 4 hold the same four
            // channels of 16 different pixels: butterfly over lane & 15, then the WGM row-waves meet in LDS
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] = wdg_row16_sum(s1[r]);
                s2[r] = wdg_row16_sum(s2[r]);
            }
            if ((lane & 15) == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = wn * (BN / WGN) + b * 16 + q4 + r;
                    red[(wm * BN + col) * 2 + 0] = s1[r];
                    red[(wm * BN + col) * 2 + 1] = s2[r];
                }
            }
        }
    }
    if constexpr (EPI == 1) {
        if (p.splitk > 1) return;      // (never launched: the host keeps the statistics hook off split-K launches)
        // column tiles past the padded channel count wrote nothing: their slots are not read below (n < Ncols)
        __syncthreads();
        double* slab = p.stats + (size_t)((blockIdx.x + blockIdx.z) % (unsigned)p.stats_rep) * 2 * p.stats_C;
        for (int c = t; c < BN; c += 256) {
            const int n = n0 + c;
            if (n < p.Ncols) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WGM; ++w) {
                    s1 += red[(w * BN + c) * 2 + 0];
                    s2 += red[(w * BN + c) * 2 + 1];
                }
                atomicAdd(slab + n, (double)s1);
                atomicAdd(slab + p.stats_C + n, (double)s2);
            }
        }
    }
}

// split-K second stage: sum the slabs, apply the epilogue, scatter to the output view.
__global__ void __launch_bounds__(256) wdg_igemm_reduce_kernel(const WdgIgemm p) {
    const WdgPhase ph = p.ph[blockIdx.z];
    const int PaPb = ph.Pa * ph.Pb;
    const int Mph = p.n_img * PaPb;
    const int NcP = (p.Ncols + 3) & ~3;     // slab rows are padded to 4 columns (16-byte stores of the main kernel)
    const long long total = (long long)Mph * p.Ncols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx / p.Ncols);
        const int n = (int)(idx - (long long)m * p.Ncols);
        const float* src = p.partial + (((long long)blockIdx.z * p.splitk) * p.Mmax + m) * NcP + n;
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += src[(long long)s * p.Mmax * NcP];
        const int img = m / PaPb;
        const int rem = m - img * PaPb;
        const int pa = rem / ph.Pb;
        const int pb = rem - pa * ph.Pb;
        const int oh = pa * p.o_mul + ph.o_off_h;
        const int ow = pb * p.o_mul + ph.o_off_w;
        float* dst = p.Out + (long long)img * p.imgStrideO + ((long long)oh * p.Wo + ow) * p.ldO + n;
        if (p.bias) v += p.bias[n];
        if (p.act) v = wdg_lrelu(v, p.slope);
        if (p.accumulate) v += *dst;
        *dst = v;
    }
}

// split-K second stage of a forward conv that feeds a LayerNormalization: one wave per output pixel sums the slabs of
// its row (16-byte loads, lanes along the channels), applies bias + LeakyReLU, normalises the row with two wave
// reductions and writes y, z and (mean, rstd) — the separate wdg_ln_fwd launch and its re-read of y disappear.
// Ncols % 4 == 0, Ncols <= 1024, one phase.
template <int WPR>
__global__ void __launch_bounds__(256) wdg_igemm_reduce_ln_kernel(const WdgIgemm p) {
    // WPR = 4 (few rows: the 2 x 2 and 8 x 8 maps, 128 .. 2048 rows of up to 32 slabs each): the four waves of a block share ONE
    // row — wave w sums slabs w, w + 4, ... and the partial rows meet in LDS (fixed order) — so the chain of dependent slab reads is
    // a quarter as long (one wave per row: 17 us for 128 rows x 32 slabs, twice per discriminator forward)
    __shared__ f32x4 part[WPR > 1 ? 3 * 256 : 1];
    const WdgPhase ph = p.ph[0];
    const int PaPb = ph.Pa * ph.Pb;
    const int Mph = p.n_img * PaPb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4n = p.Ncols >> 2;
    const float invC = 1.f / (float)p.Ncols;
    const int rows_per_block = WPR > 1 ? 1 : 4;
    for (int m0 = blockIdx.x * rows_per_block; m0 < Mph; m0 += gridDim.x * rows_per_block) {
        const int m = WPR > 1 ? m0 : m0 + wave;
        f32x4 v[4];
        float s_ = 0.f;
        if (WPR > 1 || m < Mph) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = lane + 64 * j;
                v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (c4 < c4n) {
                    const float* src = p.partial + (long long)m * p.Ncols + 4 * c4;
                    for (int sp = (WPR > 1 ? wave : 0); sp < p.splitk; sp += WPR) v[j] += *reinterpret_cast<const f32x4*>(src + (long long)sp * p.Mmax * p.Ncols);
                }
            }
        }
        if constexpr (WPR > 1) {
            if (wave > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) part[((wave - 1) * 4 + j) * 64 + lane] = v[j];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += part[(w * 4 + j) * 64 + lane];
            }
        }
        if ((WPR > 1 && wave == 0) || (WPR == 1 && m < Mph)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = lane + 64 * j;
                if (c4 < c4n) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (p.bias) v[j][r] += p.bias[4 * c4 + r];
                        if (p.act) v[j][r] = wdg_lrelu(v[j][r], p.slope);
                        s_ += v[j][r];
                    }
                }
            }
            const float mean = wdg_wave_sum(s_) * invC;
            float q_ = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j < c4n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) q_ += (v[j][r] - mean) * (v[j][r] - mean);
            const float rstd = 1.f / sqrtf(wdg_wave_sum(q_) * invC + p.ln_eps);
            const int img = m / PaPb;
            const int rem = m - img * PaPb;
            const int pa = rem / ph.Pb;
            const int pb = rem - pa * ph.Pb;
            const long long off = (long long)img * p.imgStrideO + ((long long)(pa * p.o_mul + ph.o_off_h) * p.Wo + pb * p.o_mul + ph.o_off_w) * p.ldO;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = lane + 64 * j;
                if (c4 < c4n) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + 4 * c4), bt = *reinterpret_cast<const f32x4*>(p.ln_beta + 4 * c4);
                    f32x4 z;
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = (v[j][r] - mean) * rstd * g[r] + bt[r];
                    *reinterpret_cast<f32x4*>(p.Out + off + 4 * c4) = v[j];
                    *reinterpret_cast<f32x4*>(p.Out2 + off + 4 * c4) = z;
                }
            }
            if (p.mean_rstd && lane == 0) {
                p.mean_rstd[2 * (long long)m] = mean;
                p.mean_rstd[2 * (long long)m + 1] = rstd;
            }
        }
        if constexpr (WPR > 1) __syncthreads();     // (the next row reuses `part`)
    }
}

// split-K second stage of a data gradient whose output is the dz of a LayerNormalization (wdg_conv_dgrad_lnbwd on a split-K
// launch: the discriminator's small maps): one wave per output pixel sums the slabs of its row — so it holds dz for all channels
// of the pixel —, runs the LayerNorm + LeakyReLU backward on channels [lnb_c0, lnb_c0 + lnb_C) and writes the row; parameter
// gradients as in the EPI 5 epilogue (wave partials -> LDS -> one atomic per block, channel and quantity into a replica slab).
// Ncols % 4 == 0, Ncols <= 1024, every phase of the launch (blockIdx.z).
__global__ void __launch_bounds__(256) wdg_igemm_reduce_lnbwd_kernel(const WdgIgemm p) {
    __shared__ float red[4 * 3 * 1024];
    const WdgPhase ph = p.ph[blockIdx.z];
    const int PaPb = ph.Pa * ph.Pb;
    const int Mph = p.n_img * PaPb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4n = p.Ncols >> 2;
    const int c0 = p.lnb_c0, C = p.lnb_C;
    const float invC = 1.f / (float)C;
    const int HoWo = p.Ho * p.Wo;
    const bool par = p.lnb_par != nullptr;
    f32x4 pg[4], pb[4], pd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pg[j] = pb[j] = pd[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int m = blockIdx.x * 4 + wave; m < Mph; m += gridDim.x * 4) {
        const int img = m / PaPb;
        const int rem = m - img * PaPb;
        const int pa = rem / ph.Pb;
        const int pb_ = rem - pa * ph.Pb;
        const long long pix = (long long)(pa * p.o_mul + ph.o_off_h) * p.Wo + pb_ * p.o_mul + ph.o_off_w;
        const long long off = (long long)img * p.imgStrideO + pix * p.ldO;
        const float* yrow = p.lnb_y + (long long)img * p.lnb_imgStride + pix * p.lnb_ldy;
        const float mean = p.lnb_stats[2 * ((long long)img * HoWo + pix)], rstd = p.lnb_stats[2 * ((long long)img * HoWo + pix) + 1];
        f32x4 v[4], y[4], g[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c4 = lane + 64 * j;
            v[j] = y[j] = g[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c4 < c4n) {
                const float* src = p.partial + (((long long)blockIdx.z * p.splitk) * p.Mmax + m) * p.Ncols + 4 * c4;
                for (int sp = 0; sp < p.splitk; ++sp) v[j] += *reinterpret_cast<const f32x4*>(src + (long long)sp * p.Mmax * p.Ncols);
                const int n = 4 * c4;
                if (n >= c0 && n < c0 + C) {
                    y[j] = *reinterpret_cast<const f32x4*>(yrow + (n - c0));
                    g[j] = *reinterpret_cast<const f32x4*>(p.lnb_gamma + (n - c0));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (y[j][r] - mean) * rstd, gg = v[j][r] * g[j][r];
                        s1 += gg;
                        s2 += gg * xh;
                    }
                }
            }
        }
        s1 = wdg_wave_sum(s1) * invC;
        s2 = wdg_wave_sum(s2) * invC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c4 = lane + 64 * j;
            if (c4 >= c4n) continue;
            const int n = 4 * c4;
            f32x4 o = v[j];
            if (n >= c0 && n < c0 + C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (y[j][r] - mean) * rstd;
                    float d = rstd * (v[j][r] * g[j][r] - s1 - xh * s2);
                    if (p.lnb_slope >= 0.f) d *= (y[j][r] > 0.f ? 1.f : p.lnb_slope);
                    pg[j][r] = fmaf(v[j][r], xh, pg[j][r]);
                    pb[j][r] += v[j][r];
                    pd[j][r] += d;
                    o[r] = d;
                }
            }
            *reinterpret_cast<f32x4*>(p.Out + off + n) = o;
        }
    }
    if (!par) return;
    // per-wave partials (a lane owns its channels: no cross-lane sum needed) -> LDS -> the block's four waves summed -> atomics
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < c4n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(wave * 3 + 0) * 1024 + 4 * c4 + r] = pg[j][r];
                red[(wave * 3 + 1) * 1024 + 4 * c4 + r] = pb[j][r];
                red[(wave * 3 + 2) * 1024 + 4 * c4 + r] = pd[j][r];
            }
        }
    }
    __syncthreads();
    float* slab = p.lnb_par + (size_t)((blockIdx.x + blockIdx.z) % (unsigned)p.lnb_rep) * 3 * C;
    for (int idx = threadIdx.x; idx < 3 * p.Ncols; idx += 256) {
        const int which = idx / p.Ncols, n = idx - which * p.Ncols;
        if (n < c0 || n >= c0 + C) continue;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += red[(w * 3 + which) * 1024 + n];
        atomicAdd(slab + which * C + (n - c0), t);
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dW[(tap,ci)][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co].
// Rows are the k4 groups of the forward table (4 channels each); the reduction runs over output
// pixels, 32 per step.  LDS tiles are pixel-major ([pixel][row], row stride = ROWS + 16 floats so
// the two 16-lane halves of a ds_read_b32 hit disjoint banks).
// ------------------------------------------------------------------------------------------
struct WdgWgrad {
    const float* X;
    const float* DY;
    float* dW;
    float* partial;     // [split][rows][Cout]
    const int4* ktab;   // forward table (a_off, dh, dw, valid>=0)
    const int2* wrow;   // per k4 group: {dW element offset of its first row, valid rows (0..4)}
    long long imgStrideX, imgStrideY;
    int n_img, H, W, ldx;
    int Ho, Wo, ldy;
    int stride, pad_h, pad_w;
    int K4;             // row groups (padded to a multiple of 8; padding has valid rows = 0)
    int Cout, Cout_p;
    int w_ld;           // row stride of dW (= Cout unless the plan covers a channel range of a wider layer: wdg_conv_plan_create_sliced)
    int accumulate;
    int splitk;
    long long pix_per_split, Ptot;
    wdg_fastdiv div_howo, div_wo;   // pixel index -> (image, row, column) without integer division
    int xcd_tiles;      // > 0: 1-D grid of xcd_tiles * splitk workgroups, remapped so that one XCD runs the row / column tiles of a
                        // pixel split back to back (they stream the same x and dy window through that XCD's L2)
};

// Fragment layout.  A lane's LDS read is ONE vector of MT (NT) consecutive rows (columns) of one pixel:
// element j of it is the operand of accumulator tile j, i.e. MFMA tile j of a wave owns the rows
// {base + MT*i + j : i = 0..15} — a strided row set instead of 16 consecutive rows, which costs nothing
// (the epilogue undoes the permutation) and turns 2*(MT+NT) ds_read_b32 per 4 pixels into one
// ds_read_b128/b64 per operand.  Pixel-row strides are chosen per read width so that the reads are
// bank-conflict free (MI355X_MICROARCH.md, LDS: b128 -> stride = 0 mod 64 words, b64 -> 32 mod 64,
// b32 -> 16 mod 32).  The fragments of pixel group s+1 are read before the MFMAs of group s.
#ifndef WDG_WGRAD_LB
#define WDG_WGRAD_LB 1
#endif
template <int W>
struct WdgFrag {
    float v[W];
};
template <int W>
__device__ __forceinline__ WdgFrag<W> wdg_lds_frag(const float* q) {
    WdgFrag<W> f;
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

template <int BN, int WGM, int WGN>
__global__ void __launch_bounds__(256, WDG_WGRAD_LB) wdg_wgrad_kernel(const WdgWgrad p) {
    constexpr int BM = 128;
    constexpr int MT = BM / WGM / 16;
    constexpr int NT = BN / WGN / 16;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert((MT == 4 || MT == 2) && (NT == 4 || NT == 2 || NT == 1), "fragment widths");
    constexpr int RSA = MT == 4 ? BM : BM + 32;
    constexpr int RSB = NT == 4 ? (BN + 63) / 64 * 64 : NT == 2 ? (BN % 64 == 32 ? BN : BN + 32) : (BN % 32 == 16 ? BN : BN + 16);
    __shared__ __attribute__((aligned(16))) float ldsA[32 * RSA];
    __shared__ __attribute__((aligned(16))) float ldsB[32 * RSB];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g4 = t & 31;   // row group (A) / column group (B) handled by this thread
    const int ps = t >> 5;   // pixel slot 0..7

    // (round 1 measured an XCD-aware split -> XCD assignment 3-15 % slower: profiles/r01t_perf_conv_wgrad_xcd_negative.log;
    // re-measured per layer in round 4 — wdg_set_tuning("wgrad_xcd"))
    int bx = blockIdx.x, split_id = blockIdx.y;
    if (p.xcd_tiles) {
        const int w = wdg_xcd_remap(blockIdx.x, gridDim.x);
        split_id = w / p.xcd_tiles;
        bx = w - split_id * p.xcd_tiles;
    }
    const int tiles_m = (p.K4 * 4 + BM - 1) / BM;
    const int tm = bx % tiles_m;
    const int tn = bx / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int r4 = (m0 >> 2) + g4;
    int4 e = (int4){0, 0, 0, -1};
    if (r4 < p.K4) e = p.ktab[r4];
    const bool a_row_ok = e.w >= 0;
    const int nb = n0 + 4 * g4;
    const bool b_col_ok = (4 * g4 < BN) && (nb < p.Cout_p);

    const long long pix_begin = (long long)split_id * p.pix_per_split;
    long long pix_end = pix_begin + p.pix_per_split;
    if (pix_end > p.Ptot) pix_end = p.Ptot;
    const int nk = pix_end > pix_begin ? (int)((pix_end - pix_begin + 31) >> 5) : 0;
    const int HoWo = p.Ho * p.Wo;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // operands come through buffer loads relative to the first image of this pixel split; lanes that fall
    // on padding, outside the image or past the end of the split get an out-of-range offset (-> zeros)
    const int img0 = (int)(pix_begin / HoWo);
    const wdg_srd srdX = wdg_make_srd(p.X + (long long)img0 * p.imgStrideX);
    const wdg_srd srdY = wdg_make_srd(p.DY + (long long)img0 * p.imgStrideY);
    const unsigned q0 = (unsigned)(pix_begin - (long long)img0 * HoWo);   // pixel index relative to image img0
    const unsigned q_end = (unsigned)(pix_end - (long long)img0 * HoWo);
    f32x4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned q = q0 + (unsigned)(kt * 32 + ps + 8 * i);
            const unsigned im = wdg_fastdiv_do(q, p.div_howo);
            const unsigned rem = q - im * (unsigned)HoWo;
            const int oh = (int)wdg_fastdiv_do(rem, p.div_wo);
            const int ow = (int)rem - oh * p.Wo;
            const int ih0 = oh * p.stride - p.pad_h, iw0 = ow * p.stride - p.pad_w;
            const int ih = ih0 + e.y, iw = iw0 + e.z;
            const bool in = q < q_end;
            const bool okx = in && a_row_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
            const int offx = (int)im * (int)p.imgStrideX + (ih0 * p.W + iw0) * p.ldx + e.x;
            const int offy = (int)im * (int)p.imgStrideY + (int)rem * p.ldy + nb;
            if constexpr (WDG_KLOOP_EXP & 1) {
                ra[i] = (f32x4){(float)offx, 1.f, (float)okx, 2.f};
                rb[i] = (f32x4){(float)offy, 1.f, (float)in, 2.f};
                continue;
            }
            ra[i] = wdg_buffer_load_f32x4(srdX, okx ? (unsigned)offx << 2 : WDG_SRD_OOB);
            rb[i] = wdg_buffer_load_f32x4(srdY, (in && b_col_ok) ? (unsigned)offy << 2 : WDG_SRD_OOB);
        }
    };

    const float* fragA = ldsA + (lane >> 4) * RSA + wm * (BM / WGM) + MT * (lane & 15);
    const float* fragB = ldsB + (lane >> 4) * RSB + wn * (BN / WGN) + NT * (lane & 15);

    if (nk > 0) load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = ps + 8 * i;
            if constexpr (WDG_KLOOP_EXP & 2) {
                asm volatile("" :: "v"(ra[i][0]), "v"(rb[i][0]));
                continue;
            }
            *reinterpret_cast<f32x4*>(&ldsA[px * RSA + 4 * g4]) = ra[i];
            if (BN >= 128 || 4 * g4 < BN) *reinterpret_cast<f32x4*>(&ldsB[px * RSB + 4 * g4]) = rb[i];
        }
        if constexpr (!(WDG_KLOOP_EXP & 8)) __syncthreads();
        load_tile(kt + 1);   // unconditional (past the end every lane is out of range -> no memory traffic): keeps the
                             // address arithmetic and the loads in the MFMA basic block, where they interleave
        WdgFrag<MT> af[2];
        WdgFrag<NT> bf[2];
        af[0] = wdg_lds_frag<MT>(fragA);
        bf[0] = wdg_lds_frag<NT>(fragB);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) {
                if constexpr (WDG_KLOOP_EXP & 4) {
                    af[(s + 1) & 1] = af[s & 1];
                    bf[(s + 1) & 1] = bf[s & 1];
                    asm volatile("" : "+v"(af[(s + 1) & 1].v[0]), "+v"(bf[(s + 1) & 1].v[0]));
                } else {
                af[(s + 1) & 1] = wdg_lds_frag<MT>(fragA + 4 * (s + 1) * RSA);
                bf[(s + 1) & 1] = wdg_lds_frag<NT>(fragB + 4 * (s + 1) * RSB);
                }
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1].v[a], bf[s & 1].v[b], acc[a][b], 0, 0, 0);
        }
        if constexpr (!(WDG_KLOOP_EXP & 8)) __syncthreads();
    }

    // ---- epilogue.  Accumulator (a, reg r) of lane (i = lane & 15, q = lane >> 4) is logical row 4q + r of
    // tile a = tile row MT*(4q + r) + a, and column NT*i + b of the wave's column block.
    const int q = lane >> 4;
    const int ncol0 = n0 + wn * (BN / WGN) + NT * (lane & 15);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * (BM / WGM) + MT * (4 * q + r) + a;
            const int rg = row >> 2, rr = row & 3;
            if (rg >= p.K4) continue;
            if (p.splitk > 1) {
                float* dst = p.partial + ((long long)split_id * p.K4 * 4 + row) * p.Cout + ncol0;
                if (NT == 4 && (p.Cout & 3) == 0) {
                    if (ncol0 < p.Cout)
                        *reinterpret_cast<f32x4*>(dst) = (f32x4){acc[a][0][r], acc[a][NT > 1 ? 1 : 0][r], acc[a][NT > 2 ? 2 : 0][r], acc[a][NT > 3 ? 3 : 0][r]};
                } else {
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        if (ncol0 + b < p.Cout) dst[b] = acc[a][b][r];
                }
            } else {
                const int2 wr = p.wrow[rg];
                if (rr >= wr.y) continue;
                float* dst = p.dW + wr.x + (long long)rr * p.w_ld + ncol0;
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    if (ncol0 + b < p.Cout) {
                        float v = acc[a][b][r];
                        if (p.accumulate) v += dst[b];
                        dst[b] = v;
                    }
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) wdg_wgrad_reduce_kernel(const WdgWgrad p) {
    // block = 16 consecutive outputs x 16 split lanes; fixed summation order -> reproducible
    __shared__ float red[256];
    const long long total = (long long)p.K4 * 4 * p.Cout;
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (long long base = (long long)blockIdx.x * 16; base < total; base += (long long)gridDim.x * 16) {
        const long long idx = base + el;
        float v = 0.f;
        if (idx < total)
            for (int s = sl; s < p.splitk; s += 16) v += p.partial[(long long)s * p.K4 * 4 * p.Cout + idx];
        red[threadIdx.x] = v;
        __syncthreads();
        if (sl == 0 && idx < total) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k * 16 + el];
            const int R = (int)(idx / p.Cout);
            const int n = (int)(idx - (long long)R * p.Cout);
            const int2 wr = p.wrow[R >> 2];
            const int r = R & 3;
            if (r < wr.y) {
                float* dst = p.dW + wr.x + (long long)r * p.w_ld + n;
                if (p.accumulate) t += *dst;
                *dst = t;
            }
        }
        __syncthreads();
    }
}

// The same for Cout % 4 == 0 (every layer of the two networks but the 2-channel ends): 16-byte accesses.  block = 64 groups of four
// consecutive outputs x 4 split lanes; a lane walks its splits four at a time with the four loads in flight together (the scalar
// form above moved 64 contiguous bytes per split and wave: 31 MB of slabs of the 7x7 stride-3 32 -> 64 layer in 33 us, 1 TB/s).
// Fixed summation order (lane l: splits l, l + 4, ...; then lanes 0..3) -> reproducible.
__global__ void __launch_bounds__(256) wdg_wgrad_reduce4_kernel(const WdgWgrad p) {
    __shared__ f32x4 red[256];
    const long long total4 = (long long)p.K4 * p.Cout;              // groups of four consecutive columns of one row
    const long long stride4 = total4;                                 // one split's slab, in groups
    const int g = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const f32x4* part = reinterpret_cast<const f32x4*>(p.partial);
    for (long long base = (long long)blockIdx.x * 64; base < total4; base += (long long)gridDim.x * 64) {
        const long long idx = base + g;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (idx < total4) {
            int s = sl;
            for (; s + 12 < p.splitk; s += 16) {
                const f32x4 a0 = part[(long long)s * stride4 + idx], a1 = part[(long long)(s + 4) * stride4 + idx],
                            a2 = part[(long long)(s + 8) * stride4 + idx], a3 = part[(long long)(s + 12) * stride4 + idx];
                v += a0;
                v += a1;
                v += a2;
                v += a3;
            }
            for (; s < p.splitk; s += 4) v += part[(long long)s * stride4 + idx];
        }
        red[threadIdx.x] = v;
        __syncthreads();
        if (sl == 0 && idx < total4) {
            f32x4 t = red[g];
            t += red[64 + g];
            t += red[128 + g];
            t += red[192 + g];
            const long long e = idx * 4;
            const int R = (int)(e / p.Cout);
            const int n = (int)(e - (long long)R * p.Cout);
            const int2 wr = p.wrow[R >> 2];
            const int r = R & 3;
            if (r < wr.y) {
                f32x4* dst = reinterpret_cast<f32x4*>(p.dW + wr.x + (long long)r * p.w_ld + n);
                if (p.accumulate) t += *dst;
                *dst = t;
            }
        }
        __syncthreads();
    }
}

// repack master HWIO -> wF [Cout][taps][Cin_p] and wD [taps][Cin][Cout_p]
__global__ void __launch_bounds__(256) wdg_weight_pack_kernel(const float* __restrict__ w, float* wF,
                                                              float* wD, int taps, int Cin, int Cout,
                                                              int Cin_p, int Cout_p) {
    const long long nF = (long long)Cout * taps * Cin_p;
    const long long nD = wD ? (long long)taps * Cin * Cout_p : 0;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nF + nD;
         idx += (long long)gridDim.x * 256) {
        if (idx < nF) {
            const int ci = (int)(idx % Cin_p);
            const long long r = idx / Cin_p;
            const int tap = (int)(r % taps);
            const int co = (int)(r / taps);
            wF[idx] = ci < Cin ? w[((long long)tap * Cin + ci) * Cout + co] : 0.f;
        } else {
            const long long j = idx - nF;
            const int co = (int)(j % Cout_p);
            const long long r = j / Cout_p;  // tap*Cin + ci
            wD[j] = co < Cout ? w[r * Cout + co] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Host side: plans
// ------------------------------------------------------------------------------------------
static int g_cus = 0;
extern "C" int wdg_device_cus(void) {
    if (g_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_cus;
}

static int g_tile160 = 1;
static int g_tile80 = 2;     // 80-column tiles for column counts like 400 (no padding): 2 = 128x80 (75.56 -> 75.29 ms/step), 1 = 256x80 (slower)
struct TileCfg {
    int BM, BN;
};
// largest tile whose padded column count stays within 13 % of the best achievable padding
// Wide layers on small maps (the discriminator's 84->27->8 stages): when the 128x128 tiling yields between 16 and
// g_small_m tiles — less than one workgroup per CU — the 128x64 tile doubles the tile count and needs a smaller
// split-K factor (64->128 forward 211 -> 189 us, 128->256 forward 82 -> 77 us); with only a handful of tiles (the
// 8->2 stage) the split factor dominates either way and 128x128 stays.  wdg_set_tuning("small_m", n): n = 0 disables.
static int g_small_m = 256;
static int g_tile64 = 2;     // 64x64 tile when even the 128x64 tiling leaves fewer than two workgroups per CU (T=24 step 120.0 -> 118.9 ms, headline 78.4 -> 78.2); 2: also for layers with fewer than 16 tiles of 128x128 (another -0.35 ms)
static TileCfg pick_tile(int ncols, bool igemm = true, long long M = -1) {
    if (igemm && M >= 0 && ncols >= 128) {
        const long long t128 = ((M + 127) / 128) * ((ncols + 127) / 128);
        if (g_tile64 && (t128 >= 16 || g_tile64 >= 2) && 2 * t128 < 512) return TileCfg{64, 64};
        if (t128 >= 16 && t128 <= g_small_m) return TileCfg{128, 64};
    }
    // 160 columns (the generator's widest decoder layer): one 128 x 160 tile, five column fragments per wave,
    // instead of five 256 x 32 tiles (2.2 instead of 1.3 MFMAs per LDS fragment read)
    if (igemm && g_tile160 && ncols % 160 == 0) return TileCfg{128, 160};
    if (igemm && g_tile80 && ncols % 80 == 0 && ncols >= 240) return TileCfg{g_tile80 == 2 ? 128 : 256, 80};
    static const TileCfg cand[4] = {{128, 128}, {128, 64}, {256, 32}, {256, 16}};
    int best = 1 << 30;
    for (auto& c : cand) best = std::min(best, wdg_round_up(ncols, c.BN));
    // many columns: the 128x128 tile beats the 128x64 one even with up to 30 % padding (400 columns of the column-form
    // upsample-conv forward: 0.82 vs 0.91 ms)
    if (igemm && ncols >= 256 && wdg_round_up(ncols, 128) * 100 <= best * 130) return cand[0];
    for (auto& c : cand)
        if (wdg_round_up(ncols, c.BN) * 100 <= best * 113) return c;
    return cand[3];
}
// wdg_set_tuning("wgrad_xcd", n): XCD-contiguous order for weight gradients with 2..n tiles per pixel split (0 = off).  Round 5: ON — the
// row / column tiles of one pixel split then run back to back on ONE XCD and share its L2 copy of the x / dy window: fabric fetch of
// the 7x7 stride-3 32 -> 64 weight gradient 1,726 -> 438 MB per launch, L2 hit 0.16 -> 0.78 (profiles/r05b_*: time in isolation
// unchanged, 464 vs 467 us — the kernel is not bound by its fetch —, but 1.3 GB less fabric traffic per launch for whatever runs beside it)
static int g_wgrad_xcd = 64;
static int g_reduce_wpr = 1;        // wdg_set_tuning("reduce_wpr", 0/1): four waves per row in the split-K + LayerNorm second stage of the small maps
static int g_wgrad_reduce4 = 1;     // wdg_set_tuning("wgrad_reduce4", 0/1): 16-byte second stage of the split weight gradients
static int g_tap_chunk_order = 1;   // wdg_set_tuning("tap_chunk_order", 0/1): stride-1 layers with > 32 input channels in chunk-major tap order (plans created afterwards)
static int g_tap_class_order = 1;   // wdg_set_tuning("tap_class_order", 0/1): forward tables of strided layers in residue-class order (plans created afterwards)
static int g_wgrad_bn160 = 32;   // wdg_set_tuning("wgrad_bn160", 32 | 64 | 128): column tile of 160-column weight gradients
static int pick_wgrad_bn(int ncols) {
    if (ncols == 160 && g_wgrad_bn160 != 32) return g_wgrad_bn160;
    return pick_tile(ncols, false).BN;
}
// wdg_set_tuning("force_{fwd,dgrad,wgrad}_split", n): n > 0 overrides the split chosen at plan creation (sweeps)
static int g_force_split[3] = {0, 0, 0};
// resident workgroups per CU (512 unified VGPRs per lane and SIMD: 244 -> 2 waves, 156 -> 3, ...)
static int igemm_blocks_per_cu(int bn) { return bn >= 128 ? 2 : bn >= 64 ? 3 : bn >= 32 ? 2 : 3; }
static int wgrad_blocks_per_cu(int bn) { return bn >= 128 ? 2 : bn >= 64 ? 4 : 5; }

// Split factor from a wave-quantisation model: `tiles * s` equal workgroups run in rounds of `cap` = CUs x
// resident workgroups; a last round that leaves at most one workgroup per CU finishes in ~0.65 of a full
// round (no sharing of the matrix pipe).  Relative time = rounds / s, plus 0.3 % per split for the slab
// write + reduce.  Constants fitted to profiles/r01u_sweep_split.log (multiples of the resident capacity win;
// 1.5 rounds lose 20-25 %).
static int pick_split_model(long long tiles, int max_split, int cus, int per_cu) {
    const long long cap = (long long)cus * per_cu;
    if (tiles >= cap || max_split <= 1) return 1;   // a full round exists already: the slab traffic would cost more than the tail
    double best_t = 1e30;
    int best = 1;
    for (int s = 1; s <= max_split && s <= 256; ++s) {
        const long long blocks = tiles * s;
        const long long rounds = (blocks + cap - 1) / cap;
        const long long rem = blocks - (rounds - 1) * cap;
        double t = ((double)(rounds - 1) + (rem <= cus ? 0.55 : 1.0)) / s;
        t *= 1.0 + 0.003 * (s - 1);
        if (t < best_t * 0.999) {
            best_t = t;
            best = s;
        }
    }
    return best;
}
// implicit GEMM: at least 16 k4 groups (two K-steps) per split
static int pick_split(long long tiles, int K4, int cus, int bn) {
    if (K4 < 32) return 1;
    return pick_split_model(tiles, std::max(1, K4 / 16), cus, igemm_blocks_per_cu(bn));
}

static int conv_plan_create_impl(wdg_conv_plan** out, const wdg_conv_geom* g, int w_ld);
extern "C" int wdg_conv_plan_create(wdg_conv_plan** out, const wdg_conv_geom* g) { return conv_plan_create_impl(out, g, 0); }

// A plan for output channels [n0, n0 + geom->Cout) of a layer with w_ld output channels (the live gate columns of a ConvLSTM2D at
// n_timesteps = 1: the forget gate multiplies c_0 = 0, its quarter of the input convolution is dead in all three directions).
// Forward: wF rows are contiguous per output channel, so the range is a row range of the full packed matrix (any plan does).
// wdg_conv_dgrad / wdg_conv_wgrad: pass wD + n0 / dW + n0 of the FULL HWIO tensors; y / dy views start at channel n0.
extern "C" int wdg_conv_plan_create_sliced(wdg_conv_plan** out, const wdg_conv_geom* g, int w_ld) {
    WDG_CHECK_ARG(g && w_ld % 4 == 0 && w_ld >= g->Cout && g->Cout % 4 == 0, "sliced plan: w_ld and Cout multiples of 4, w_ld >= Cout");
    const int rc = conv_plan_create_impl(out, g, w_ld);
    if (rc != WDG_OK) return rc;
    wdg_conv_plan* pl = *out;
    if ((pl->halo_auto_fwd && pl->halo_fwd_nt) || (pl->halo_auto_dgrad && pl->halo_dgrad_nt) || wdg_wgrad_halo_eligible(pl) ||
        wdg_wgrad_thin_eligible(pl)) {
        wdg_conv_plan_destroy(pl);
        *out = nullptr;
        wdg_set_error("sliced plan: only layers that run on the implicit-GEMM kernels");
        return WDG_ERR_ARG;
    }
    return WDG_OK;
}

static int conv_plan_create_impl(wdg_conv_plan** out, const wdg_conv_geom* g, int w_ld) {
    WDG_CHECK_ARG(out && g, "null argument");
    WDG_CHECK_ARG(g->n_img > 0 && g->H > 0 && g->W > 0 && g->Cin > 0 && g->Cout > 0, "bad sizes");
    WDG_CHECK_ARG(g->kh > 0 && g->kw > 0 && g->stride > 0 && g->stride <= 3, "bad kernel/stride");
    WDG_CHECK_ARG(g->ldx % 4 == 0 && g->ldy % 4 == 0, "pixel strides must be multiples of 4");
    WDG_CHECK_ARG(g->ldx >= wdg_round_up(g->Cin, 4) && g->ldy >= wdg_round_up(g->Cout, 4), "ld < padded C");
    const int Ho = (g->H + 2 * g->pad_h - g->kh) / g->stride + 1;
    const int Wo = (g->W + 2 * g->pad_w - g->kw) / g->stride + 1;
    WDG_CHECK_ARG(Ho == g->Ho && Wo == g->Wo, "Ho/Wo inconsistent with H,W,k,stride,pad");
    WDG_CHECK_ARG((long long)g->n_img * g->Ho * g->Wo < (1LL << 31) && (long long)g->n_img * g->H * g->W < (1LL << 31),
                  "pixel count overflows int32");

    wdg_conv_plan* pl = new wdg_conv_plan();
    pl->g = *g;
    pl->Cin_p = wdg_round_up(g->Cin, 4);
    pl->Cout_p = wdg_round_up(g->Cout, 4);
    pl->taps = g->kh * g->kw;
    pl->cus = wdg_device_cus();
    pl->w_ld = w_ld;
    const int s = g->stride;

    // ---- forward table
    std::vector<int4> tf;
    std::vector<int2> wr;
    // Tap order.  With stride s > 1 the taps fall into s * s residue classes (th mod s, tw mod s): the taps of one class read
    // the SAME residue grid of input pixels, shifted by whole output pixels, while taps of different classes share no input
    // pixel at all.  In row-major tap order a class comes back every s-th tap / tap row — for the 7 x 7 stride-3 layer a tile's
    // input line is re-read after ~21 K-steps, by which time the CU's other workgroups have pushed it out of L1 and L2 (fabric
    // fetch 4.3 x the input).  Class-major order makes the re-reads consecutive K-steps.  (The table carries the weight offset
    // of every entry, so the reduction order is free; the weight-gradient rows follow the same order through `wrow`.)
    std::vector<std::pair<int, int>> tap_order;
    const int cls = g_tap_class_order ? g->stride : 1;
    for (int a = 0; a < cls; ++a)
        for (int b = 0; b < cls; ++b)
            for (int th = a; th < g->kh; th += cls)
                for (int tw = b; tw < g->kw; tw += cls) tap_order.push_back(std::make_pair(th, tw));
    // Channel chunks.  A K-step is 8 table entries = 32 channels of one tap.  With more than 32 input channels and stride 1 (one
    // residue class: consecutive taps read the same lines shifted by a pixel) tap-major order walks ALL channel chunks of a tap
    // before the next tap returns to the first chunk's lines — Cin / 32 K-steps later, by which time the other resident workgroups
    // have evicted them (3x3 128 -> 64 forward: 541 MB fetched for a 67 MB input).  Chunk-major order (all taps of one 32-channel
    // chunk, then the next chunk) makes the re-reads consecutive K-steps.  wdg_set_tuning("tap_chunk_order", 0/1).
    const int nchunk = (g_tap_chunk_order && g->stride == 1 && pl->Cin_p > 32 && pl->Cin_p % 32 == 0) ? pl->Cin_p / 32 : 1;
    const int c4_per_chunk = pl->Cin_p / 4 / nchunk;
    for (int chunk = 0; chunk < nchunk; ++chunk)
    for (auto& tt : tap_order) {
        const int th = tt.first, tw = tt.second;
        {
            for (int c4 = chunk * c4_per_chunk; c4 < (chunk + 1) * c4_per_chunk; ++c4) {
                const int tap = th * g->kw + tw;
                int4 e;
                e.x = (th * g->W + tw) * g->ldx + 4 * c4;
                e.y = th;
                e.z = tw;
                e.w = tap * pl->Cin_p + 4 * c4;
                tf.push_back(e);
                int2 r;
                r.x = (tap * g->Cin + 4 * c4) * (w_ld ? w_ld : g->Cout);
                r.y = std::min(4, g->Cin - 4 * c4);
                wr.push_back(r);
            }
        }
    }
    while (tf.size() % 8) {
        tf.push_back((int4){0, -(1 << 28), -(1 << 28), -1});
        wr.push_back((int2){0, 0});
    }
    pl->K4_fwd = (int)tf.size();

    // ---- dgrad tables, one phase per (rh, rw) residue
    std::vector<int4> td;
    for (int rh = 0; rh < s; ++rh)
        for (int rw = 0; rw < s; ++rw) {
            WdgPhase ph;
            ph.Pa = rh < g->H ? (g->H - rh + s - 1) / s : 0;
            ph.Pb = rw < g->W ? (g->W - rw + s - 1) / s : 0;
            ph.a_off_h = 0;
            ph.a_off_w = 0;
            ph.o_off_h = rh;
            ph.o_off_w = rw;
            ph.tab_off = (int)td.size();
            int cnt = 0;
            // (the taps of a phase form one residue class; with more than 32 output channels the reduction walks them chunk-major,
            // as the forward table: a 3x3 128 -> 512 data gradient otherwise returns to a dy line 16 K-steps later)
            const int dchunks = (g_tap_chunk_order && pl->Cout_p > 32 && pl->Cout_p % 32 == 0) ? pl->Cout_p / 32 : 1;
            const int dc4 = pl->Cout_p / 4 / dchunks;
            for (int chunk = 0; chunk < dchunks; ++chunk)
            for (int th = (rh + g->pad_h) % s; th < g->kh; th += s)
                for (int tw = (rw + g->pad_w) % s; tw < g->kw; tw += s) {
                    const int dh = (rh + g->pad_h - th) / s;  // exact: numerator divisible by s
                    const int dw = (rw + g->pad_w - tw) / s;
                    const int tap = th * g->kw + tw;
                    for (int c4 = chunk * dc4; c4 < (chunk + 1) * dc4; ++c4) {
                        int4 e;
                        e.x = (dh * g->Wo + dw) * g->ldy + 4 * c4;
                        e.y = dh;
                        e.z = dw;
                        e.w = tap * g->Cin * (w_ld ? w_ld : pl->Cout_p) + 4 * c4;
                        td.push_back(e);
                        ++cnt;
                    }
                }
            while (cnt % 8) {
                td.push_back((int4){0, -(1 << 28), -(1 << 28), -1});
                ++cnt;
            }
            ph.K4 = cnt;
            wdg_phase_finish(ph);
            pl->K4_dgrad_max = std::max(pl->K4_dgrad_max, cnt);
            pl->ph_dgrad.push_back(ph);
        }

    // ---- split-K choices and workspace
    {
        const long long M = (long long)g->n_img * g->Ho * g->Wo;
        TileCfg tc = pick_tile(g->Cout, true, M);
        long long tiles = ((M + tc.BM - 1) / tc.BM) * ((g->Cout + tc.BN - 1) / tc.BN);
        pl->fwd_split = pick_split(tiles, pl->K4_fwd, pl->cus, tc.BN);
        if (g_force_split[0] > 0) pl->fwd_split = std::min(g_force_split[0], std::max(1, pl->K4_fwd / 8));
        if (pl->fwd_split > 1) pl->ws_bytes = std::max(pl->ws_bytes, (size_t)((size_t)pl->fwd_split * M * pl->Cout_p * 4));
    }
    {
        long long Mmax = 0;
        for (auto& ph : pl->ph_dgrad) Mmax = std::max(Mmax, (long long)g->n_img * ph.Pa * ph.Pb);
        TileCfg tc = pick_tile(g->Cin, true, Mmax);
        long long tiles = ((Mmax + tc.BM - 1) / tc.BM) * ((g->Cin + tc.BN - 1) / tc.BN) * (long long)pl->ph_dgrad.size();
        pl->dgrad_split = pick_split(tiles, pl->K4_dgrad_max, pl->cus, tc.BN);
        if (g_force_split[1] > 0) pl->dgrad_split = std::min(g_force_split[1], std::max(1, pl->K4_dgrad_max / 8));
        if (pl->dgrad_split > 1)
            pl->ws_bytes = std::max(pl->ws_bytes, (size_t)((size_t)pl->dgrad_split * pl->ph_dgrad.size() * Mmax * pl->Cin_p * 4));
    }
    {
        const long long P = (long long)g->n_img * g->Ho * g->Wo;
        const int bn = pick_wgrad_bn(g->Cout);
        long long tiles = (long long)((pl->K4_fwd * 4 + 127) / 128) * ((g->Cout + bn - 1) / bn);
        long long maxs = std::max<long long>(1, P / 512);  // >= 512 pixels (16 steps) per split
        long long want = pick_split_model(tiles, (int)std::min<long long>(maxs, 256), pl->cus, wgrad_blocks_per_cu(bn));
        if (g_force_split[2] > 0) want = std::min<long long>(g_force_split[2], std::max<long long>(1, P / 64));
        pl->wgrad_split = (int)std::max<long long>(1, want);
        if (pl->wgrad_split > 1)
            pl->ws_bytes = std::max(pl->ws_bytes, (size_t)pl->wgrad_split * pl->K4_fwd * 4 * g->Cout * 4);
    }

    auto upload = [&](const void* src, size_t bytes, void** dst) -> int {
        WDG_HIP(hipMalloc(dst, std::max<size_t>(bytes, 16)));
        if (bytes) WDG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return WDG_OK;
    };
    int rc;
    if ((rc = upload(tf.data(), tf.size() * sizeof(int4), (void**)&pl->d_tab_fwd)) != WDG_OK) { delete pl; return rc; }
    if ((rc = upload(wr.data(), wr.size() * sizeof(int2), (void**)&pl->d_wrow)) != WDG_OK) { delete pl; return rc; }
    if ((rc = upload(td.data(), td.size() * sizeof(int4), (void**)&pl->d_tab_dgrad)) != WDG_OK) { delete pl; return rc; }
    if ((rc = wdg_halo_plan_init(pl)) != WDG_OK) { delete pl; return rc; }
    pl->ws_bytes = std::max(pl->ws_bytes, wdg_wgrad_halo_ws_bytes(pl));
    pl->ws_bytes = std::max(pl->ws_bytes, wdg_wgrad_thin_ws_bytes(pl));
    pl->ws_bytes = std::max(pl->ws_bytes, wdg_dgrad_s3_ws_bytes(pl));
    *out = pl;
    return WDG_OK;
}

extern "C" int wdg_conv_plan_destroy(wdg_conv_plan* pl) {
    if (!pl) return WDG_OK;
    if (pl->d_tab_fwd) (void)hipFree(pl->d_tab_fwd);
    if (pl->d_wrow) (void)hipFree(pl->d_wrow);
    if (pl->d_tab_dgrad) (void)hipFree(pl->d_tab_dgrad);
    wdg_halo_plan_free(pl);
    delete pl;
    return WDG_OK;
}

extern "C" size_t wdg_conv_ws_bytes(const wdg_conv_plan* pl) { return pl ? pl->ws_bytes : 0; }

// info[0..7] = {fwd BM, fwd BN, fwd split, dgrad BM, dgrad BN, dgrad split, wgrad BN, wgrad split}
extern "C" int wdg_conv_plan_info(const wdg_conv_plan* pl, int32_t* info) {
    WDG_CHECK_ARG(pl && info, "null argument");
    long long Mf = (long long)pl->g.n_img * pl->g.Ho * pl->g.Wo, Md = 0;
    for (auto& ph : pl->ph_dgrad) Md = std::max(Md, (long long)pl->g.n_img * ph.Pa * ph.Pb);
    TileCfg f = pick_tile(pl->g.Cout, true, Mf), d = pick_tile(pl->g.Cin, true, Md);
    info[0] = f.BM; info[1] = f.BN; info[2] = pl->fwd_split;
    info[3] = d.BM; info[4] = d.BN; info[5] = pl->dgrad_split;
    info[6] = pick_wgrad_bn(pl->g.Cout); info[7] = pl->wgrad_split;
    if (wdg_wgrad_halo_eligible(pl)) { info[6] = 0; info[7] = 1; }   // BN = 0 marks the halo weight-gradient kernel
    if (wdg_wgrad_thin_eligible(pl)) { info[6] = -1; info[7] = 1; }  // BN = -1 marks the thin 3x3 weight-gradient kernel
    if (pl->halo_auto_fwd && pl->halo_fwd_nt) { info[0] = 0; info[1] = 16 * pl->halo_fwd_nt; info[2] = 1; }     // BM = 0 marks the halo kernel
    if (pl->halo_auto_dgrad && pl->halo_dgrad_nt) { info[3] = 0; info[4] = 16 * pl->halo_dgrad_nt; info[5] = 1; }
    return WDG_OK;
}

// tuning knob (wdg_set_tuning): 0 = single LDS stage / two barriers, 1 = double-buffered LDS / one barrier,
// 2 = 1 + fragment prefetch
static int g_xcd_swizzle = 1;
static int g_n_fastest = 1;
static long long g_n_fastest_bytes = 2 << 20;
static int g_tile2d = 1;        // 2-D row tiles in the implicit GEMM (WdgPhase::t2_w)
static int g_phase_major = 1;   // strided data gradients: the s*s phases of an output tile adjacent in the launch order (same XCD)
static int g_igemm_pipe = 3;   // measured (profiles/r01ad_perf_conv_pipe3.log): the rotated single-block loop is 2-12 % faster than 0, 1, 2

static int g_igemm_dma = 4;        // wdg_set_tuning("igemm_dma", mask): tiles that run the LDS-DMA K loop (launch_igemm)
// wdg_set_tuning("dgrad_lnbwd", bits): bit 0: LayerNorm backward in the data gradient's epilogue / split-K second stage (wdg_conv_dgrad_lnbwd);
// bit 1: a 64 x 128 tile (a 128-channel row in one wave) where the plain launch would take 64 x 64 tiles — measured SLOWER on the 27 x 27 map
// of the discriminator's third block (137 us against 101 + 15 us for the 64 x 64 launch + the standalone pass: matrix pipe busy 0.38 at
// 1.4 workgroups per CU, profiles/r05k_pmc_summary.csv) and off
static int g_igemm_pad_kb[4] = {0, 0, 0, 0};      // (see launch_variant) BN 32, 64, 80 / 160, 128
static int g_dgrad_lnbwd = 1;
static int g_ln_wave = 1;     // wdg_set_tuning("ln_wave", 0/1): the 128 x 64 tile's LayerNorm epilogue on 4 x 1 waves (in-wave reductions)
static int g_igemm_kg2 = 1;   // wdg_set_tuning("igemm_kg2", 0/1): in-workgroup split of the reduction for the ConvLSTM step epilogue
static int g_tuning_epoch = 0;
extern "C" int wdg_tuning_epoch(void) { return g_tuning_epoch; }
extern "C" int wdg_set_tuning(const char* key, int value) {
    ++g_tuning_epoch;          // (callers that cache launch sequences — captured HIP graphs — key them on this)
    if (key && !strcmp(key, "igemm_pipe")) {
        if (value < 0 || value > 3) return WDG_ERR_ARG;
        g_igemm_pipe = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "patch_lstm_small")) {
        wdg_patch_h16_set_lstm_small(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "ln_wave")) {
        g_ln_wave = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "igemm_kg2")) {
        g_igemm_kg2 = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "lstm16_step")) {
        wdg_lstm16_set_step(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo_weights_global")) {
        wdg_halo_set_wg(value);
        return WDG_OK;
    }
    if (key && (!strcmp(key, "force_fwd_split") || !strcmp(key, "force_dgrad_split") || !strcmp(key, "force_wgrad_split"))) {
        g_force_split[key[6] == 'f' ? 0 : key[6] == 'd' ? 1 : 2] = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "tile160")) {   // takes effect for plans created afterwards
        g_tile160 = value != 0;
        return WDG_OK;
    }

    if (key && !strcmp(key, "tile80")) {
        g_tile80 = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "tile64")) {   // takes effect for plans created afterwards
        g_tile64 = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "small_m")) {   // takes effect for plans created afterwards
        g_small_m = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "h16_small_tiles")) {
        wdg_h16_set_small_tiles(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo_max_cin")) {   // takes effect for plans created afterwards
        wdg_halo_set_max_cin(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo16_thin")) {
        wdg_halo_bf16_set_thin(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo1_stage")) {
        wdg_halo_set_stage(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo_persistent")) {
        wdg_halo_set_persistent(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "wgrad_thin")) {
        wdg_wgrad_thin_enable(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "wgrad_xcd")) {
        g_wgrad_xcd = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "bn_bwd_blocks")) {
        wdg_bn_set_bwd_blocks(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "gather_xcd")) {
        wdg_upconv_set_gather_xcd(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "igemm_dma")) {
        g_igemm_dma = value;
        return WDG_OK;
    }
    if (key && !strncmp(key, "igemm_pad", 9)) {
        const int bn = atoi(key + 9);
        g_igemm_pad_kb[bn == 32 ? 0 : bn == 64 ? 1 : bn == 128 ? 3 : 2] = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "dgrad_s3")) {
        wdg_dgrad_s3_set(value);        // 0: the 7 x 7 stride-3 32 -> 64 data gradient back on the implicit-GEMM route
        return WDG_OK;
    }
    if (key && !strcmp(key, "dgrad_lnbwd")) {
        g_dgrad_lnbwd = value;          // bit 0: on; bit 1: a 64 x 128 tile for 128-channel rows on small maps
        return WDG_OK;
    }
    if (key && !strcmp(key, "reduce_wpr")) {
        g_reduce_wpr = value != 0;
        return WDG_OK;
    }
    if (key && !strcmp(key, "wgrad_reduce4")) {
        g_wgrad_reduce4 = value != 0;
        return WDG_OK;
    }
    if (key && !strcmp(key, "tap_chunk_order")) {
        g_tap_chunk_order = value != 0;
        return WDG_OK;
    }
    if (key && !strcmp(key, "tap_class_order")) {
        g_tap_class_order = value != 0;
        return WDG_OK;
    }
    if (key && !strcmp(key, "wgrad_bn160")) {
        g_wgrad_bn160 = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "n_fastest")) {
        g_n_fastest = value != 0;
        if (value > 1) g_n_fastest_bytes = (long long)value << 10;   // value > 1: B-operand limit in KiB
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo_ln")) {
        wdg_halo_set_ln(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "halo_th4")) {
        wdg_halo_set_th4(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "lstm_step_fused")) {
        wdg_halo_set_lstm_fused(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "lstm16_fused")) {
        wdg_h16_set_lstm_fused(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "patch_flat")) {
        wdg_patch_h16_set_flat(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "patch_nloop")) {
        wdg_patch_h16_set_nloop(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "patch_dbg")) {
        wdg_patch_h16_set_dbg(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "patch_h16")) {
        wdg_patch_h16_set(value != 0);
        if (value > 1) wdg_patch_h16_set_budget(value);   // value > 1: LDS bytes of a patch chunk in KiB
        return WDG_OK;
    }
    if (key && !strcmp(key, "tile2d")) {      // 0 off, 1 on, 2 = only the evenly dividing power-of-two patches (A/B of the ragged ones)
        g_tile2d = value;
        return WDG_OK;
    }
    if (key && !strcmp(key, "lstm2_thin")) {
        wdg_cl2_set_thin(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "convlstm1_mfma")) {
        wdg_convlstm1_set_mfma(value);
        return WDG_OK;
    }
    if (key && !strcmp(key, "phase_major")) {
        g_phase_major = value != 0;
        return WDG_OK;
    }
    if (key && !strcmp(key, "xcd_swizzle")) {
        g_xcd_swizzle = value != 0;
        return WDG_OK;
    }
    wdg_set_error("wdg_set_tuning: unknown key");
    return WDG_ERR_ARG;
}

// wdg_set_tuning("igemm_pad<BN>", k): k KB added to the LDS request of the implicit-GEMM tiles with BN columns — fewer resident
// workgroups per CU (measurement: the patch data-gradient kernel is faster with two per CU than with three)
static int igemm_pad_slot(int bn) { return bn == 32 ? 0 : bn == 64 ? 1 : bn == 128 ? 3 : 2; }

template <int BM, int BN, int WGM, int WGN, int PIPE, int EPI = 0, int KG = 1>
static int launch_variant(dim3 grid, dim3 block, hipStream_t st, const WdgIgemm& p) {
    constexpr size_t lds = (size_t)((PIPE == 0 || PIPE == 3) ? KG : 2) * 8 * (BM + BN) * sizeof(f32x4);   // (PIPE 5: two stages)
    static_assert(KG == 1 || lds >= (size_t)BM * BN * sizeof(float), "the second group's accumulators fit the two stages");
    static_assert(EPI != 1 || lds >= (size_t)WGM * BN * 2 * sizeof(float), "statistics scratch fits the K-loop stage");
    static_assert(EPI != 3 || lds >= (size_t)BM * 4 * WGN * sizeof(float), "LayerNorm scratch fits the K-loop stage");
    static_assert(EPI != 5 || lds >= (size_t)WGM * 3 * BN * sizeof(float), "LayerNorm-backward scratch fits the K-loop stage");
    static_assert(PIPE != 5 || EPI == 0 || EPI == 3, "LDS-DMA loop: instantiated for the plain and LayerNorm-forward epilogues");
    static bool attr_set = false;
    if (!attr_set) {
        if (lds > 48 * 1024)
            WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_igemm_kernel<BM, BN, WGM, WGN, PIPE, EPI, KG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const size_t pad = (size_t)g_igemm_pad_kb[igemm_pad_slot(BN)] * 1024;
    if (pad) {
        static bool attr_pad = false;
        if (!attr_pad) {
            WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_igemm_kernel<BM, BN, WGM, WGN, PIPE, EPI, KG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_pad = true;
        }
    }
    hipLaunchKernelGGL((wdg_igemm_kernel<BM, BN, WGM, WGN, PIPE, EPI, KG>), grid, dim3(block.x * KG), lds + pad, st, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// tiles whose EPI 5 instantiation exists (4 x 1 waves: a row's channels in one wave)
static bool lnb_tile_ok(const TileCfg& tc) { return (tc.BM == 256 && tc.BN <= 32) || (tc.BM == 128 && tc.BN == 64) || (tc.BM == 64 && tc.BN == 128); }

static int launch_igemm(WdgIgemm& p, int nphase, int K4max, int split, void* ws, size_t ws_bytes,
                        hipStream_t st, bool* bn_fused = nullptr, const TileCfg* force_tile = nullptr) {
    TileCfg tc = force_tile ? *force_tile : pick_tile(p.Ncols, true, p.Mmax);
    // ConvLSTM step epilogue (complete sums needed, no split-K across workgroups): with fewer tiles than CUs the reduction is
    // split inside the workgroup instead — the widest tile that still gives every other CU a workgroup
    int kg = 1;
    if (p.lstm_F && g_igemm_kg2 && split == 1 && nphase == 1 && g_igemm_pipe == 3 && K4max % 16 == 0 && p.Ncols % 64 == 0) {
        const int cus = wdg_device_cus();
        const long long t128 = (long long)((p.Mmax + 127) / 128) * ((p.Ncols + 127) / 128);
        const long long t64 = (long long)((p.Mmax + 63) / 64) * ((p.Ncols + 63) / 64);
        if (g_igemm_kg2 == 3 && p.Ncols % 128 == 0 && t128 <= cus && 2 * t128 >= cus) { tc.BM = 128; tc.BN = 128; kg = 2; }
        else if (t64 <= (g_igemm_kg2 == 2 ? 3 : 1) * cus) { tc.BM = 64; tc.BN = 64; kg = 2; }
    }
    const int tiles_m = (p.Mmax + tc.BM - 1) / tc.BM;
    const int tiles_n = (p.Ncols + tc.BN - 1) / tc.BN;
    if (p.Mmax <= 0) return WDG_OK;
    p.splitk = split;
    p.nphase = nphase;
    int per = (K4max + split - 1) / split;
    per = wdg_round_up(per, 8);
    p.k4_per_split = per;
    // drop empty trailing splits
    split = (K4max + per - 1) / per;
    p.splitk = split;
    if (kg == 2) p.k4_per_split = K4max / 2;       // (a multiple of 8: both groups run the same number of K-steps)
    if (split > 1) {
        const size_t need = (size_t)split * nphase * p.Mmax * wdg_round_up(p.Ncols, 4) * sizeof(float);
        if (!ws || ws_bytes < need) {
            wdg_set_error("igemm: workspace too small (%zu < %zu)", ws_bytes, need);
            return WDG_ERR_WORKSPACE;
        }
        p.partial = (float*)ws;
    } else {
        p.partial = nullptr;
    }
    int tiles_m_launch = tiles_m;
    if (g_tile2d && split == 1) {
        // rows of a tile = a 2-D patch of output pixels (8 x 16 for 128 rows, 8 x 8 for 64, 16 x 16 for 256; 6 x 21 on the
        // 84 x 84 map of the discriminator's first strided layer) where the phase's pixel grid divides: a 128-row tile of the
        // 8x8 stride-2 layer then touches 22 x 38 input pixels instead of 8 x 262.  A patch may leave a few rows of its tile
        // idle, so the launch's row space is per phase (virtual rows = tiles * BM).  (Split-K launches keep the linear
        // order: their second stage maps rows linearly.)
        long long rows_max = 0;
        for (int i = 0; i < nphase; ++i) {
            const int per_img = g_tile2d >= 2 && !(p.ph[i].Pa % 8 == 0 && p.ph[i].Pb % 16 == 0) ? 0 : wdg_phase_tile2d(p.ph[i], tc.BM);
            rows_max = std::max(rows_max, per_img ? (long long)p.n_img * per_img : (long long)p.n_img * p.ph[i].Pa * p.ph[i].Pb);
        }
        tiles_m_launch = (int)((rows_max + tc.BM - 1) / tc.BM);
        p.Mmax = (int)rows_max;        // (the kernel derives its row-tile count from it; the split-K slabs that index by it are not in play)
    }
    dim3 grid(tiles_m_launch * tiles_n, split, nphase), block(256);
    // several column tiles over a small B operand (the 400-column GEMM of the column-form upsample-conv: 5 tiles, 256 KB of
    // weights): walk the column tiles of a row tile back to back, so its A rows are fetched from the fabric once, not once
    // per column tile
    p.n_fastest = g_n_fastest && tiles_n > 1 && (long long)p.Ncols * K4max * 16 <= g_n_fastest_bytes;
    p.phase_in_x = 0;
    if (nphase > 1 && g_phase_major) {
        grid = dim3(tiles_m_launch * tiles_n * nphase, split, 1);
        p.phase_in_x = nphase;
    }
    p.xcd_swizzle = g_xcd_swizzle && grid.x >= 16;
    const int pipe = g_igemm_pipe;
    // LDS-DMA loop (PIPE 5) on the tiles wdg_set_tuning("igemm_dma", mask) names: bit 0 = 128x64, 1 = 256x32 / 256x16, 2 = 64x64, 3 = 128x128,
    // 4 = 128x80 / 128x160 / 256x80
    const int dma_bit = (tc.BM == 128 && tc.BN == 64) ? 1 : (tc.BM == 256 && tc.BN <= 32) ? 2 : (tc.BM == 64 && tc.BN == 64) ? 4
                        : (tc.BM == 128 && tc.BN == 128) ? 8 : 16;
    const bool pipe5 = pipe == 3 && (g_igemm_dma & dma_bit) && kg == 1;
    int rc = WDG_OK;
    // fused BatchNorm hooks: only without split-K (the reduce kernel owns the epilogue then; callers fall back to the
    // standalone passes — see conv_fused_bn) and only on the default pipeline
    const bool ln_ok = p.ln_gamma && split == 1 && tiles_n == 1 && nphase == 1 && (p.Ncols & 3) == 0;
    const bool lnb_reduce = p.lnb_y && split > 1;          // split-K: the norm's backward runs in the second stage
    const int epi = lnb_reduce ? 0 : p.lnb_y ? (p.lnb_par ? 5 : 6) : p.lstm_F ? 4 : (p.stats && split == 1) ? 1 : (p.affine && split == 1) ? 2 : ln_ok ? 3 : 0;
    if (epi >= 5 && !(split == 1 && tiles_n == 1 && pipe == 3 && !p.accumulate && !p.bias && !p.act && lnb_tile_ok(tc))) {
        wdg_set_error("igemm: the LayerNorm-backward epilogue needs one column tile of a 4 x 1 wave layout, no split-K, the fp32 pipeline");
        return WDG_ERR_ARG;
    }
    if (epi == 4 && (split != 1 || nphase != 1)) {
        wdg_set_error("igemm: the ConvLSTM step epilogue needs one phase, no split-K and the fp32 pipeline");
        return WDG_ERR_ARG;
    }
    if (bn_fused) *bn_fused = epi != 0;
#define WDG_IGEMM_CASE(BM_, BN_, WM_, WN_)                                                              \
    if (tc.BM == BM_ && tc.BN == BN_) {                                                                 \
        if (epi == 4) {                                                                                 \
            if constexpr (BN_ % 64 == 0 && BM_ <= 128) {                                                \
                if constexpr (BM_ == BN_) {                                                             \
                    if (kg == 2) rc = launch_variant<BM_, BN_, WM_, WN_, 3, 4, 2>(grid, block, st, p);  \
                    else rc = launch_variant<BM_, BN_, WM_, WN_, 3, 4>(grid, block, st, p);             \
                } else rc = launch_variant<BM_, BN_, WM_, WN_, 3, 4>(grid, block, st, p);               \
            } else rc = WDG_ERR_ARG;                                                                    \
        }                                                                                               \
        else if (epi >= 5) {                                                                            \
            if constexpr ((BM_ == 256 && BN_ <= 32) || (BM_ == 128 && BN_ == 64) || (BM_ == 64 && BN_ == 128)) { \
                if (epi == 5) rc = launch_variant<BM_, BN_, 4, 1, 3, 5>(grid, block, st, p);            \
                else rc = launch_variant<BM_, BN_, 4, 1, 3, 6>(grid, block, st, p);                     \
            } else rc = WDG_ERR_ARG;                                                                    \
        }                                                                                               \
        else if (epi == 1) rc = launch_variant<BM_, BN_, WM_, WN_, 3, 1>(grid, block, st, p);           \
        else if (epi == 2) rc = launch_variant<BM_, BN_, WM_, WN_, 3, 2>(grid, block, st, p);           \
        else if (epi == 3) {                                                                            \
            if constexpr (BM_ == 128 && BN_ == 64) {                                                    \
                if (pipe5) rc = launch_variant<128, 64, 4, 1, 5, 3>(grid, block, st, p);                \
                else if (g_ln_wave) rc = launch_variant<128, 64, 4, 1, 3, 3>(grid, block, st, p);       \
                else rc = launch_variant<BM_, BN_, WM_, WN_, 3, 3>(grid, block, st, p);                 \
            } else rc = launch_variant<BM_, BN_, WM_, WN_, 3, 3>(grid, block, st, p);                   \
        }                                                                                               \
        else if (pipe5 && epi == 0) rc = launch_variant<BM_, BN_, WM_, WN_, 5>(grid, block, st, p);     \
        else if (pipe == 0) rc = launch_variant<BM_, BN_, WM_, WN_, 0>(grid, block, st, p);             \
        else if (pipe == 1) rc = launch_variant<BM_, BN_, WM_, WN_, 1>(grid, block, st, p);             \
        else if (pipe == 3) rc = launch_variant<BM_, BN_, WM_, WN_, 3>(grid, block, st, p);             \
        else rc = launch_variant<BM_, BN_, WM_, WN_, 2>(grid, block, st, p);                            \
    }
    if (tc.BM == 64 && tc.BN == 128) {       // (the LayerNorm-backward epilogue of a 128-channel row on a small map: instantiated for it alone)
        if (epi == 5) rc = launch_variant<64, 128, 4, 1, 3, 5>(grid, block, st, p);
        else if (epi == 6) rc = launch_variant<64, 128, 4, 1, 3, 6>(grid, block, st, p);
        else rc = WDG_ERR_ARG;
    }
    WDG_IGEMM_CASE(128, 160, 2, 2)
    WDG_IGEMM_CASE(256, 80, 4, 1)
    WDG_IGEMM_CASE(128, 80, 4, 1)
    WDG_IGEMM_CASE(128, 128, 2, 2)
    WDG_IGEMM_CASE(128, 64, 2, 2)
    WDG_IGEMM_CASE(64, 64, 2, 2)
    WDG_IGEMM_CASE(256, 32, 4, 1)
    WDG_IGEMM_CASE(256, 16, 4, 1)
#undef WDG_IGEMM_CASE
    if (rc != WDG_OK) return rc;
    WDG_LAUNCH_CHECK();
    if (split > 1 && lnb_reduce) {
        const int blocks = (int)std::min<long long>(((long long)p.Mmax + 3) / 4, 2048);
        hipLaunchKernelGGL(wdg_igemm_reduce_lnbwd_kernel, dim3(blocks, 1, nphase), block, 0, st, p);
        WDG_LAUNCH_CHECK();
        return WDG_OK;
    }
    if (split > 1) {
        if (p.ln_gamma && nphase == 1 && (p.Ncols & 3) == 0 && p.Ncols <= 1024 && !p.accumulate) {
            // (the slabs of a 4-aligned column count are dense: NcP == Ncols)
            if (g_reduce_wpr && p.Mmax <= 2048 && split >= 8) {
                hipLaunchKernelGGL(wdg_igemm_reduce_ln_kernel<4>, dim3((unsigned)p.Mmax), block, 0, st, p);
            } else {
                const int blocks = (int)std::min<long long>(((long long)p.Mmax + 3) / 4, 8192);
                hipLaunchKernelGGL(wdg_igemm_reduce_ln_kernel<1>, dim3(blocks), block, 0, st, p);
            }
            WDG_LAUNCH_CHECK();
            if (bn_fused) *bn_fused = true;
            return WDG_OK;
        }
        long long total = (long long)p.Mmax * p.Ncols;
        int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(wdg_igemm_reduce_kernel, dim3(blocks, 1, nphase), block, 0, st, p);
        WDG_LAUNCH_CHECK();
    }
    return WDG_OK;
}

// BatchNorm hooks shared by the forward / data-gradient entry points.  `stats` (training-mode producer): the replica
// slabs [stats_rep][2][stats_C] (fp64) receive the per-channel sum and sum of squares of the written output; `affine`
// (inference mode): y = act(conv + bias) * affine[c] + affine[affine_ld + c].  When the launch cannot carry the hook in
// its epilogue (split-K second stage, halo-tile kernel) the standalone pass runs behind it, so the result is the same.
struct WdgBnHook {
    double* stats;
    int stats_C, stats_rep;
    const float* affine;
    int affine_ld;
};

static int bn_hook_fallback(const WdgBnHook& h, float* y, int ldy, int64_t img_stride, int n_img, int Ho, int Wo, int C,
                            wdg_stream stream) {
    if (img_stride != (int64_t)Ho * Wo * ldy) {
        wdg_set_error("conv + BatchNorm hook: the unfused path needs contiguous output images");
        return WDG_ERR_ARG;
    }
    const int64_t P = (int64_t)n_img * Ho * Wo;
    const int Cp = wdg_round_up(C, 4);      // (pad channels of y are zero: their statistics are zero, their affine is applied to zeros)
    if (h.stats) return wdg_bn_stats(y, P, Cp, ldy, h.stats, stream);
    return wdg_bn_apply(y, ldy, h.affine, y, ldy, P, Cp, stream);
}

static int conv_fwd_impl(const wdg_conv_plan* pl, const float* x, const float* wF, const float* bias, float* y, int act,
                         float slope, int accumulate, const WdgBnHook* hook, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF && y, "null argument");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wF & 15) == 0 && ((uintptr_t)y & 15) == 0, "x / wF / y must be 16-byte aligned");
    const wdg_conv_geom& g = pl->g;
    if (pl->halo_auto_fwd && pl->halo_fwd_nt) {
        const int rc = wdg_halo_launch(pl, false, x, g.ldx, g.img_stride_x, 0, wF, bias, y, act, slope, accumulate,
                                       (hipStream_t)stream);
        if (rc != WDG_OK || !hook) return rc;
        return bn_hook_fallback(*hook, y, g.ldy, g.img_stride_y, g.n_img, g.Ho, g.Wo, g.Cout, stream);
    }
    WdgIgemm p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.B = wF; p.Out = y; p.bias = bias; p.ktab = pl->d_tab_fwd;
    p.imgStrideA = g.img_stride_x; p.imgStrideO = g.img_stride_y;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldA = g.ldx;
    p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy;
    p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p;
    p.a_mul = g.stride; p.o_mul = 1;
    p.act = act; p.slope = slope; p.accumulate = accumulate;
    p.Mmax = g.n_img * g.Ho * g.Wo;
    if (hook) { p.stats = hook->stats; p.stats_C = hook->stats_C; p.stats_rep = hook->stats_rep; p.affine = hook->affine; p.affine_ld = hook->affine_ld; }
    WdgPhase ph;
    ph.Pa = g.Ho; ph.Pb = g.Wo; ph.a_off_h = -g.pad_h; ph.a_off_w = -g.pad_w;
    ph.o_off_h = 0; ph.o_off_w = 0; ph.K4 = pl->K4_fwd; ph.tab_off = 0;
    wdg_phase_finish(ph);
    p.ph[0] = ph;
    bool fused = false;
    const int rc = launch_igemm(p, 1, pl->K4_fwd, pl->fwd_split, ws, ws_bytes, (hipStream_t)stream, &fused);
    if (rc != WDG_OK || !hook || fused) return rc;
    return bn_hook_fallback(*hook, y, g.ldy, g.img_stride_y, g.n_img, g.Ho, g.Wo, g.Cout, stream);
}

static int conv_dgrad_impl(const wdg_conv_plan* pl, const float* dy, const float* wD, const float* bias, float* dx, int act,
                           float slope, int accumulate, const WdgBnHook* hook, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl && dy && wD && dx, "null argument");
    WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)wD & 15) == 0 && ((uintptr_t)dx & 15) == 0, "dy / wD / dx must be 16-byte aligned");
    const wdg_conv_geom& g = pl->g;
    if (pl->halo_auto_dgrad && pl->halo_dgrad_nt) {
        const int rc = wdg_halo_launch(pl, true, dy, g.ldy, g.img_stride_y, 0, wD, bias, dx, act, slope, accumulate,
                                       (hipStream_t)stream);
        if (rc != WDG_OK || !hook) return rc;
        return bn_hook_fallback(*hook, dx, g.ldx, g.img_stride_x, g.n_img, g.H, g.W, g.Cin, stream);
    }
    WdgIgemm p;
    memset(&p, 0, sizeof(p));
    p.A = dy; p.B = wD; p.Out = dx; p.bias = bias; p.ktab = pl->d_tab_dgrad;
    p.imgStrideA = g.img_stride_y; p.imgStrideO = g.img_stride_x;
    p.n_img = g.n_img; p.H = g.Ho; p.W = g.Wo; p.ldA = g.ldy;
    p.Ho = g.H; p.Wo = g.W; p.ldO = g.ldx;
    p.Ncols = g.Cin; p.ldB = pl->w_ld ? pl->w_ld : pl->Cout_p;
    p.a_mul = 1; p.o_mul = g.stride;
    p.act = act; p.slope = slope; p.accumulate = accumulate;
    if (hook) { p.stats = hook->stats; p.stats_C = hook->stats_C; p.stats_rep = hook->stats_rep; p.affine = hook->affine; p.affine_ld = hook->affine_ld; }
    int Mmax = 0;
    const int np = (int)pl->ph_dgrad.size();
    for (int i = 0; i < np; ++i) {
        p.ph[i] = pl->ph_dgrad[i];
        Mmax = std::max(Mmax, g.n_img * p.ph[i].Pa * p.ph[i].Pb);
    }
    p.Mmax = Mmax;
    bool fused = false;
    const int rc = launch_igemm(p, np, pl->K4_dgrad_max, pl->dgrad_split, ws, ws_bytes, (hipStream_t)stream, &fused);
    if (rc != WDG_OK || !hook || fused) return rc;
    return bn_hook_fallback(*hook, dx, g.ldx, g.img_stride_x, g.n_img, g.H, g.W, g.Cin, stream);
}

// conv -> bias -> LeakyReLU -> LayerNormalization (models.py:113-116, 122-125, 134-136; tf_utils.py:29-31) in one call:
// y = act(conv(x) + bias), z = LN(y; gamma, beta, eps) over the channels, mean_rstd[pixel] = (mean, 1/sqrt(var + eps)) for
// the backward pass.  The normalisation runs in the epilogue that owns complete rows — the implicit-GEMM epilogue when one
// tile spans all output channels, the split-K second stage otherwise — and as the standalone wdg_ln_fwd pass behind the
// convolution on every other route; results do not depend on the route.
// ---- ConvLSTM2D recurrent step through the implicit GEMM (the generator's 128-feature layer at n_timesteps > 1): one launch
// per timestep instead of three (accumulating convolution + its split-K second stage + cell kernel).  wF_il: the packed forward
// weights with gate-interleaved rows (row n = gate n & 3 of feature n >> 2), see wdgan.h.
extern "C" int wdg_convlstm_step_gemm_supported(const wdg_conv_plan* pl, int F) {
    if (!pl || F <= 0 || (F & 15)) return 0;
    const wdg_conv_geom& g = pl->g;
    if (g.Cout != 4 * F || g.stride != 1 || g.H != g.Ho || g.W != g.Wo || g.ldy != 4 * F ||
        g.img_stride_y != (int64_t)g.Ho * g.Wo * 4 * F)
        return 0;
    if (pl->halo_auto_fwd && pl->halo_fwd_nt) return 0;           // (the halo-tile kernel owns the thin layers: wdg_convlstm_step)
    const TileCfg tc = pick_tile(g.Cout, true, (long long)g.n_img * g.Ho * g.Wo);
    return tc.BN % 64 == 0 && tc.BM <= 128;
}

extern "C" int wdg_convlstm_step_gemm(const wdg_conv_plan* pl, const float* h_prev, const float* wF_il, float* gates,
                                      const float* c_prev, float* c_out, int ldc, float* h_out, int ldh, int F, wdg_stream stream) {
    WDG_CHECK_ARG(pl && h_prev && wF_il && gates && c_prev && c_out && h_out && wdg_convlstm_step_gemm_supported(pl, F),
                  "not supported for this geometry");
    WDG_CHECK_ARG(ldc >= F && ldh >= F && ((uintptr_t)h_prev & 15) == 0 && ((uintptr_t)wF_il & 15) == 0, "bad strides / alignment");
    const wdg_conv_geom& g = pl->g;
    WdgIgemm p;
    memset(&p, 0, sizeof(p));
    p.A = h_prev; p.B = wF_il; p.Out = gates; p.ktab = pl->d_tab_fwd;
    p.imgStrideA = g.img_stride_x; p.imgStrideO = g.img_stride_y;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldA = g.ldx;
    p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy;
    p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p;
    p.a_mul = 1; p.o_mul = 1;
    p.Mmax = g.n_img * g.Ho * g.Wo;
    p.lstm_F = F; p.ldc = ldc; p.ldh = ldh; p.c_prev = c_prev; p.c_out = c_out; p.h_out = h_out;
    WdgPhase ph;
    ph.Pa = g.Ho; ph.Pb = g.Wo; ph.a_off_h = -g.pad_h; ph.a_off_w = -g.pad_w;
    ph.o_off_h = 0; ph.o_off_w = 0; ph.K4 = pl->K4_fwd; ph.tab_off = 0;
    wdg_phase_finish(ph);
    p.ph[0] = ph;
    return launch_igemm(p, 1, pl->K4_fwd, 1, nullptr, 0, (hipStream_t)stream);
}

extern "C" int wdg_conv_fwd_ln(const wdg_conv_plan* pl, const float* x, const float* wF, const float* bias, float* y, float* z,
                               const float* gamma, const float* beta, float eps, float* mean_rstd, int act, float slope,
                               void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl, "null plan");
    return wdg_conv_fwd_ln_strided(pl, x, wF, bias, y, z, pl->g.ldy, pl->g.img_stride_y, gamma, beta, eps, mean_rstd, act, slope, ws,
                                   ws_bytes, stream);
}

extern "C" int wdg_conv_fwd_ln_strided(const wdg_conv_plan* pl, const float* x, const float* wF, const float* bias, float* y, float* z,
                                       int ldz, int64_t img_stride_z, const float* gamma, const float* beta, float eps,
                                       float* mean_rstd, int act, float slope, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && wF && y && z && gamma && beta, "null argument");
    WDG_CHECK_ARG(ldz % 4 == 0 && ldz >= wdg_round_up(pl->g.Cout, 4), "z: pixel stride must be a multiple of 4 and >= the padded channel count");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wF & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)z & 15) == 0 &&
                  ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0, "x / wF / y / z / gamma / beta must be 16-byte aligned");
    const wdg_conv_geom& g = pl->g;
    const int64_t P = (int64_t)g.n_img * g.Ho * g.Wo;
    bool fused = false;
    const bool same_view = ldz == g.ldy && img_stride_z == g.img_stride_y;
    if (wdg_halo_ln_eligible(pl) && act) {
        // thin full-resolution 16 -> 16 layer (models.py:102-105): the norm in the epilogue of the persistent halo-tile kernel
        WdgHaloLn ln;
        ln.z = z; ln.ldz = ldz; ln.img_stride_z = img_stride_z; ln.gamma = gamma; ln.beta = beta; ln.eps = eps; ln.mean_rstd = mean_rstd;
        return wdg_halo_launch(pl, false, x, g.ldx, g.img_stride_x, 0, wF, bias, y, act, slope, 0, (hipStream_t)stream, nullptr, &ln);
    }
    if (!(pl->halo_auto_fwd && pl->halo_fwd_nt) && g.Cout % 4 == 0 && same_view) {
        WdgIgemm p;
        memset(&p, 0, sizeof(p));
        p.A = x; p.B = wF; p.Out = y; p.bias = bias; p.ktab = pl->d_tab_fwd;
            p.imgStrideA = g.img_stride_x; p.imgStrideO = g.img_stride_y;
        p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldA = g.ldx;
        p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy;
        p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p;
        p.a_mul = g.stride; p.o_mul = 1;
        p.act = act; p.slope = slope; p.accumulate = 0;
        p.Mmax = g.n_img * g.Ho * g.Wo;
        p.Out2 = z; p.ln_gamma = gamma; p.ln_beta = beta; p.ln_eps = eps; p.mean_rstd = mean_rstd;
        WdgPhase ph;
        ph.Pa = g.Ho; ph.Pb = g.Wo; ph.a_off_h = -g.pad_h; ph.a_off_w = -g.pad_w;
        ph.o_off_h = 0; ph.o_off_w = 0; ph.K4 = pl->K4_fwd; ph.tab_off = 0;
        wdg_phase_finish(ph);
        p.ph[0] = ph;
        const int rc = launch_igemm(p, 1, pl->K4_fwd, pl->fwd_split, ws, ws_bytes, (hipStream_t)stream, &fused);
        if (rc != WDG_OK || fused) return rc;
    } else {
        const int rc = conv_fwd_impl(pl, x, wF, bias, y, act, slope, 0, nullptr, ws, ws_bytes, stream);
        if (rc != WDG_OK) return rc;
    }
    if (g.img_stride_y != (int64_t)g.Ho * g.Wo * g.ldy || img_stride_z != (int64_t)g.Ho * g.Wo * ldz) {
        wdg_set_error("wdg_conv_fwd_ln: the unfused path needs contiguous output images");
        return WDG_ERR_ARG;
    }
    return wdg_ln_fwd(y, g.ldy, gamma, beta, eps, z, ldz, mean_rstd, P, g.Cout, stream);
}

extern "C" int wdg_conv_fwd(const wdg_conv_plan* pl, const float* x, const float* wF, const float* bias,
                            float* y, int act, float slope, int accumulate, void* ws, size_t ws_bytes,
                            wdg_stream stream) {
    return conv_fwd_impl(pl, x, wF, bias, y, act, slope, accumulate, nullptr, ws, ws_bytes, stream);
}

extern "C" int wdg_conv_dgrad(const wdg_conv_plan* pl, const float* dy, const float* wD, const float* bias,
                              float* dx, int act, float slope, int accumulate, void* ws, size_t ws_bytes,
                              wdg_stream stream) {
    return conv_dgrad_impl(pl, dy, wD, bias, dx, act, slope, accumulate, nullptr, ws, ws_bytes, stream);
}


// which way a wdg_conv_dgrad_lnbwd call goes: 0 data gradient + wdg_ln_bwd, 1 the implicit-GEMM epilogue (tc_force: its tile when
// not the default one), 2 the patch kernel of dgrad_patch_s3.hip (ws_bytes: the caller's scratch; 0 rules it out)
static int dgrad_lnbwd_route(const wdg_conv_plan* pl, int c0, int C, int ldy_act, bool /*with_par*/, bool par_ok, size_t ws_bytes, TileCfg* tc_force) {
    const wdg_conv_geom& g = pl->g;
    if (g_dgrad_lnbwd && ws_bytes && wdg_dgrad_s3_ok(pl, c0, C, ldy_act) && ws_bytes >= wdg_dgrad_s3_ws_bytes(pl)) return 2;
    bool fuse = g_dgrad_lnbwd && !(pl->halo_auto_dgrad && pl->halo_dgrad_nt) && g_igemm_pipe == 3 && par_ok;
    if (fuse && pl->dgrad_split > 1) {
        fuse = g.Cin % 4 == 0 && g.Cin <= 1024;                        // second stage: wdg_igemm_reduce_lnbwd_kernel
    } else if (fuse) {
        long long Mmax = 0;
        for (auto& ph : pl->ph_dgrad) Mmax = std::max(Mmax, (long long)g.n_img * ph.Pa * ph.Pb);
        TileCfg tc = pick_tile(g.Cin, true, Mmax);
        if (tc.BM == 64 && tc.BN == 64 && g.Cin == 128 && (g_dgrad_lnbwd & 2)) tc = *tc_force = TileCfg{64, 128};   // a 128-channel row in one wave
        fuse = lnb_tile_ok(tc) && g.Cin <= tc.BN;
    }
    return fuse ? 1 : 0;
}

extern "C" int wdg_conv_dgrad_lnbwd_route(const wdg_conv_plan* pl, int c0, int C, int ldy_act, int with_param_grads, size_t ws_bytes) {
    if (!pl) return -1;
    TileCfg tc = {0, 0};
    return dgrad_lnbwd_route(pl, c0, C, ldy_act, with_param_grads != 0, true, ws_bytes, &tc);
}

// Data gradient of a convolution whose INPUT tensor (channels [c0, c0 + C) of it) was produced by conv -> bias -> LeakyReLU -> LayerNormalization:
// dx = dgrad(dy), then the LayerNorm + LeakyReLU backward applied to those channels of dx IN PLACE (dx[..., c0:c0+C] becomes the gradient
// w.r.t. the producer's pre-activation), with dgamma / dbeta / dbias accumulated when given.  One launch where an implicit-GEMM tile owns
// complete pixels (the norm's two reductions run on the accumulators), dgrad + wdg_ln_bwd otherwise: the results do not depend on the route.
extern "C" int wdg_conv_dgrad_lnbwd(const wdg_conv_plan* pl, const float* dy, const float* wD, float* dx, const float* y, int ldy_act,
                                    int64_t img_stride_act, const float* mean_rstd, const float* gamma, int c0, int C, float act_slope,
                                    float* dgamma, float* dbeta, float* dbias, float* par_ws, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl && dy && wD && dx && y && mean_rstd && gamma, "null argument");
    const wdg_conv_geom& g = pl->g;
    WDG_CHECK_ARG(c0 >= 0 && C > 0 && c0 % 4 == 0 && C % 4 == 0 && c0 + C <= g.Cin, "bad channel group");
    WDG_CHECK_ARG(((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && ldy_act % 4 == 0 && ldy_act >= C, "y / gamma alignment");
    const bool want_par = dgamma || dbeta || dbias;
    TileCfg tc_force = {0, 0};
    const int route = dgrad_lnbwd_route(pl, c0, C, ldy_act, want_par && par_ws, !want_par || par_ws, ws_bytes, &tc_force);
    if (route == 2 && ws && ((uintptr_t)ws & 15) == 0) {
        // 7 x 7 stride-3 32 -> 64: the dy patch of a 24 x 24 block of dx pixels in LDS, all nine residue classes in one workgroup (dgrad_patch_s3.hip)
        WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)wD & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)mean_rstd & 7) == 0,
                      "dy / wD / dx must be 16-byte aligned");
        return wdg_dgrad_s3_launch(pl, dy, wD, dx, y, ldy_act, img_stride_act, mean_rstd, gamma, c0, C, act_slope, dgamma, dbeta, dbias, ws,
                                   (hipStream_t)stream);
    }
    const bool fuse = route >= 1 && (route == 1 || dgrad_lnbwd_route(pl, c0, C, ldy_act, false, !want_par || par_ws, 0, &tc_force) == 1);
    if (!fuse) {
        int rc = conv_dgrad_impl(pl, dy, wD, nullptr, dx, 0, 0.f, 0, nullptr, ws, ws_bytes, stream);
        if (rc != WDG_OK) return rc;
        if (g.img_stride_x != (int64_t)g.H * g.W * g.ldx || img_stride_act != (int64_t)g.H * g.W * ldy_act) {
            wdg_set_error("wdg_conv_dgrad_lnbwd: the unfused path needs contiguous images");
            return WDG_ERR_ARG;
        }
        return wdg_ln_bwd(dx + c0, g.ldx, y, ldy_act, mean_rstd, gamma, act_slope, dx + c0, g.ldx, dgamma, dbeta, dbias,
                          (int64_t)g.n_img * g.H * g.W, C, stream);
    }
    WDG_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)wD & 15) == 0 && ((uintptr_t)dx & 15) == 0, "dy / wD / dx must be 16-byte aligned");
    WdgIgemm p;
    memset(&p, 0, sizeof(p));
    p.A = dy; p.B = wD; p.Out = dx; p.ktab = pl->d_tab_dgrad;
    p.imgStrideA = g.img_stride_y; p.imgStrideO = g.img_stride_x;
    p.n_img = g.n_img; p.H = g.Ho; p.W = g.Wo; p.ldA = g.ldy;
    p.Ho = g.H; p.Wo = g.W; p.ldO = g.ldx;
    p.Ncols = g.Cin; p.ldB = pl->w_ld ? pl->w_ld : pl->Cout_p;
    p.a_mul = 1; p.o_mul = g.stride;
    p.lnb_y = y; p.lnb_ldy = ldy_act; p.lnb_imgStride = img_stride_act; p.lnb_stats = mean_rstd; p.lnb_gamma = gamma;
    p.lnb_c0 = c0; p.lnb_C = C; p.lnb_slope = act_slope; p.lnb_par = want_par ? par_ws : nullptr; p.lnb_rep = WDG_LNB_REP;
    int Mmax = 0;
    const int np = (int)pl->ph_dgrad.size();
    for (int i = 0; i < np; ++i) {
        p.ph[i] = pl->ph_dgrad[i];
        Mmax = std::max(Mmax, g.n_img * p.ph[i].Pa * p.ph[i].Pb);
    }
    p.Mmax = Mmax;
    const int rc = launch_igemm(p, np, pl->K4_dgrad_max, pl->dgrad_split, ws, ws_bytes, (hipStream_t)stream, nullptr, tc_force.BM ? &tc_force : nullptr);
    if (rc != WDG_OK || !want_par) return rc;
    return wdg_lnb_finish(par_ws, WDG_LNB_REP, C, dgamma, dbeta, dbias, (hipStream_t)stream);
}

// floats of the zero-initialised scratch `par_ws` of wdg_conv_dgrad_lnbwd for a group of C channels (the call leaves it zeroed)
extern "C" int64_t wdg_conv_dgrad_lnbwd_par_floats(int C) { return (int64_t)WDG_LNB_REP * 3 * C; }

static int bn_hook_make(WdgBnHook& h, double* stats, int stats_rep, const float* affine, int C) {
    WDG_CHECK_ARG((stats != nullptr) != (affine != nullptr), "exactly one of stats / affine");
    WDG_CHECK_ARG(!stats || stats_rep >= 1, "stats_rep must be >= 1");
    const int Cp = wdg_round_up(C, 4);      // per-channel vectors of the hooks are laid out at the padded channel count
    h.stats = stats; h.stats_C = Cp; h.stats_rep = stats ? stats_rep : 0; h.affine = affine; h.affine_ld = Cp;
    return WDG_OK;
}

extern "C" int wdg_conv_fwd_bn(const wdg_conv_plan* pl, const float* x, const float* wF, const float* bias, float* y,
                               int act, float slope, double* stats, int stats_rep, const float* affine, void* ws,
                               size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl, "null plan");
    WdgBnHook h;
    const int rc = bn_hook_make(h, stats, stats_rep, affine, pl->g.Cout);
    if (rc != WDG_OK) return rc;
    return conv_fwd_impl(pl, x, wF, bias, y, act, slope, 0, &h, ws, ws_bytes, stream);
}

extern "C" int wdg_conv_dgrad_bn(const wdg_conv_plan* pl, const float* dy, const float* wD, const float* bias, float* dx,
                                 int act, float slope, double* stats, int stats_rep, const float* affine, void* ws,
                                 size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl, "null plan");
    WdgBnHook h;
    const int rc = bn_hook_make(h, stats, stats_rep, affine, pl->g.Cin);
    if (rc != WDG_OK) return rc;
    return conv_dgrad_impl(pl, dy, wD, bias, dx, act, slope, 0, &h, ws, ws_bytes, stream);
}

int wdg_colsum(const float* x, int ldx, int64_t P, int C, float* out, int accumulate, wdg_stream stream);

// weight gradient + bias gradient (dbias[co] += sum_pixels dy[pixel][co]) in one call: the thin 3x3 kernel gets
// the column sums as a by-product of a constant-1 row; every other geometry runs the column-sum pass after it.
extern "C" int wdg_conv_wgrad_bias(const wdg_conv_plan* pl, const float* x, const float* dy, float* dw, float* dbias,
                                   int accumulate, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && dy && dw, "null argument");
    if (dbias && wdg_wgrad_thin_has_bias_row(pl)) {
        WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "x / dy must be 16-byte aligned");
        return wdg_wgrad_thin_launch(pl, x, dy, dw, dbias, accumulate, ws, ws_bytes, (hipStream_t)stream);
    }
    const int rc = wdg_conv_wgrad(pl, x, dy, dw, accumulate, ws, ws_bytes, stream);
    if (rc != WDG_OK || !dbias) return rc;
    const wdg_conv_geom& g = pl->g;
    if (g.img_stride_y != (int64_t)g.Ho * g.Wo * g.ldy) {
        wdg_set_error("wdg_conv_wgrad_bias: dy images must be contiguous for the column-sum pass");
        return WDG_ERR_ARG;
    }
    return wdg_colsum(dy, g.ldy, (int64_t)g.n_img * g.Ho * g.Wo, g.Cout, dbias, 1, stream);
}

extern "C" int wdg_conv_wgrad(const wdg_conv_plan* pl, const float* x, const float* dy, float* dw,
                              int accumulate, void* ws, size_t ws_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(pl && x && dy && dw, "null argument");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "x / dy must be 16-byte aligned");
    const wdg_conv_geom& g = pl->g;
    hipStream_t st = (hipStream_t)stream;
    if (wdg_wgrad_thin_eligible(pl)) return wdg_wgrad_thin_launch(pl, x, dy, dw, nullptr, accumulate, ws, ws_bytes, st);
    if (wdg_wgrad_halo_eligible(pl)) return wdg_wgrad_halo_launch(pl, x, dy, dw, accumulate, ws, ws_bytes, st);
    WdgWgrad p;
    memset(&p, 0, sizeof(p));
    p.X = x; p.DY = dy; p.dW = dw; p.ktab = pl->d_tab_fwd; p.wrow = pl->d_wrow;
    p.imgStrideX = g.img_stride_x; p.imgStrideY = g.img_stride_y;
    p.n_img = g.n_img; p.H = g.H; p.W = g.W; p.ldx = g.ldx;
    p.Ho = g.Ho; p.Wo = g.Wo; p.ldy = g.ldy;
    p.stride = g.stride; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
    p.K4 = pl->K4_fwd; p.Cout = g.Cout; p.Cout_p = pl->Cout_p;
    p.w_ld = pl->w_ld ? pl->w_ld : g.Cout;
    p.accumulate = accumulate;
    p.Ptot = (long long)g.n_img * g.Ho * g.Wo;
    int split = pl->wgrad_split;
    long long per = (p.Ptot + split - 1) / split;
    per = (per + 31) / 32 * 32;
    split = (int)((p.Ptot + per - 1) / per);
    p.splitk = split;
    p.pix_per_split = per;
    p.div_howo = wdg_fastdiv_make((unsigned)(g.Ho * g.Wo));
    p.div_wo = wdg_fastdiv_make((unsigned)g.Wo);
    {
        // 32-bit buffer offsets are relative to the first image of a split
        const long long imgs = per / ((long long)g.Ho * g.Wo) + 2;
        if (imgs * std::max(g.img_stride_x, g.img_stride_y) * 4 >= (1LL << 31) || per + (long long)g.Ho * g.Wo >= (1LL << 31)) {
            wdg_set_error("wgrad: a pixel split spans more than 2 GiB of x or dy");
            return WDG_ERR_ARG;
        }
    }
    if (split > 1) {
        const size_t need = (size_t)split * p.K4 * 4 * p.Cout * sizeof(float);
        if (!ws || ws_bytes < need) {
            wdg_set_error("wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
            return WDG_ERR_WORKSPACE;
        }
        p.partial = (float*)ws;
    }
    const int bn = pick_wgrad_bn(g.Cout);
    const int tiles_m = (p.K4 * 4 + 127) / 128;
    const int tiles_n = (g.Cout + bn - 1) / bn;
    dim3 grid(tiles_m * tiles_n, split, 1), block(256);
    if (g_wgrad_xcd && split >= 8 && tiles_m * tiles_n > 1 && tiles_m * tiles_n <= g_wgrad_xcd) {
        p.xcd_tiles = tiles_m * tiles_n;
        grid = dim3(tiles_m * tiles_n * split, 1, 1);
    }
    if (bn == 128)
        hipLaunchKernelGGL((wdg_wgrad_kernel<128, 2, 2>), grid, block, 0, st, p);
    else if (bn == 64)
        hipLaunchKernelGGL((wdg_wgrad_kernel<64, 2, 2>), grid, block, 0, st, p);
    else if (bn == 32)
        hipLaunchKernelGGL((wdg_wgrad_kernel<32, 4, 1>), grid, block, 0, st, p);
    else
        hipLaunchKernelGGL((wdg_wgrad_kernel<16, 4, 1>), grid, block, 0, st, p);
    WDG_LAUNCH_CHECK();
    if (split > 1) {
        long long total = (long long)p.K4 * 4 * p.Cout;
        if (g_wgrad_reduce4 && (p.Cout & 3) == 0 && ((uintptr_t)p.partial & 15) == 0 && ((uintptr_t)dw & 15) == 0) {
            int blocks = (int)std::min<long long>((total / 4 + 63) / 64, 8192);
            hipLaunchKernelGGL(wdg_wgrad_reduce4_kernel, dim3(blocks), block, 0, st, p);
        } else {
            int blocks = (int)std::min<long long>((total + 15) / 16, 8192);
            hipLaunchKernelGGL(wdg_wgrad_reduce_kernel, dim3(blocks), block, 0, st, p);
        }
        WDG_LAUNCH_CHECK();
    }
    return WDG_OK;
}

extern "C" int wdg_weight_pack(const float* w_hwio, float* wF, float* wD, int taps, int Cin, int Cout,
                               wdg_stream stream) {
    WDG_CHECK_ARG(w_hwio && wF, "null argument");
    const int Cin_p = wdg_round_up(Cin, 4), Cout_p = wdg_round_up(Cout, 4);
    long long total = (long long)Cout * taps * Cin_p + (wD ? (long long)taps * Cin * Cout_p : 0);
    int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(wdg_weight_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, wF,
                       wD, taps, Cin, Cout, Cin_p, Cout_p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}
