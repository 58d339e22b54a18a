// dgrad_patch_s3.hip — data gradient of the discriminator's 7 x 7 stride-3 convolution with 32 input and 64 output channels, with
// the LayerNormalization + LeakyReLU backward of the layer below applied to the accumulators (the one-launch form of
// wdg_conv_dgrad_lnbwd, conv_igemm.hip).  gfx950 only, fp32 MFMA 16x16x4.
//
// Reference: the gradient tf.GradientTape takes through Conv2D(64, (7, 7), strides=3) -> LayerNormalization -> LeakyReLU of
// /root/reference/src/gan/models.py:79-97 (discriminator image / time branches); oracle: oracle/torch_model.py.
//
// Why its own kernel.  The implicit-GEMM route runs the nine residue classes (iy + pad) mod 3 x (ix + pad) mod 3 of a stride-3 data
// gradient as nine row ranges of one launch: every 256-pixel row tile stages its dy operand AND the weights of its taps through LDS
// behind a barrier per 32-deep K-step; at 32 columns the tile holds 2 MFMAs per staged fragment and measured 0.55-0.58 of the fp32
// peak (skeleton with loads and staging removed: 0.755).  Here a workgroup owns an 8 x 8 block of BASE positions
// b = (i + pad) div 3 — a 24 x 24 block of dx pixels, all nine classes — and
//   * stages the 10 x 10 x 64 dy patch those pixels reduce over ONCE (27 KB of LDS, one barrier per workgroup);
//   * gives each wave whole classes (13 / 12 / 12 / 12 of the 49 taps; the role rotates with a hash of the tile number): a tap's
//     32 x 64 weights are read once per workgroup, from a copy in fragment order (one 1 KB line-contiguous request per fragment,
//     rebuilt per launch by a 5 us kernel), straight into the A fragments, and used by the 4 pixel fragments of the class —
//     128 MFMAs per 8 weight requests and 16 LDS reads, no barrier in the reduction;
//   * owns complete pixels (all 32 channels of a pixel in one wave), so the norm's two reductions are in-lane sums + two lane
//     swaps, the arithmetic of epilogues 5 / 6 of the implicit GEMM; the norm's input and statistics reach the epilogue through
//     LDS by DMA, requested a whole class ahead.
// Algorithmic work of a launch: 2 * 49 * 32 * 64 flops per dy pixel; tile padding (bases 86 -> 88 per axis on the 256 x 256 map) 4.7 %.
//
// Measured (32 images of 256 x 256, clocks warm, profiles/r06ae_dgrad_s3.txt): input-gradient-only launch 445-447 us = 0.645 of the
// fp32 peak (implicit GEMM 497 = 0.58), with parameter gradients 461 incl. the pack and finish kernels (521); whole training step
// -0.65 .. -1.1 ms on three boxes (profiles/r06ae_ab_tune_dgrad_s3.txt).  Where the rest goes, from the skeletons and in-kernel clocks:
//   * MFMAs alone (no weights, no epilogue, no DMA): 373 us = 0.77 — the practical ceiling of this reduction shape (no fp32 MFMA
//     kernel of this library exceeds 0.81 of the nominal peak); without epilogue and DMA 400, without weight requests 414;
//   * a wave's serial phases — prologue (patch fill + barrier) 16 % of its life, epilogue 10 %, DMA requests 7 % — are covered by ONE
//     other wave per SIMD, which alone issues an MFMA every ~49 clocks (clocks of the one-workgroup-per-CU run), not every 32-40;
//   * three workgroups per CU are SLOWER than two (472 vs 444 us; one: 480): the weight stream (401 KB per workgroup, every line
//     an L1 miss) queues up — launch-to-first-MFMA 114 k clocks against 38 k and 20 k.  The launch pads its LDS request to two per CU.
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "conv_plan.h"

#ifndef WDG_S3_PROF
#define WDG_S3_PROF 0                 // measurement builds: shader-clock totals per phase and wave, summed over the launch (wdg_s3_prof)
#endif
#if WDG_S3_PROF
__device__ unsigned long long s3_prof[8];
#define S3_MARK(k) do { __builtin_amdgcn_s_waitcnt(0xc07f); const long long now_ = (long long)__builtin_amdgcn_s_memtime(); pp[k] += now_ - pp_last; pp_last = now_; } while (0)
#else
#define S3_MARK(k) do {} while (0)
#endif

namespace {

int g_dgrad_s3 = 1;

struct WdgDgS3 {
    const float* dy;
    const float* wD;       // master data-gradient weights [tap][ci][ldB]
    const f32x4* wS;       // the same in fragment order (wdg_dgrad_s3_pack_kernel)
    float* dx;
    const float* y;        // the producer's LayerNorm input (pre-norm, post-activation), [img][H][W][ldY]
    const float* stats;    // {mean, rstd} per dx pixel
    const float* gamma;
    float* par;            // per-workgroup sums [ntiles][3][32] (dgamma, dbeta, dbias shares by dx channel) or null
    long long imgStrideDy, imgStrideDx, imgStrideY;
    int ldDy, ldDx, ldY, ldB;
    int H, W, Ho, Wo, pad_h, pad_w;
    int bmin_h, bmin_w, tiles_w, tiles_img, ntiles;
    int rep;
    int c0, C;             // the LayerNorm group: channels [c0, c0 + C) of dx (multiples of 4); the others leave as the plain data gradient
    float slope;
};

constexpr int PW = 10;         // patch pixels per row / column: 8 bases + 2 more rows of dy above / left (taps j = 0..2 read base - j)
constexpr int PPITCH = 17;     // f32x4 slots per patch pixel: 16 of data + 1 (a fragment's 16 pixels then spread over the banks)
constexpr int PATCH_SLOTS = PW * PW * PPITCH;

// classes (ry * 3 + rx) by wave role, one per nibble from the low end, 0xF ends the list: {0, 4}, {1, 2}, {3, 6}, {5, 7, 8}.
// Taps per class = (ry ? 2 : 3) * (rx ? 2 : 3): 13 / 12 / 12 / 12 per role.  (Constants, not a table in memory: a table lookup per
// class is a vector load + s_waitcnt vmcnt(0), which also waits for the previous class's stores.)
// sum over the four 16-lane rows of a wave (a pixel's four channel quads), result in every lane: two lane swaps + two adds on
// the VALU (v_permlane16_swap: odd rows of a <-> even rows of b; v_permlane32_swap: upper half of a <-> lower half of b) instead
// of two ds_bpermute round trips through LDS
__device__ __forceinline__ float wdg_s3_quarter_sum(float v) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // (the two results go through scalar copies: __builtin_bit_cast applied to the vector ELEMENT r[1] reads element 0 with this
    // compiler — clang 22 / ROCm 7.2 —, probed in isolation)
    unsigned u = __builtin_bit_cast(unsigned, v);
    u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned a = r[0], b = r[1];
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    u = __builtin_bit_cast(unsigned, v);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    a = r[0];
    b = r[1];
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ unsigned wdg_s3_role_classes(int role) { return role == 0 ? 0xFF40u : role == 1 ? 0xFF21u : role == 2 ? 0xFF63u : 0xF875u; }

// Weights in fragment order: chunk ((tap * 2 + bt) * 4 + g) * 64 + lane holds W[tap][ci = 16 bt + (lane & 15)][co = 16 g + 4 (lane >> 4) .. + 3],
// so a wave's A-fragment request is 1 KB of consecutive bytes (eight full lines).  Read from the [tap][ci][co] layout directly the
// same request touches 16 lines for 64 bytes each, and the 8 requests per wave and tap cost the kernel 86 us of its 505
// (profiles/r06ae_dgrad_s3.txt).  401 KB per launch, rebuilt by every call: the weights change every step.
__global__ void __launch_bounds__(256) wdg_dgrad_s3_pack_kernel(const float* __restrict__ wD, int ldB, f32x4* __restrict__ wS) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 49 * 2 * 4 * 64) return;
    const int lane = idx & 63, g = (idx >> 6) & 3, bt = (idx >> 8) & 1, tap = idx >> 9;
    wS[idx] = *reinterpret_cast<const f32x4*>(wD + (long long)(tap * 32 + 16 * bt + (lane & 15)) * ldB + 16 * g + 4 * (lane >> 4));
}

// [rows][3][32] (one row per workgroup of the launch) -> dgamma / dbeta / dbias of the group's C channels (accumulated; any may be null): one block of 1024 threads per
// (quantity, channel), four independent partial sums per thread — the rows are 384 bytes apart, every load its own round trip, and a
// serial walk (256 threads x 15 dependent loads) took 3 x as long
__global__ void __launch_bounds__(1024) wdg_dgrad_s3_finish_kernel(const float* __restrict__ part, int ntiles /* rows */, int c0, int C, float* dgamma,
                                                                   float* dbeta, float* dbias) {
    __shared__ float red[16];
    const int which = blockIdx.x / C, c = blockIdx.x - which * C;
    float* dst = which == 0 ? dgamma : which == 1 ? dbeta : dbias;
    if (!dst) return;
    const float* src = part + which * 32 + c0 + c;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    for (int i = threadIdx.x; i < ntiles; i += 4096) {
        v0 += src[(size_t)i * 96];
        if (i + 1024 < ntiles) v1 += src[(size_t)(i + 1024) * 96];
        if (i + 2048 < ntiles) v2 += src[(size_t)(i + 2048) * 96];
        if (i + 3072 < ntiles) v3 += src[(size_t)(i + 3072) * 96];
    }
    const float v = wdg_wave_sum((v0 + v1) + (v2 + v3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        dst[c] += t;
    }
}

#ifndef WDG_S3_PAIR
#define WDG_S3_PAIR 0                 // experiment (-DWDG_S3_PAIR=1): two tiles per workgroup of 8 waves, the waves of one role paired on a
                                      // SIMD so that their weight requests meet in L1 — correct, and SLOWER: 513 us against 449
                                      // (profiles/r06ae_dgrad_s3.txt)
#endif
template <bool PAR, int DBG = 0>
__global__ void __launch_bounds__(WDG_S3_PAIR ? 512 : 256, WDG_S3_PAIR ? 1 : 2) wdg_dgrad_s3_kernel(const WdgDgS3 p) {
#if WDG_S3_PROF
    long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pp_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    extern __shared__ f32x4 lds_all[];
#if WDG_S3_PAIR
    // two tiles per workgroup: waves w and w + 4 take the same role on the same SIMD and request the same weights within a step of
    // each other — the second request meets the first in the CU's L1 (or its miss queue) instead of going to L2 again
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    f32x4* const lds = lds_all + sub * (PATCH_SLOTS + (4 * 3 * 32 + 4 * (64 * p.C + 128)) / 4);
    const int t = threadIdx.x & 255;
    const int tile2 = 2 * wdg_xcd_remap(blockIdx.x, gridDim.x) + sub;
    const int tile = tile2 < p.ntiles ? tile2 : p.ntiles - 1;          // (an odd tile count: the last half repeats its neighbour's work, same values)
#else
    f32x4* const lds = lds_all;
    const int t = threadIdx.x;
    const int tile = wdg_xcd_remap(blockIdx.x, gridDim.x);
#endif
    float* const red = reinterpret_cast<float*>(lds + PATCH_SLOTS);      // [4 waves][3][32], then the waves' norm-input buffers
    const int lane = t & 63, wv = t >> 6, li = lane & 15, lq = lane >> 4;
    const int img = tile / p.tiles_img;
    const int trem = tile - img * p.tiles_img;
    const int ty = trem / p.tiles_w, tx = trem - ty * p.tiles_w;
    const int b0h = p.bmin_h + 8 * ty, b0w = p.bmin_w + 8 * tx;

    // this lane's two channel quads n = 16 bt + 4 lq .. + 3: inside the norm's group?  (gamma = 0 outside: no share in the reductions;
    // loaded in front of the barrier, whose vmcnt(0) then covers it: behind it the compiler would put that wait in front of the loop)
    const int nq0 = 4 * lq - p.c0, nq1 = 16 + 4 * lq - p.c0;
    const bool in0 = nq0 >= 0 && nq0 < p.C, in1 = nq1 >= 0 && nq1 < p.C;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 gm0 = in0 ? *reinterpret_cast<const f32x4*>(p.gamma + nq0) : zero4, gm1 = in1 ? *reinterpret_cast<const f32x4*>(p.gamma + nq1) : zero4;
    const float invC = 1.f / (float)p.C;

    // Wave role, decorrelated over the workgroups that share a CU (their tile numbers differ by multiples of the CU count: a plain
    // (wave + tile) rotation hands one SIMD the same role — the same 13-tap share and the same epilogue times — from all of them)
#if WDG_S3_PAIR
    const int role = __builtin_amdgcn_readfirstlane((wv + __builtin_popcount((unsigned)tile2 >> 1)) & 3);    // (the same for both halves)
#else
    const int role = __builtin_amdgcn_readfirstlane((wv + __builtin_popcount((unsigned)tile)) & 3);
#endif
    // weights in fragment order: lane (li, lq) of A tile bt holds ci = 16 bt + li, co = 16 g + 4 lq .. + 3
    const f32x4* const wl = p.wS + lane;
    auto wtap_ptr = [&](int cls, int tt) -> const f32x4* {
        const int ry = cls / 3, rx = cls - 3 * ry, ntx = rx ? 2 : 3;
        const int jy = tt / ntx, jx = tt - jy * ntx;
        return wl + ((ry + 3 * jy) * 7 + rx + 3 * jx) * 512;
    };
#ifndef WDG_S3_WPOL
#define WDG_S3_WPOL ""                // cache-policy modifiers of the weight requests (experiment: " nt", " sc1", " sc0 sc1")
#endif
#define WDG_S3_LOAD(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF WDG_S3_WPOL : "=v"(dst) : "v"(ptr) : "memory")
    // the wave's first tap: requested here, in front of the patch fill, waited for behind the barrier
    f32x4 af[2][4];
    const unsigned role_cls = wdg_s3_role_classes(role);
    int cls = (int)(role_cls & 15u);
    const wdg_srd srdO = wdg_make_srd(p.dx + (long long)img * p.imgStrideDx);
    {
        const f32x4* wp = wtap_ptr(cls, 0);
        const f32x4* wp1 = wp + 256;
        WDG_S3_LOAD(af[0][0], wp, 0);    WDG_S3_LOAD(af[1][0], wp1, 0);
        WDG_S3_LOAD(af[0][1], wp, 1024); WDG_S3_LOAD(af[1][1], wp1, 1024);
        WDG_S3_LOAD(af[0][2], wp, 2048); WDG_S3_LOAD(af[1][2], wp1, 2048);
        WDG_S3_LOAD(af[0][3], wp, 3072); WDG_S3_LOAD(af[1][3], wp1, 3072);
    }

    // ---- the dy patch: rows b0h - 2 .. b0h + 7, 64 channels = 16 slots per pixel, zero outside the map
    {
        // (all of a thread's requests first, then the LDS writes: a rolled loop waits for each request in turn — seven round trips)
        // (buffer loads: a pixel outside the map is an offset beyond the descriptor and reads zeros — no branches around the requests)
        const wdg_srd srdDy = wdg_make_srd(p.dy + (long long)img * p.imgStrideDy);
        constexpr int NP = (PW * PW * 16 + 255) / 256;
        f32x4 v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = t + 256 * i, px = e >> 4, kq = e & 15;
            const int py = px / PW, pxx = px - py * PW;
            const int oy = b0h - 2 + py, ox = b0w - 2 + pxx;
            const bool ok = e < PW * PW * 16 && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
            v[i] = wdg_buffer_load_f32x4(srdDy, ok ? ((unsigned)(oy * p.Wo + ox) * (unsigned)p.ldDy + 4u * kq) << 2 : WDG_SRD_OOB);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = t + 256 * i;
            if (e < PW * PW * 16) lds[(e >> 4) * PPITCH + (e & 15)] = v[i];
        }
    }
    __syncthreads();
    S3_MARK(0);                                      // index arithmetic, dy patch into LDS, barrier

    // this lane's pixel of fragment 0 at tap (0, 0): base (li >> 3, li & 7) -> patch (.. + 2, .. + 2)
    const int lh = li >> 3, lx = li & 7;
    const f32x4* const pbase = lds + ((lh + 2) * PW + lx + 2) * PPITCH + lq;

    // The norm's input rows and statistics of a class's 64 pixels travel global -> LDS by DMA (no registers) at the START of the
    // class's reduction and are read in its epilogue, 8-13 taps later: the epilogue meets no memory latency.  Per wave:
    // ybuf [64 pixels][C] floats (the DMA's destination is lane-linear: lane l of request i fills 16-byte chunk 64 i + l), then
    // mean [64], rstd [64].  Pixels outside the map: offset beyond the descriptor, the DMA writes zeros.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int wv_u = __builtin_amdgcn_readfirstlane(wv);
    const int ywave = 64 * p.C + 128;                                     // floats per wave
    float* const ybuf = red + 4 * 3 * 32 + wv_u * ywave;
    float* const sbuf = ybuf + 64 * p.C;
    const wdg_srd srdY = wdg_make_srd(p.y + (long long)img * p.imgStrideY);
    const wdg_srd srdS = wdg_make_srd(p.stats + 2ll * img * p.H * p.W);
    const int cq = p.C >> 2;                                              // 16-byte chunks per pixel
    auto pixel_of = [&](int pixel, int ry, int rx, bool& ok) -> int {     // pixel = 16 f + li of the class -> linear dx pixel
        const int f = pixel >> 4, l16 = pixel & 15;
        const int iy = 3 * (b0h + 2 * f + (l16 >> 3)) + ry - p.pad_h, ix = 3 * (b0w + (l16 & 7)) + rx - p.pad_w;
        ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        return iy * p.W + ix;
    };
    auto issue_y = [&](int ry, int rx) {
        for (int i = 0; i < cq; ++i) {
            const int k = 64 * i + lane, pixel = k / cq, part = k - pixel * cq;
            bool ok;
            const int pix = pixel_of(pixel, ry, rx, ok);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdY, (lds_ptr_t)(ybuf + 256 * i), 16,
                                                     ok ? (int)(((unsigned)pix * (unsigned)p.ldY + 4u * part) << 2) : (int)WDG_SRD_OOB, 0, 0, 0);
        }
        bool ok;
        const int pix = pixel_of(lane, ry, rx, ok);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdS, (lds_ptr_t)sbuf, 4, ok ? (int)((unsigned)pix << 3) : (int)WDG_SRD_OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdS, (lds_ptr_t)(sbuf + 64), 4, ok ? (int)(((unsigned)pix << 3) + 4u) : (int)WDG_SRD_OOB, 0, 0, 0);
    };

    // this lane's share of the parameter gradients — dgamma, dbeta, dbias per channel register — over all its pixels (24 registers
    // through the whole reduction: the kernel runs two waves per SIMD, see the launch, and has 256 to spend)
    float pg[PAR ? 2 : 1][4][3];
    if constexpr (PAR) {
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pg[bt][r][0] = pg[bt][r][1] = pg[bt][r][2] = 0.f;
    }

    // Reduction: per tap 4 steps g of 16 output channels; a step = 4 LDS reads (one per pixel fragment) + 32 MFMAs.  One weight
    // buffer: the two fragments of step g are dead after the step and are refilled at once with those of the wave's NEXT tap (of
    // this class or the first of its next class) — three quarters of a tap (3072 matrix-pipe cycles) ahead of their use; the dy
    // fragments of the next step are read before this step's MFMAs.  The scheduling barriers keep both where they are (left alone
    // the scheduler sinks the weight requests behind all 128 MFMAs of the tap, straight in front of their wait).
    //
    // The weight requests are inline assembly with hand-counted waits.  vmcnt retires in issue order, loads and stores alike, and
    // at the start of a class the queue holds [first tap's weights: 8 | epilogue stores: 8 | DMA of the norm's input: C / 4 + 2]:
    // the fragment a step needs is the OLDEST entry, but the compiler's count for the loop (its states merged over the back
    // edges) is the steady one, vmcnt(6) — which there means "the stores have been acknowledged and the DMA has come back from
    // HBM": measured 44 + 15 us apart and 95 us together of a 505 us launch (profiles/r06ae_dgrad_s3.txt).  Counted by hand:
    //   steady step g: newer than its pair = the 3 pairs requested since -> vmcnt(6);
    //   first tap of a class, step g: (3 - g) pairs of this tap + 8 stores + >= 3 DMA requests + g refills -> vmcnt(17)
    //   (the wave's first class has no stores in front: its first tap's weights are requested in front of the patch fill and waited
    //   for in full behind the barrier).
    // Every count must be a LOWER bound of the requests really issued, so the stores and the DMA are unconditional (pixels outside
    // the map: offset beyond the buffer descriptor, the hardware drops the access).  The waits name the fragment registers as
    // in/out operands: their MFMAs cannot move above them.
    // (the first tap's weights were requested in front of the patch fill: this wait costs nothing, and it makes the first class's
    // queue a subset of the later classes', for which the counts above are written)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(af[0][0]), "+v"(af[1][0]), "+v"(af[0][1]), "+v"(af[1][1]), "+v"(af[0][2]), "+v"(af[1][2]), "+v"(af[0][3]), "+v"(af[1][3]));
    S3_MARK(1);                                      // first tap's weights requested and arrived
    for (int ci_ = 0; ci_ < 3 && cls >= 0; ++ci_) {
        const int nraw = (int)((role_cls >> (4 * (ci_ + 1))) & 15u), ncls = nraw == 15 ? -1 : nraw;
        const int ry = cls / 3, rx = cls - 3 * ry;
        const int nty = ry ? 2 : 3, ntx = rx ? 2 : 3, nt = nty * ntx;
        if constexpr (!(DBG & 8)) issue_y(ry, rx);
        S3_MARK(2);                                  // DMA of the norm's input issued

        f32x4 acc[4][2];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f][0] = acc[f][1] = zero4;
        auto patch_ptr = [&](int tt) -> const f32x4* {
            const int jy = tt / ntx, jx = tt - jy * ntx;
            return pbase - (jy * PW + jx) * PPITCH;
        };
        f32x4 bf[4];
        {
            const f32x4* pt = patch_ptr(0);
#pragma unroll
            for (int f = 0; f < 4; ++f) bf[f] = pt[f * 2 * PW * PPITCH];
        }
        auto tap = [&](auto wait_c, int tt) {
            constexpr int WAITC = decltype(wait_c)::value;
            const bool more = tt + 1 < nt;
            const f32x4* pt = patch_ptr(tt);
            const f32x4* ptn = patch_ptr(more ? tt + 1 : tt);
            // (the wave's very last tap requests its own weights again: a branch would make the request count conditional)
            const f32x4* wpn = more ? wtap_ptr(cls, tt + 1) : wtap_ptr(ncls >= 0 ? ncls : cls, 0);
            const f32x4* wpn1 = wpn + 256;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bn[4];
                if constexpr (DBG & 2) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) bn[f] = bf[f];
                } else if (g < 3) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) bn[f] = pt[f * 2 * PW * PPITCH + 4 * (g + 1)];
                } else {
#pragma unroll
                    for (int f = 0; f < 4; ++f) bn[f] = ptn[f * 2 * PW * PPITCH];     // (after the last tap: read again, unused)
                }
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(af[0][g]), "+v"(af[1][g]) : "n"(WAITC));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][g][e], bf[f][e], acc[f][0], 0, 0, 0);
                        acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][g][e], bf[f][e], acc[f][1], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(DBG & 1)) {
                    if (g == 0) { WDG_S3_LOAD(af[0][0], wpn, 0);    WDG_S3_LOAD(af[1][0], wpn1, 0); }
                    if (g == 1) { WDG_S3_LOAD(af[0][1], wpn, 1024); WDG_S3_LOAD(af[1][1], wpn1, 1024); }
                    if (g == 2) { WDG_S3_LOAD(af[0][2], wpn, 2048); WDG_S3_LOAD(af[1][2], wpn1, 2048); }
                    if (g == 3) { WDG_S3_LOAD(af[0][3], wpn, 3072); WDG_S3_LOAD(af[1][3], wpn1, 3072); }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 4; ++f) bf[f] = bn[f];
            }
        };
        tap(std::integral_constant<int, 17>{}, 0);
        S3_MARK(3);                                  // first tap of the class
        for (int tt = 1; tt < nt; ++tt) tap(std::integral_constant<int, 6>{}, tt);
        S3_MARK(4);                                  // its other taps

        // ---- epilogue of the class: LayerNorm + LeakyReLU backward per pixel (lane li = pixel, registers r of tile bt = channel
        // 16 bt + 4 lq + r), the same arithmetic as epilogues 5 / 6 of wdg_igemm_kernel.  (The DMA of this class's rows was issued
        // before its first tap's weight requests, whose results the reduction has waited for: the rows are in LDS.)
        if constexpr (DBG & 4) {
            // (skeleton: one store per lane keeps the accumulators alive)
            f32x4 sum = zero4;
#pragma unroll
            for (int f = 0; f < 4; ++f) sum += acc[f][0] + acc[f][1];
            if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) p.dx[t] = sum[0];
        } else
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            bool rv;
            const int pix = pixel_of(16 * f + li, ry, rx, rv);
            const unsigned ooff = rv ? ((unsigned)pix * (unsigned)p.ldDx + 4u * lq) << 2 : WDG_SRD_OOB;
            const float* yl = ybuf + (16 * f + li) * p.C;
            const f32x4 y0 = in0 ? *reinterpret_cast<const f32x4*>(yl + nq0) : zero4, y1 = in1 ? *reinterpret_cast<const f32x4*>(yl + nq1) : zero4;
            const float mean = sbuf[16 * f + li], rstd = sbuf[64 + 16 * f + li], rvf = rv ? 1.f : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const f32x4 yv = bt ? y1 : y0, gm = bt ? gm1 : gm0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xh = (yv[r] - mean) * rstd;
                    const float gg = acc[f][bt][r] * gm[r];
                    s1 += gg;
                    s2 += gg * xh;
                }
            }
            s1 = wdg_s3_quarter_sum(s1);
            s2 = wdg_s3_quarter_sum(s2);
            s1 *= invC;
            s2 *= invC;
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const f32x4 yv = bt ? y1 : y0, gm = bt ? gm1 : gm0;
                f32x4 v = acc[f][bt];
                if (bt ? in1 : in0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (yv[r] - mean) * rstd;
                        float d = rstd * (v[r] * gm[r] - s1 - xh * s2);
                        if (p.slope >= 0.f) d *= (yv[r] > 0.f ? 1.f : p.slope);
                        if constexpr (PAR) {
                            // (a pixel outside the map has mean = rstd = 0 and y = 0 from the dropped DMA: xh = d = 0 by themselves,
                            // only the plain sum needs the pixel's 0 / 1 weight — no selects, no branch)
                            pg[bt][r][0] = __builtin_fmaf(v[r], xh, pg[bt][r][0]);
                            pg[bt][r][1] = __builtin_fmaf(v[r], rvf, pg[bt][r][1]);
                            pg[bt][r][2] += d;
                        }
                        v[r] = d;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), srdO, (int)(rv ? ooff + 64u * bt : WDG_SRD_OOB), 0, 0);
            }
        }
        cls = ncls;
        S3_MARK(5);                                  // epilogue (stores issued)
    }
#if WDG_S3_PROF
    __builtin_amdgcn_s_waitcnt(0x0070);
    S3_MARK(6);                                      // ... and everything outstanding completed
    if (lane == 0) {
        for (int k = 0; k < 7; ++k) atomicAdd(&s3_prof[k], (unsigned long long)pp[k]);
        atomicAdd(&s3_prof[7], 1ull);
    }
#endif
    if constexpr (PAR && DBG == 48) {
        // (skeleton: the per-lane sums stay alive through one never-taken store; no row sums, no barrier, no final stores)
        float keep = 0.f;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int r = 0; r < 4; ++r) keep += pg[bt][r][0] + pg[bt][r][1] + pg[bt][r][2];
        if (keep == 12345.678f) p.par[t] = keep;
    }
    if constexpr (PAR && DBG != 48) {
        if (!p.par) return;
        // The 16 pixels of a lane row meet in DPP row sums, the four waves in LDS; the workgroup then stores ITS row [3][32] of the
        // scratch (plain stores, summed by wdg_dgrad_s3_finish_kernel — not float atomics into shared slabs: the sums are the
        // same bits on every run).  (A row per wave, without the barrier: 4 us less here, 14 us more in the finish kernel.)
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float sq = wdg_row16_sum(pg[bt][r][q]);
                    if (li == 0) red[wv * 96 + q * 32 + bt * 16 + 4 * lq + r] = sq;
                }
        __syncthreads();
        if (t < 96 && !(DBG & 16)) p.par[(size_t)tile * 96 + t] = (red[t] + red[96 + t]) + (red[192 + t] + red[288 + t]);
    }
}

}  // namespace

#if WDG_S3_PROF
extern "C" int wdg_s3_prof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(s3_prof), sizeof(s3_prof)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(s3_prof), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

void wdg_dgrad_s3_set(int v) { g_dgrad_s3 = v; }

// the layer shape this kernel is built for; everything else stays on the implicit-GEMM route
bool wdg_dgrad_s3_ok(const wdg_conv_plan* pl, int c0, int C, int ldy_act) {
    const wdg_conv_geom& g = pl->g;
    return g_dgrad_s3 && g.kh == 7 && g.kw == 7 && g.stride == 3 && g.Cin == 32 && g.Cout == 64 && pl->Cout_p == 64 && c0 % 4 == 0 && C % 4 == 0 && C > 0 && c0 + C <= 32 &&
           g.pad_h >= 0 && g.pad_w >= 0 && g.pad_h < 7 && g.pad_w < 7 && g.ldx % 4 == 0 && g.ldy % 4 == 0 && ldy_act % 4 == 0 &&
           (pl->w_ld == 0 || pl->w_ld % 4 == 0) &&
           (int64_t)g.H * g.W * ldy_act * 4 < (int64_t)1 << 31 && (int64_t)g.H * g.W * g.ldx * 4 < (int64_t)1 << 31 &&
           (int64_t)g.Ho * g.Wo * g.ldy * 4 < (int64_t)1 << 31;     // (32-bit byte offsets into one image of the norm's input)
}

constexpr size_t WDG_S3_WS_WEIGHTS = (size_t)49 * 32 * 64 * 4;
// workgroups of a launch: 8 x 8 blocks of base positions b = (i + pad) div 3 over the dx map
static int wdg_s3_tiles(const wdg_conv_geom& g, int* tiles_w, int* tiles_img) {
    const int nb_h = (g.H - 1 + g.pad_h) / 3 - g.pad_h / 3 + 1, nb_w = (g.W - 1 + g.pad_w) / 3 - g.pad_w / 3 + 1;
    const int th = (nb_h + 7) / 8, tw = (nb_w + 7) / 8;
    if (tiles_w) *tiles_w = tw;
    if (tiles_img) *tiles_img = th * tw;
    return th * tw * g.n_img;
}
// scratch of the route; 0: the plan's layer is not this kernel's
size_t wdg_dgrad_s3_ws_bytes(const wdg_conv_plan* pl) {
    const wdg_conv_geom& g = pl->g;
    if (!(g.kh == 7 && g.kw == 7 && g.stride == 3 && g.Cin == 32 && g.Cout == 64 && g.pad_h >= 0 && g.pad_w >= 0)) return 0;
    return WDG_S3_WS_WEIGHTS + (size_t)wdg_s3_tiles(g, nullptr, nullptr) * 96 * 4;     // the weights in fragment order + the workgroups' parameter sums
}

int wdg_dgrad_s3_launch(const wdg_conv_plan* pl, const float* dy, const float* wD, float* dx, const float* y, int ldy_act,
                        int64_t img_stride_act, const float* mean_rstd, const float* gamma, int c0, int C, float act_slope, float* dgamma, float* dbeta, float* dbias,
                        void* ws, hipStream_t stream) {
    const wdg_conv_geom& g = pl->g;
    WdgDgS3 p;
    p.dy = dy; p.wD = wD; p.wS = reinterpret_cast<const f32x4*>(ws); p.dx = dx; p.y = y; p.stats = mean_rstd; p.gamma = gamma;
    const bool par = dgamma || dbeta || dbias;
    p.par = par ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + WDG_S3_WS_WEIGHTS) : nullptr;
    p.imgStrideDy = g.img_stride_y; p.imgStrideDx = g.img_stride_x; p.imgStrideY = img_stride_act;
    p.ldDy = g.ldy; p.ldDx = g.ldx; p.ldY = ldy_act; p.ldB = pl->w_ld ? pl->w_ld : pl->Cout_p;
    p.H = g.H; p.W = g.W; p.Ho = g.Ho; p.Wo = g.Wo; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
    p.bmin_h = g.pad_h / 3; p.bmin_w = g.pad_w / 3;
    p.ntiles = wdg_s3_tiles(g, &p.tiles_w, &p.tiles_img);
    p.rep = 0; p.c0 = c0; p.C = C; p.slope = act_slope;
    // TWO workgroups per CU, not the three the kernel's own 47 KB would allow: with three the weight stream (401 KB per workgroup, every
    // line a miss of the CU's L1) queues up — in-kernel clocks: 114 k clocks from launch to the first MFMA of a wave against 38 k with two and
    // 20 k with one workgroup per CU — and the launch takes 472 us instead of 444 (one per CU: 480; profiles/r06ae_dgrad_s3.txt).  The request is
    // padded to 54 KB; wdg_set_tuning("dgrad_s3", 1 + 256 * k) sets the padding to k KB instead (measurement).
    size_t lds_bytes = (size_t)PATCH_SLOTS * 16 + 4 * 3 * 32 * 4 + 4 * (64 * (size_t)C + 128) * 4;
    if (g_dgrad_s3 >> 8) lds_bytes += (size_t)((g_dgrad_s3 >> 8) - 1) * 1024;
    else lds_bytes = std::max(lds_bytes, (size_t)54 * 1024);
#if WDG_S3_PAIR
    lds_bytes = 2 * ((size_t)PATCH_SLOTS * 16 + 4 * 3 * 32 * 4 + 4 * (64 * (size_t)C + 128) * 4);
    const dim3 s3_grid((unsigned)((p.ntiles + 1) / 2)), s3_block(512);
    {
        static bool done = false;
        if (!done) {
            WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_dgrad_s3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_dgrad_s3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            done = true;
        }
    }
#else
    const dim3 s3_grid((unsigned)p.ntiles), s3_block(256);
#endif
    hipLaunchKernelGGL(wdg_dgrad_s3_pack_kernel, dim3(49 * 2 * 4 * 64 / 256), dim3(256), 0, stream, wD, p.ldB, reinterpret_cast<f32x4*>(ws));
#ifdef WDG_S3_SKELETONS
    // measurement builds only (tools/ab_dgrad_s3_skeletons.sh): wdg_set_tuning("dgrad_s3", 1 + 2 * DBG) runs the input-gradient-only
    // kernel with parts removed — 1: no weight refills, 2: no dy fragment reads, 4: no epilogue, 8: no DMA of the norm's input
    const int dbg = (g_dgrad_s3 >> 1) & 127;
#define WDG_S3_CASE(D) if (dbg == D) { hipLaunchKernelGGL((wdg_dgrad_s3_kernel<false, D>), s3_grid, s3_block, lds_bytes, stream, p); WDG_LAUNCH_CHECK(); return WDG_OK; }
    WDG_S3_CASE(1) WDG_S3_CASE(2) WDG_S3_CASE(4) WDG_S3_CASE(8) WDG_S3_CASE(12) WDG_S3_CASE(13) WDG_S3_CASE(15)
    if (dbg == 16 && par) { hipLaunchKernelGGL((wdg_dgrad_s3_kernel<true, 16>), s3_grid, s3_block, lds_bytes, stream, p); WDG_LAUNCH_CHECK(); return WDG_OK; }   // 16: parameter sums without their final stores
    if (dbg == 48 && par) { hipLaunchKernelGGL((wdg_dgrad_s3_kernel<true, 48>), s3_grid, s3_block, lds_bytes, stream, p); WDG_LAUNCH_CHECK(); return WDG_OK; }   // 48: ... and without the final reduction
#endif
    if (par)
        hipLaunchKernelGGL(wdg_dgrad_s3_kernel<true>, s3_grid, s3_block, lds_bytes, stream, p);
    else
        hipLaunchKernelGGL(wdg_dgrad_s3_kernel<false>, s3_grid, s3_block, lds_bytes, stream, p);
    WDG_LAUNCH_CHECK();
    if (par) {
        hipLaunchKernelGGL(wdg_dgrad_s3_finish_kernel, dim3(3 * C), dim3(1024), 0, stream, p.par, p.ntiles, c0, C, dgamma, dbeta, dbias);
        WDG_LAUNCH_CHECK();
    }
    return WDG_OK;
}
