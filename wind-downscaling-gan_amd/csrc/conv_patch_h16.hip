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
    int fw_shift, tfx;         // fragment width 1 << fw_shift (16 or 4), fragments per fragment-row of the tile
    int TH, TW, PH, PW, PWs, pitch;   // tile, patch, columns per parity plane, slots per channel-group plane
    int CK8, nchunk, kcn;      // channel groups per chunk, chunks, K-steps per tap and chunk
    int tiles_x, tiles_y, tiles_n;
    wdg_fastdiv div_tn, div_tx, div_ty, div_ck, div_pw;
};

// MT fragments of 16 pixels per wave (2 waves along the pixels), NT 16-channel tiles per wave (2 waves along the channels)
template <int FMT, int MT, int NT>
__global__ void __launch_bounds__(256) wdg_conv_patch_h16_kernel(const WdgPatchH16 p) {
    typedef wdg_h16x8<FMT> h16x8;
    constexpr int BN = 2 * NT * 16;
    constexpr int B_LOADS = 8 * BN / 256;          // 16-byte weight slots per thread and stage (2 K-steps)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    h16x8* ldsB = reinterpret_cast<h16x8*>(smem_raw);              // [2 stages][8 planes][BN]
    h16x8* ldsP = ldsB + 2 * 8 * BN;                                // [CK8][pitch] patch

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
    const int n0 = tn * BN;

    const wdg_srd srdA = wdg_make_srd(p.A + (long long)img * p.imgStrideA);
    const wdg_srd srdB = wdg_make_srd(p.B);

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
        opix[a] = (oy0 + oyl) * p.Wo + ox0 + oxl;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- this thread's weight slots of a stage: j = (K-step of the stage, channel group q), column n
    int b_row[B_LOADS];        // element offset of row n, or -1
    int b_slot[B_LOADS];       // LDS slot inside a stage
    const int bj = t & 7, bh = bj >> 2, bq = bj & 3;
#pragma unroll
    for (int r = 0; r < B_LOADS; ++r) {
        const int n = (t >> 3) + 32 * r;
        b_row[r] = (n0 + n < p.Ncols) ? (n0 + n) * p.ldB : -1;
        b_slot[r] = bj * BN + (n ^ bj);
    }

    const int nks = p.kh * p.kw * p.kcn;            // K-steps per chunk
    const int nstage = (nks + 1) >> 1;
    const int npatch = p.CK8 * p.PH * p.PW;

    for (int ck = 0; ck < p.nchunk; ++ck) {
        __syncthreads();                             // every wave is done with the previous chunk's patch and weight stages
        // ---- patch chunk: global fp32 -> 16-bit -> LDS, 4 slots per thread in flight
        for (int base = 0; base < npatch; base += 4 * 256) {
            f32x4 v[4][2];
            int slot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * 256 + t;
                const int pix = (int)wdg_fastdiv_do((unsigned)idx, p.div_ck);
                const int c = idx - pix * p.CK8;
                const int y = (int)wdg_fastdiv_do((unsigned)pix, p.div_pw);
                const int x = pix - y * p.PW;
                const int gy = iy0 + y, gx = ix0 + x;
                const bool ok = idx < npatch && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const unsigned off = ok ? (unsigned)((gy * p.W + gx) * p.ldA + (ck * p.CK8 + c) * 8) << 2 : WDG_SRD_OOB;
                v[u][0] = wdg_buffer_load_f32x4(srdA, off);
                v[u][1] = wdg_buffer_load_f32x4(srdA, ok ? off + 16u : WDG_SRD_OOB);
                slot[u] = idx < npatch ? c * p.pitch + (((y << p.sshift) + (x & (s - 1))) * p.PWs) + (x >> p.sshift) : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (slot[u] >= 0) ldsP[slot[u]] = wdg_pack_h16<FMT>(v[u][0], v[u][1]);
        }

        // ---- weight stages: (tap, kc) counters of the stage being FETCHED (f*) and of the stage being COMPUTED (c*)
        int f_kc = 0, f_kx = 0, f_ky = 0;
        int c_kc = 0, c_kx = 0, c_ky = 0;
        u32x4 rb[B_LOADS];
        auto advance = [&](int& kc, int& kx, int& ky) {
            if (++kc == p.kcn) {
                kc = 0;
                if (++kx == p.kw) { kx = 0; ++ky; }
            }
        };
        auto fetch_stage = [&]() {
            // the two K-steps of the stage; this thread serves K-step bh, channel group bq
            int kc0 = f_kc, kx0 = f_kx, ky0 = f_ky;
            advance(f_kc, f_kx, f_ky);
            int kc1 = f_kc, kx1 = f_kx, ky1 = f_ky;
            advance(f_kc, f_kx, f_ky);
            const int kc = bh ? kc1 : kc0, kx = bh ? kx1 : kx0, ky = bh ? ky1 : ky0;
            const int g8 = kc * 4 + bq;              // channel group inside the chunk
            const bool kok = ky < p.kh && g8 < p.CK8;
            const int koff = (ky * p.kw + kx) * p.Cin_p + (ck * p.CK8 + g8) * 8;
#pragma unroll
            for (int r = 0; r < B_LOADS; ++r)
                rb[r] = __builtin_amdgcn_raw_buffer_load_b128(srdB, (kok && b_row[r] >= 0) ? (int)((unsigned)(b_row[r] + koff) << 1) : (int)WDG_SRD_OOB, 0, 0);
        };
        fetch_stage();
        for (int st = 0; st < nstage; ++st) {
            h16x8* sB = ldsB + (st & 1) * 8 * BN;
#pragma unroll
            for (int r = 0; r < B_LOADS; ++r) sB[b_slot[r]] = __builtin_bit_cast(h16x8, rb[r]);
            __syncthreads();
            if (st + 1 < nstage) fetch_stage();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (c_ky < p.kh) {                   // (an odd K-step count leaves the last stage half empty)
                    const int g8 = c_kc * 4 + lq;
                    // lanes whose channel group is past the chunk read group 0 (finite data); their weights are zero
                    const int tapoff = (((c_ky << p.sshift) + (c_kx & (s - 1))) * p.PWs) + (c_kx >> p.sshift) + (g8 < p.CK8 ? g8 : 0) * p.pitch;
                    const int pl = h * 4 + lq;
                    h16x8 af[MT], bf[NT];
#pragma unroll
                    for (int b = 0; b < NT; ++b) bf[b] = sB[pl * BN + ((wn * (BN / 2) + b * 16 + li) ^ pl)];
#pragma unroll
                    for (int a = 0; a < MT; ++a) af[a] = ldsP[fbase[a] + tapoff];
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
                            acc[a][b] = wdg_mfma16<FMT>(bf[b], af[a], acc[a][b]);
                }
                advance(c_kc, c_kx, c_ky);
            }
        }
    }

    // ---- epilogue: lane (li, lq) holds channels 4*lq .. 4*lq+3 of pixel li of every fragment
    float* outImg = p.Out + (long long)img * p.imgStrideO;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + wn * (BN / 2) + b * 16 + 4 * lq;
        if (n >= p.Ncols) continue;
        const bool full = n + 3 < p.Ncols;
        f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f}, sc4 = (f32x4){1.f, 1.f, 1.f, 1.f}, sh4 = bias4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < p.Ncols) {
                if (p.bias) bias4[r] = p.bias[n + r];
                if (p.affine) { sc4[r] = p.affine[n + r]; sh4[r] = p.affine[p.Ncols + n + r]; }
            }
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            float* dst = outImg + (long long)opix[a] * p.ldO + n;
            f32x4 v = acc[a][b] + bias4;
            if (p.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = wdg_lrelu(v[r], p.slope);
            }
            if (p.affine) v = v * sc4 + sh4;
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

// ---------------------------------------------------------------------------------------------------------------------
static int g_patch_h16 = 1;
static int g_patch_budget = 44 * 1024;     // LDS bytes of a patch chunk: with the 32 KiB of weight stages two workgroups share a CU
void wdg_patch_h16_set(int v) { g_patch_h16 = v; }
void wdg_patch_h16_set_budget(int kib) { g_patch_budget = kib * 1024; }

struct WdgPatchCfg {
    int fw_shift, tfx, TH, TW, MT;
};

// tile shape for the output map, or false
static bool patch_shape(const wdg_conv_geom& g, WdgPatchCfg& c) {
    if (g.Ho % 8) return false;
    if (g.Wo % 16 == 0) { c = {4, 1, 8, 16, 4}; return true; }       // 8 fragments of 1 x 16
    if (g.Wo % 24 == 0) { c = {2, 6, 8, 24, 6}; return true; }       // 12 fragments of 4 x 4
    return false;
}

static bool patch_plan(const wdg_conv_plan* pl, WdgPatchH16& p) {
    const wdg_conv_geom& g = pl->g;
    if (!g_patch_h16 || (g.stride != 1 && g.stride != 2) || pl->Cin_p % 8 || g.Cout < 32 || g.ldy % 4) return false;
    WdgPatchCfg c;
    if (!patch_shape(g, c)) return false;
    const int s = g.stride;
    p.sshift = s == 2 ? 1 : 0;
    p.fw_shift = c.fw_shift; p.tfx = c.tfx; p.TH = c.TH; p.TW = c.TW;
    p.PH = (c.TH - 1) * s + g.kh;
    p.PW = (c.TW - 1) * s + g.kw;
    int PWs = (p.PW + s - 1) / s;
    if (c.fw_shift == 2)
        while (((s * s * PWs) & 15) != 4 && ((s * s * PWs) & 15) != 12) ++PWs;    // rows of a 4 x 4 fragment on disjoint banks
    p.PWs = PWs;
    p.pitch = wdg_round_up(p.PH * s * PWs, 16);
    const int C8 = pl->Cin_p / 8;
    int CK8 = 0;
    if ((long long)C8 * p.pitch * 16 <= g_patch_budget) CK8 = C8;
    else
        for (int k = 4; k < C8; k += 4)
            if (C8 % k == 0 && (long long)k * p.pitch * 16 <= g_patch_budget) CK8 = k;
    if (!CK8) return false;
    p.CK8 = CK8; p.nchunk = C8 / CK8; p.kcn = (CK8 + 3) / 4;
    p.tiles_x = g.Wo / c.TW; p.tiles_y = g.Ho / c.TH;
    return true;
}

int wdg_patch_h16_eligible(const wdg_conv_plan* pl) {
    WdgPatchH16 p;
    memset(&p, 0, sizeof(p));
    return patch_plan(pl, p) ? 1 : 0;
}

template <int FMT, int MT, int NT>
static int patch_launch(const WdgPatchH16& p, int blocks, size_t lds, hipStream_t st) {
    static size_t lds_set = 0;
    if (lds > lds_set) {
        WDG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wdg_conv_patch_h16_kernel<FMT, MT, NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL((wdg_conv_patch_h16_kernel<FMT, MT, NT>), dim3(blocks), dim3(256), lds, st, p);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// returns WDG_OK after launching, or 1 when the geometry is not eligible (caller falls back to the gather kernel)
int wdg_patch_h16_launch(const wdg_conv_plan* pl, const float* x, const void* w16, const float* bias, const float* affine,
                         float* y, int act, float slope, int accumulate, int fmt, hipStream_t st) {
    WdgPatchH16 p;
    memset(&p, 0, sizeof(p));
    if (!patch_plan(pl, p)) return 1;
    const wdg_conv_geom& g = pl->g;
    p.A = x; p.B = w16; p.Out = y; p.bias = bias; p.affine = affine;
    p.imgStrideA = g.img_stride_x; p.imgStrideO = g.img_stride_y;
    p.H = g.H; p.W = g.W; p.ldA = g.ldx; p.Ho = g.Ho; p.Wo = g.Wo; p.ldO = g.ldy;
    p.Ncols = g.Cout; p.ldB = pl->taps * pl->Cin_p; p.Cin_p = pl->Cin_p;
    p.kh = g.kh; p.kw = g.kw; p.pad_h = g.pad_h; p.pad_w = g.pad_w;
    p.act = act; p.slope = slope; p.accumulate = accumulate;
    if ((long long)g.H * g.W * g.ldx * 4 >= (1LL << 31) || (long long)g.Cout * p.ldB * 2 >= (1LL << 31)) return 1;
    const int MT = p.tfx == 1 ? 4 : 6;
    // 64-channel tiles when the map is small (the per-timestep recurrent convolution) or the layer is narrow
    const long long tiles_px = (long long)g.n_img * p.tiles_x * p.tiles_y;
    const bool narrow = g.Cout <= 64 || tiles_px * ((g.Cout + 127) / 128) < (long long)pl->cus * 3 / 2;
    const int BN = narrow ? 64 : 128;
    p.tiles_n = (g.Cout + BN - 1) / BN;
    p.div_tn = wdg_fastdiv_make((unsigned)p.tiles_n);
    p.div_tx = wdg_fastdiv_make((unsigned)p.tiles_x);
    p.div_ty = wdg_fastdiv_make((unsigned)p.tiles_y);
    p.div_ck = wdg_fastdiv_make((unsigned)p.CK8);
    p.div_pw = wdg_fastdiv_make((unsigned)p.PW);
    const long long blocks = tiles_px * p.tiles_n;
    if (blocks <= 0 || blocks >= (1LL << 31)) return 1;
    const size_t lds = (size_t)2 * 8 * BN * 16 + (size_t)p.CK8 * p.pitch * 16;
#define WDG_PATCH_CASE(F, M, N) if (fmt == F && MT == M && BN == 32 * N) return patch_launch<F, M, N>(p, (int)blocks, lds, st)
    WDG_PATCH_CASE(0, 4, 4); WDG_PATCH_CASE(0, 4, 2); WDG_PATCH_CASE(0, 6, 4); WDG_PATCH_CASE(0, 6, 2);
    WDG_PATCH_CASE(1, 4, 4); WDG_PATCH_CASE(1, 4, 2); WDG_PATCH_CASE(1, 6, 4); WDG_PATCH_CASE(1, 6, 2);
#undef WDG_PATCH_CASE
    return 1;
}
