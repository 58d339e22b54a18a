// sn_adam.hip — spectral-normalisation power iteration (tfa.layers.SpectralNormalization as used at
// /root/reference/src/downscaling/gan/models.py:33,39,49,55,95,103,114,123,134) and the TF-form Adam
// update (/root/reference/src/downscaling/gan/train.py:34-35,57-58), plus library bookkeeping.
//
// The SN update must be bit-identical on every data-parallel rank (weights are replicated and the
// update is a function of the weights only), so every reduction below runs in a fixed order:
// no float atomics.
#include "common.h"
#include <algorithm>
#include <stdarg.h>
#include <vector>

static thread_local char g_err[512] = "";
void wdg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* wdg_last_error(void) { return g_err; }

// CRC-32C (Castagnoli), the checksum of TensorFlow's tensor-bundle checkpoint format (per-tensor and per-block
// checksums of <prefix>.index / .data-*; /root/reference/src/downscaling/gan/ganbase.py:132-140 saves through it).
// Host code: table-driven, byte at a time (checkpoints are tens of MB and written rarely).
extern "C" uint32_t wdg_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    const unsigned char* p = (const unsigned char*)data;
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
extern "C" const char* wdg_version(void) { return "wdgan 0.1 gfx950"; }

// ---- Spectral-norm power iteration, three stages -----------------------------------------------------------------------
// W is [rows][cols] row-major (cols = last kernel axis), u [cols].  tfa: v = l2n(u W^T), u' = l2n(v W), sigma = v W u'^T,
// w <- w / sigma.  The second product is linear in v, so it is formed from the UNNORMALISED v_raw = u W^T and scaled by
// 1 / |v_raw| afterwards: both products come from ONE pass over W.
//   stage 1 (one block per 32-row chunk): the chunk (32 x cols) is read once into registers — 32 * CPT independent loads in
//            flight per thread —, v_raw of its rows (wave sums, then the four waves in fixed order), part1[chunk] = sum of their
//            squares, part2[chunk][c] = sum_r v_raw[r] W[r][c];
//   stage 2 (one block per 64 columns): 1 / |v_raw| from part1, u_raw[c] = that * sum_chunks part2 (four interleaved chunk
//            slices per column, combined in fixed order) -> u, and the block's share of |u_raw|^2 -> normparts;
//   stage 3 (scale, + repack in the batched form): |u_raw|^2 from the <= 16 normparts, sc = 1 / |u_raw|, u <- u_raw * sc,
//            1 / sigma = 1 / (|u_raw|^2 sc), w <- w / sigma.
// Every reduction runs in a fixed order (no float atomics): bit-identical on every data-parallel rank, and the batched and
// the single-layer entry points share these functions (bit-identical to each other, tests/test_ops_gpu.py).
constexpr int SN_R = 32;          // rows per stage-1 chunk
constexpr int SN_CG = 64;         // columns per stage-2 block
constexpr int SN_MAX_COLS = 1024;

template <int CPT>
__device__ __forceinline__ void wdg_sn_chunk_rows(int blk, const float* __restrict__ w, const float* __restrict__ u, int rows,
                                                  int cols, float* part1, float* part2, float (*red)[SN_R], float* vs) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int r0 = blk * SN_R;
    float wr[SN_R][CPT], uu[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) uu[j] = t + 256 * j < cols ? u[t + 256 * j] : 0.f;
#pragma unroll
    for (int r = 0; r < SN_R; ++r)
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            wr[r][j] = (r0 + r < rows && t + 256 * j < cols) ? w[(size_t)(r0 + r) * cols + t + 256 * j] : 0.f;
#pragma unroll
    for (int r = 0; r < SN_R; ++r) {
        float p_ = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) p_ += uu[j] * wr[r][j];
        p_ = wdg_wave_sum_fast(p_);
        if (lane == 0) red[wave][r] = p_;
    }
    __syncthreads();
    if (t < SN_R) vs[t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    __syncthreads();
    if (t == 0) {
        float sq = 0.f;
        for (int r = 0; r < SN_R; ++r) sq += vs[r] * vs[r];
        part1[blk] = sq;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        float s_ = 0.f;
#pragma unroll
        for (int r = 0; r < SN_R; ++r) s_ += vs[r] * wr[r][j];
        if (t + 256 * j < cols) part2[(size_t)blk * cols + t + 256 * j] = s_;
    }
}
__device__ __forceinline__ void wdg_sn_chunk_block(int blk, const float* __restrict__ w, const float* __restrict__ u, int rows,
                                                   int cols, float* part1, float* part2) {
    __shared__ float red[4][SN_R];
    __shared__ float vs[SN_R];
    if (cols <= 256) wdg_sn_chunk_rows<1>(blk, w, u, rows, cols, part1, part2, red, vs);
    else if (cols <= 512) wdg_sn_chunk_rows<2>(blk, w, u, rows, cols, part1, part2, red, vs);
    else wdg_sn_chunk_rows<4>(blk, w, u, rows, cols, part1, part2, red, vs);
}

// fixed-order sum of 256 per-thread values (tree through LDS); result in every thread
__device__ __forceinline__ float wdg_block_sum256(float v, float* red) {
    const int t = threadIdx.x;
    __syncthreads();
    red[t] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ void wdg_sn_colfinish_block(int cg, const float* __restrict__ part1, const float* __restrict__ part2,
                                                       int nchunks, int cols, float* u, float* normparts) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    float a = 0.f;
    for (int i = t; i < nchunks; i += 256) a += part1[i];
    const float vscale = 1.f / sqrtf(fmaxf(wdg_block_sum256(a, red), 1e-12f));      // tf.math.l2_normalize of v
    const int c = cg * SN_CG + (t & 63), ks = t >> 6;
    float s_ = 0.f;
    if (c < cols)
        for (int k = ks; k < nchunks; k += 4) s_ += part2[(size_t)k * cols + c];
    red[t] = s_;
    __syncthreads();
    float sq = 0.f;
    if (t < 64) {
        const float ur = ((red[t] + red[t + 64]) + (red[t + 128] + red[t + 192])) * vscale;
        if (c < cols) u[c] = ur;                                                    // u_raw; normalised in stage 3
        sq = c < cols ? ur * ur : 0.f;
    }
    const float n2 = wdg_block_sum256(sq, red);
    if (t == 0) normparts[cg] = n2;
}

// stage-3 scalars of one layer: (sc = 1 / |u_raw|, 1 / sigma)
__device__ __forceinline__ void wdg_sn_sigma(const float* __restrict__ normparts, int ncg, float& sc, float& inv_sigma) {
    float n2 = 0.f;
    for (int g = 0; g < ncg; ++g) n2 += normparts[g];
    sc = 1.f / sqrtf(fmaxf(n2, 1e-12f));
    inv_sigma = 1.f / (n2 * sc);                                                    // sigma = <u_raw, u_new> = n2 * sc
}

__global__ void __launch_bounds__(256) wdg_sn_chunk_kernel(const float* __restrict__ w, const float* __restrict__ u, int rows,
                                                           int cols, float* part1, float* part2) {
    wdg_sn_chunk_block(blockIdx.x, w, u, rows, cols, part1, part2);
}
__global__ void __launch_bounds__(256) wdg_sn_colfinish_kernel(const float* __restrict__ part1, const float* __restrict__ part2,
                                                               int nchunks, int cols, float* u, float* normparts) {
    wdg_sn_colfinish_block(blockIdx.x, part1, part2, nchunks, cols, u, normparts);
}
__global__ void __launch_bounds__(256) wdg_sn_scale_kernel(float* w, int64_t n, float* u, int cols, const float* __restrict__ normparts,
                                                           int ncg) {
    float sc, k;
    wdg_sn_sigma(normparts, ncg, sc, k);
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < cols; c += 256) u[c] *= sc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) w[i] *= k;
}

static inline size_t wdg_sn_chunks(int rows) { return (size_t)(rows + SN_R - 1) / SN_R; }
static inline size_t wdg_sn_colgroups(int cols) { return (size_t)(cols + SN_CG - 1) / SN_CG; }
extern "C" size_t wdg_sn_scratch_floats(int rows, int cols) {
    // part1 [nchunks] | part2 [nchunks][cols] | normparts [colgroups] (+ alignment slack)
    return wdg_sn_chunks(rows) * ((size_t)cols + 1) + wdg_sn_colgroups(cols) + 8;
}

extern "C" int wdg_sn_power_iter(float* w, float* u, int rows, int cols, float* scratch, wdg_stream stream) {
    WDG_CHECK_ARG(w && u && scratch && rows > 0 && cols > 0, "bad argument");
    WDG_CHECK_ARG(cols <= SN_MAX_COLS, "spectral normalisation: more than 1024 columns (last kernel axis) are not supported");
    hipStream_t st = (hipStream_t)stream;
    const int nchunks = (int)wdg_sn_chunks(rows), ncg = (int)wdg_sn_colgroups(cols);
    float* part1 = scratch;
    float* part2 = part1 + nchunks;
    float* normparts = part2 + (size_t)nchunks * cols;
    hipLaunchKernelGGL(wdg_sn_chunk_kernel, dim3(nchunks), dim3(256), 0, st, w, u, rows, cols, part1, part2);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_sn_colfinish_kernel, dim3(ncg), dim3(256), 0, st, part1, part2, nchunks, cols, u, normparts);
    WDG_LAUNCH_CHECK();
    const int64_t n = (int64_t)rows * cols;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 1023) / 1024, 2048));
    hipLaunchKernelGGL(wdg_sn_scale_kernel, dim3(blocks), dim3(256), 0, st, w, n, u, cols, normparts, ncg);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- Batched weight preparation of one network ----------------------------------------------------
// Every training-mode forward of the reference first runs the SN power iteration of ALL its wrapped layers
// (w <- w / sigma, u updated in place) and then needs the kernel-layout copies of the changed weights.  Layer
// by layer that is several tiny launches per layer (~400 per train step, each followed by the inter-kernel
// drain).  A batch object holds the layer table on the device; one launch per STAGE covers all layers
// (blockIdx -> layer through block-offset prefixes): stage 1 and 2 as above, and a third launch that scales the
// master weights in place AND writes the kernel layouts from the same read: a block owns 32 x 64 (ci, co) tiles of
// one tap, reads them coalesced along co, writes w (and wD) back from the same registers, transposes through LDS
// and writes wF rows coalesced along ci — W is read twice and written twice per preparation in total (was 3 + 2
// with strided gathers for the forward layout).
struct WdgPrepLayer {
    float* w;
    float* u;
    float* wF;
    float* wD;
    long long s_off;        // scratch offset of this layer (floats)
    int rows, cols;         // SN matrix view
    int taps, cin, cout;    // packing geometry
    int sn;
    int nchunks, ncg;
    int b1, b2, b5;         // first block of this layer in the chunk / column-finish / scale-pack grids
    int nb5, ntiles, tiles_ci, tiles_co;
};
struct wdg_prep_batch {
    std::vector<WdgPrepLayer> h;
    WdgPrepLayer* d = nullptr;
    int n = 0, n_sn = 0;
    int g1 = 0, g2 = 0;              // SN-stage grids (SN layers are sorted first)
    int g5_sn = 0, g5_all = 0;       // scale / pack grids: SN layers only / every layer
    size_t scratch_floats = 0;
};

__device__ __forceinline__ int wdg_prep_find(const WdgPrepLayer* L, int n, int blk, int which) {
    int l = 0;
    for (int i = 1; i < n; ++i) {
        const int b = which == 1 ? L[i].b1 : which == 2 ? L[i].b2 : L[i].b5;
        if (blk >= b) l = i;
    }
    return l;
}

__global__ void __launch_bounds__(256) wdg_prep_chunk_kernel(const WdgPrepLayer* __restrict__ L, int n, float* scratch) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 1);
    const WdgPrepLayer q = L[l];
    float* part1 = scratch + q.s_off;
    wdg_sn_chunk_block(blockIdx.x - q.b1, q.w, q.u, q.rows, q.cols, part1, part1 + q.nchunks);
}
__global__ void __launch_bounds__(256) wdg_prep_colfinish_kernel(const WdgPrepLayer* __restrict__ L, int n, float* scratch) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 2);
    const WdgPrepLayer q = L[l];
    float* part1 = scratch + q.s_off;
    float* part2 = part1 + q.nchunks;
    wdg_sn_colfinish_block(blockIdx.x - q.b2, part1, part2, q.nchunks, q.cols, q.u, part2 + (size_t)q.nchunks * q.cols);
}
// master HWIO [tap][ci][co] -> (scaled in place when do_sn and the layer is wrapped) + wF [Cout][taps][Cin_p] + wD
// [taps][Cin][Cout_p] (same maps as wdg_weight_pack_kernel)
constexpr int PK_CI = 32, PK_CO = 64;
__global__ void __launch_bounds__(256) wdg_prep_scale_pack_kernel(const WdgPrepLayer* __restrict__ L, int n, const float* scratch,
                                                                  int do_sn) {
    __shared__ float tile[PK_CI][PK_CO + 1];
    const int l = wdg_prep_find(L, n, blockIdx.x, 5);
    const WdgPrepLayer q = L[l];
    const int t = threadIdx.x;
    const bool scale = do_sn && q.sn;
    float k = 1.f;
    if (scale) {
        const float* normparts = scratch + q.s_off + q.nchunks + (size_t)q.nchunks * q.cols;
        float sc;
        wdg_sn_sigma(normparts, q.ncg, sc, k);
        if (blockIdx.x == q.b5)
            for (int c = t; c < q.cols; c += 256) q.u[c] *= sc;
    }
    const int Cin_p = (q.cin + 3) & ~3, Cout_p = (q.cout + 3) & ~3;
    for (int tl = blockIdx.x - q.b5; tl < q.ntiles; tl += q.nb5) {
        const int tco = tl % q.tiles_co;
        const int r_ = tl / q.tiles_co;
        const int tci = r_ % q.tiles_ci;
        const int tap = r_ / q.tiles_ci;
        const int ci0 = tci * PK_CI, co0 = tco * PK_CO;
        {
            const int co = co0 + (t & 63);
#pragma unroll
            for (int i = 0; i < PK_CI / 4; ++i) {
                const int cil = (t >> 6) + 4 * i;
                const int ci = ci0 + cil;
                float v = 0.f;
                if (ci < q.cin && co < q.cout) {
                    const size_t idx = ((size_t)tap * q.cin + ci) * q.cout + co;
                    v = q.w[idx] * k;
                    if (scale) q.w[idx] = v;
                }
                if (q.wD && ci < q.cin && co < Cout_p) q.wD[((size_t)tap * q.cin + ci) * Cout_p + co] = v;
                tile[cil][t & 63] = v;
            }
        }
        __syncthreads();
        if (q.wF) {
            const int ci = ci0 + (t & 31);
#pragma unroll
            for (int i = 0; i < PK_CO / 8; ++i) {
                const int col = (t >> 5) + 8 * i;
                const int co = co0 + col;
                if (ci < Cin_p && co < q.cout) q.wF[((size_t)co * q.taps + tap) * Cin_p + ci] = tile[t & 31][col];
            }
        }
        __syncthreads();
    }
}

extern "C" int wdg_prep_batch_create(wdg_prep_batch** out, const wdg_prep_layer* layers, int n) {
    WDG_CHECK_ARG(out && layers && n > 0 && n <= 64, "bad argument");
    wdg_prep_batch* b = new wdg_prep_batch();
    // SN layers first (the SN stages index only them), original order otherwise
    std::vector<int> order;
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < n; ++i)
            if ((layers[i].sn != 0) == (pass == 0)) order.push_back(i);
    long long soff = 0;
    int b1 = 0, b2 = 0, b5 = 0;
    for (int idx : order) {
        const wdg_prep_layer& s = layers[idx];
        if (!s.w || s.taps <= 0 || s.cin <= 0 || s.cout <= 0 ||
            (s.sn && (!s.u || s.rows <= 0 || s.cols <= 0 || s.cols > SN_MAX_COLS))) {
            delete b;
            wdg_set_error("wdg_prep_batch_create: bad layer %d (spectral-normalised layers: at most %d columns)", idx, SN_MAX_COLS);
            return WDG_ERR_ARG;
        }
        WdgPrepLayer q;
        memset(&q, 0, sizeof(q));
        q.w = s.w; q.u = s.u; q.wF = s.wF; q.wD = s.wD;
        q.rows = s.rows; q.cols = s.cols; q.taps = s.taps; q.cin = s.cin; q.cout = s.cout; q.sn = s.sn != 0;
        q.s_off = soff;
        q.b1 = b1; q.b2 = b2; q.b5 = b5;
        if (q.sn) {
            q.nchunks = (int)wdg_sn_chunks(q.rows);
            q.ncg = (int)wdg_sn_colgroups(q.cols);
            soff += (long long)((wdg_sn_scratch_floats(q.rows, q.cols) + 3) & ~(size_t)3);
            b1 += q.nchunks; b2 += q.ncg;
            b->n_sn++;
        }
        const int Cin_p = (q.cin + 3) & ~3, Cout_p = (q.cout + 3) & ~3;
        q.tiles_ci = (Cin_p + PK_CI - 1) / PK_CI;
        q.tiles_co = (Cout_p + PK_CO - 1) / PK_CO;
        q.ntiles = q.taps * q.tiles_ci * q.tiles_co;
        q.nb5 = std::max(1, std::min(q.ntiles, 512));
        b5 += q.nb5;
        if (q.sn) b->g5_sn = b5;
        b->h.push_back(q);
    }
    b->n = n;
    b->g1 = b1; b->g2 = b2; b->g5_all = b5;
    b->scratch_floats = (size_t)soff + 8;
    if (hipMalloc((void**)&b->d, sizeof(WdgPrepLayer) * n) != hipSuccess ||
        hipMemcpy(b->d, b->h.data(), sizeof(WdgPrepLayer) * n, hipMemcpyHostToDevice) != hipSuccess) {
        delete b;
        wdg_set_error("wdg_prep_batch_create: device allocation failed");
        return WDG_ERR_HIP;
    }
    *out = b;
    return WDG_OK;
}

extern "C" size_t wdg_prep_batch_scratch_floats(const wdg_prep_batch* b) { return b ? b->scratch_floats : 0; }

extern "C" int wdg_prep_batch_destroy(wdg_prep_batch* b) {
    if (!b) return WDG_OK;
    if (b->d) (void)hipFree(b->d);
    delete b;
    return WDG_OK;
}

// flags: WDG_PREP_SN = power iteration + in-place w / sigma on the SN layers, then repack them;
//        WDG_PREP_PACK_ALL = repack every layer (after an optimizer step or a weight load).
extern "C" int wdg_prep_batch_run(const wdg_prep_batch* b, float* scratch, int flags, wdg_stream stream) {
    WDG_CHECK_ARG(b, "null batch");
    hipStream_t st = (hipStream_t)stream;
    const bool sn = (flags & WDG_PREP_SN) && b->n_sn > 0;
    if (sn) {
        WDG_CHECK_ARG(scratch, "scratch required");
        hipLaunchKernelGGL(wdg_prep_chunk_kernel, dim3(b->g1), dim3(256), 0, st, b->d, b->n_sn, scratch);
        hipLaunchKernelGGL(wdg_prep_colfinish_kernel, dim3(b->g2), dim3(256), 0, st, b->d, b->n_sn, scratch);
        WDG_LAUNCH_CHECK();
    }
    const int g5 = (flags & WDG_PREP_PACK_ALL) ? b->g5_all : (sn ? b->g5_sn : 0);
    if (g5 > 0) {
        hipLaunchKernelGGL(wdg_prep_scale_pack_kernel, dim3(g5), dim3(256), 0, st, b->d, (flags & WDG_PREP_PACK_ALL) ? b->n : b->n_sn,
                           scratch, sn ? 1 : 0);
        WDG_LAUNCH_CHECK();
    }
    return WDG_OK;
}

// ---- Adam, TF form --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wdg_adam_tf_kernel(float* p, const float* __restrict__ g, float* m,
                                                          float* v, int64_t n, float lr_t, float b1, float b2,
                                                          float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = m[i] + (1.f - b1) * (gi - m[i]);
        const float vi = v[i] + (1.f - b2) * (gi * gi - v[i]);
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int wdg_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                           float beta2, float eps, float grad_scale, wdg_stream stream) {
    WDG_CHECK_ARG(p && g && m && v && n >= 0, "bad argument");
    if (n == 0) return WDG_OK;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192));
    hipLaunchKernelGGL(wdg_adam_tf_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t,
                       beta1, beta2, eps, grad_scale);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- several ranges of a flat buffer zeroed in ONE launch (ParamStore.zero_grad(lazy=True): the alignment ranges between the lazily
// zeroed kernel gradients were one torch fill launch each, ~25 per step on the streams' serial sections) ------------------------
struct WdgZeroRanges {
    long long begin[16], end[16];     // element ranges, every begin / end a multiple of 4
    int n;
};
__global__ void __launch_bounds__(256) wdg_zero_ranges_kernel(float* base, const WdgZeroRanges r) {
    const int which = blockIdx.y;
    if (which >= r.n) return;
    f32x4* p = reinterpret_cast<f32x4*>(base + r.begin[which]);
    const long long n4 = (r.end[which] - r.begin[which]) >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

extern "C" int wdg_zero_ranges(float* base, const int64_t* begin_end, int n, wdg_stream stream) {
    WDG_CHECK_ARG(base && begin_end && n >= 1 && n <= 16 && ((uintptr_t)base & 15) == 0, "1..16 ranges of a 16-byte aligned buffer");
    WdgZeroRanges r;
    long long longest = 0;
    for (int i = 0; i < n; ++i) {
        r.begin[i] = begin_end[2 * i];
        r.end[i] = begin_end[2 * i + 1];
        WDG_CHECK_ARG(r.begin[i] >= 0 && r.end[i] >= r.begin[i] && (r.begin[i] & 3) == 0 && (r.end[i] & 3) == 0, "ranges must be multiples of 4 elements");
        longest = std::max(longest, r.end[i] - r.begin[i]);
    }
    r.n = n;
    const int bx = (int)std::max<long long>(1, std::min<long long>((longest / 4 + 255) / 256, 512));
    hipLaunchKernelGGL(wdg_zero_ranges_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, base, r);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

