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

// ---- SN step 1: vraw[r] = <u, W[r,:]>, per-block partial sum of squares ---------------------------
__device__ __forceinline__ void wdg_sn_rowdot_block(int blk, const float* __restrict__ w, const float* __restrict__ u,
                                                    int rows, int cols, float* vraw, float* part1) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sq = 0.f;
    // each wave owns rows blk*32 + wave*8 .. +8
    for (int i = 0; i < 8; ++i) {
        const int r = blk * 32 + wave * 8 + i;
        if (r >= rows) break;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += u[c] * w[(size_t)r * cols + c];
        s = wdg_wave_sum(s);
        if (lane == 0) {
            vraw[r] = s;
            sq += s * s;
        }
    }
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part1[blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- SN step 2: per 64-row chunk, part2[chunk][c] = sum_r v[r] W[r][c] ----------------------------
__device__ __forceinline__ void wdg_sn_colpart_block(int blk, const float* __restrict__ w, const float* __restrict__ vraw,
                                                     const float* __restrict__ part1, int nb1, int rows, int cols,
                                                     float* part2) {
    __shared__ float vs[64];
    __shared__ float s_scale;
    if (threadIdx.x == 0) {
        float n2 = 0.f;
        for (int i = 0; i < nb1; ++i) n2 += part1[i];  // fixed order
        s_scale = 1.f / sqrtf(fmaxf(n2, 1e-12f));      // tf.math.l2_normalize
    }
    __syncthreads();
    const int r0 = blk * 64;
    if (threadIdx.x < 64) vs[threadIdx.x] = (r0 + threadIdx.x < rows) ? vraw[r0 + threadIdx.x] * s_scale : 0.f;
    __syncthreads();
    const int nr = min(64, rows - r0);
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s += vs[r] * w[(size_t)(r0 + r) * cols + c];
        part2[(size_t)blk * cols + c] = s;
    }
}

// ---- SN steps 1 + 2 fused (batched path): one block per 64-row chunk computes vraw for its rows AND the chunk's
// contribution to v W — the second product is linear in v, so it is formed from the UNNORMALISED vraw and scaled by
// 1 / |vraw| in the finish stage.  W is streamed from HBM once per power iteration instead of twice (the chunk, 64 x cols
// floats, is re-read from L1 / L2), and one launch + its dependency bubble disappears.
__device__ __forceinline__ void wdg_sn_chunk_block(int blk, const float* __restrict__ w, const float* __restrict__ u,
                                                   int rows, int cols, float* vraw, float* part1, float* part2) {
    __shared__ float vs[64];
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blk * 64;
    float sq = 0.f;
    for (int i = 0; i < 16; ++i) {                       // each wave owns 16 of the chunk's rows
        const int r = r0 + wave * 16 + i;
        float s_ = 0.f;
        if (r < rows)
            for (int c = lane; c < cols; c += 64) s_ += u[c] * w[(size_t)r * cols + c];
        s_ = wdg_wave_sum(s_);
        if (lane == 0) {
            vs[wave * 16 + i] = r < rows ? s_ : 0.f;
            if (r < rows) vraw[r] = s_;
            sq += r < rows ? s_ * s_ : 0.f;
        }
    }
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part1[blk] = (red[0] + red[1]) + (red[2] + red[3]);
    const int nr = min(64, rows - r0);
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s_ = 0.f;
        for (int r = 0; r < nr; ++r) s_ += vs[r] * w[(size_t)(r0 + r) * cols + c];
        part2[(size_t)blk * cols + c] = s_;
    }
}

// finish for the fused form: part2 holds UNNORMALISED chunk sums; s = 1 / max(|vraw|, eps) is applied here
__device__ __forceinline__ void wdg_sn_finish_unnorm_block(const float* __restrict__ part1, int nparts, const float* __restrict__ part2,
                                                           int nchunks, int cols, float* u, float* inv_sigma) {
    __shared__ float red[1024];
    __shared__ float s_norm2, s_vscale;
    if (threadIdx.x == 0) {
        float n2 = 0.f;
        for (int i = 0; i < nparts; ++i) n2 += part1[i];   // fixed order
        s_vscale = 1.f / sqrtf(fmaxf(n2, 1e-12f));         // tf.math.l2_normalize of v
    }
    __syncthreads();
    const float vscale = s_vscale;
    float local = 0.f;
    for (int c = threadIdx.x; c < cols; c += 1024) {
        float s_ = 0.f;
        for (int k = 0; k < nchunks; ++k) s_ += part2[(size_t)k * cols + c];
        s_ *= vscale;
        u[c] = s_;  // u_raw, normalised below
        local += s_ * s_;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) s_norm2 = red[0];
    __syncthreads();
    const float n2 = s_norm2;
    const float sc = 1.f / sqrtf(fmaxf(n2, 1e-12f));
    for (int c = threadIdx.x; c < cols; c += 1024) u[c] = u[c] * sc;
    if (threadIdx.x == 0) inv_sigma[0] = 1.f / (n2 * sc);
}

// ---- SN step 3 (one block of 1024): u_raw, its norm, sigma; writes u and 1/sigma ------------------
__device__ __forceinline__ void wdg_sn_finish_block(const float* __restrict__ part2, int nchunks, int cols, float* u,
                                                    float* inv_sigma) {
    __shared__ float red[1024];
    __shared__ float s_norm2;
    // each thread owns columns t, t+1024, ...
    float local = 0.f;
    for (int c = threadIdx.x; c < cols; c += 1024) {
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part2[(size_t)k * cols + c];
        u[c] = s;  // u_raw, normalised below
        local += s * s;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) s_norm2 = red[0];
    __syncthreads();
    const float n2 = s_norm2;
    const float sc = 1.f / sqrtf(fmaxf(n2, 1e-12f));
    for (int c = threadIdx.x; c < cols; c += 1024) u[c] = u[c] * sc;
    // sigma = <u_raw, u_new> = n2 * sc
    if (threadIdx.x == 0) inv_sigma[0] = 1.f / (n2 * sc);
}

__global__ void __launch_bounds__(256) wdg_sn_rowdot_kernel(const float* __restrict__ w,
                                                            const float* __restrict__ u, int rows, int cols,
                                                            float* vraw, float* part1) {
    wdg_sn_rowdot_block(blockIdx.x, w, u, rows, cols, vraw, part1);
}
__global__ void __launch_bounds__(256) wdg_sn_colpart_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ vraw,
                                                             const float* __restrict__ part1, int nb1, int rows,
                                                             int cols, float* part2) {
    wdg_sn_colpart_block(blockIdx.x, w, vraw, part1, nb1, rows, cols, part2);
}
__global__ void __launch_bounds__(1024) wdg_sn_finish_kernel(const float* __restrict__ part2, int nchunks,
                                                             int cols, float* u, float* inv_sigma) {
    wdg_sn_finish_block(part2, nchunks, cols, u, inv_sigma);
}

__global__ void __launch_bounds__(256) wdg_sn_chunk_kernel(const float* __restrict__ w, const float* __restrict__ u, int rows,
                                                           int cols, float* vraw, float* part1, float* part2) {
    wdg_sn_chunk_block(blockIdx.x, w, u, rows, cols, vraw, part1, part2);
}
__global__ void __launch_bounds__(1024) wdg_sn_finish2_kernel(const float* __restrict__ part1, const float* __restrict__ part2,
                                                              int nchunks, int cols, float* u, float* inv_sigma) {
    wdg_sn_finish_unnorm_block(part1, nchunks, part2, nchunks, cols, u, inv_sigma);
}

__global__ void __launch_bounds__(256) wdg_scale_inplace_kernel(float* w, int64_t n, const float* __restrict__ s) {
    const float k = s[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) w[i] *= k;
}

extern "C" size_t wdg_sn_scratch_floats(int rows, int cols) {
    const size_t nb1 = (rows + 31) / 32, nchunks = (rows + 63) / 64;
    return (size_t)rows + nb1 + nchunks * (size_t)cols + 8;
}

extern "C" int wdg_sn_power_iter(float* w, float* u, int rows, int cols, float* scratch, wdg_stream stream) {
    WDG_CHECK_ARG(w && u && scratch && rows > 0 && cols > 0, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int nb1 = (rows + 31) / 32, nchunks = (rows + 63) / 64;
    float* vraw = scratch;
    float* part1 = vraw + rows;
    float* part2 = part1 + nb1;
    float* inv_sigma = part2 + (size_t)nchunks * cols;
    // (same arithmetic as the batched path: u W^T and the chunk sums of v W from one pass, v's norm applied in the finish)
    hipLaunchKernelGGL(wdg_sn_chunk_kernel, dim3(nchunks), dim3(256), 0, st, w, u, rows, cols, vraw, part1, part2);
    WDG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wdg_sn_finish2_kernel, dim3(1), dim3(1024), 0, st, part1, part2, nchunks, cols, u, inv_sigma);
    WDG_LAUNCH_CHECK();
    const int64_t n = (int64_t)rows * cols;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(wdg_scale_inplace_kernel, dim3(blocks), dim3(256), 0, st, w, n, inv_sigma);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// ---- Batched weight preparation of one network ----------------------------------------------------
// Every training-mode forward of the reference first runs the SN power iteration of ALL its wrapped layers
// (w <- w / sigma, u updated in place) and then needs the kernel-layout copies of the changed weights.  Layer
// by layer that is 5 tiny launches per layer (~400 per train step, a third of all launches, each followed by
// the ~5 us inter-kernel drain).  A batch object holds the layer table on the device; one launch per STAGE
// covers all layers (blockIdx -> layer through block-offset prefixes).  Per-layer arithmetic and summation
// order are exactly those of wdg_sn_power_iter, so results are bit-identical to the layer-by-layer path and
// across data-parallel ranks.
struct WdgPrepLayer {
    float* w;
    float* u;
    float* wF;
    float* wD;
    long long s_off;        // scratch offset of this layer (floats)
    int rows, cols;         // SN matrix view
    int taps, cin, cout;    // packing geometry
    int sn;
    int nb1, nchunks;
    int b1, b2, b3, b4, b5; // first block of this layer in the rowdot / colpart / finish / scale / pack grids
    int nb4, nb5;
};
struct wdg_prep_batch {
    std::vector<WdgPrepLayer> h;
    WdgPrepLayer* d = nullptr;
    int n = 0, n_sn = 0;
    int g1 = 0, g2 = 0, g4 = 0;      // SN-stage grids (SN layers are sorted first)
    int g5_sn = 0, g5_all = 0;       // pack grids: SN layers only / every layer
    size_t scratch_floats = 0;
};

__device__ __forceinline__ int wdg_prep_find(const WdgPrepLayer* L, int n, int blk, int which) {
    int l = 0;
    for (int i = 1; i < n; ++i) {
        const int b = which == 1 ? L[i].b1 : which == 2 ? L[i].b2 : which == 4 ? L[i].b4 : L[i].b5;
        if (blk >= b) l = i;
    }
    return l;
}

__global__ void __launch_bounds__(256) wdg_prep_chunk_kernel(const WdgPrepLayer* __restrict__ L, int n, float* scratch) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 2);
    const WdgPrepLayer q = L[l];
    float* vraw = scratch + q.s_off;
    float* part1 = vraw + q.rows;            // [nchunks] here (the slot is sized for nb1 >= nchunks entries)
    wdg_sn_chunk_block(blockIdx.x - q.b2, q.w, q.u, q.rows, q.cols, vraw, part1, part1 + q.nb1);
}
__global__ void __launch_bounds__(1024) wdg_prep_finish2_kernel(const WdgPrepLayer* __restrict__ L, float* scratch) {
    const WdgPrepLayer q = L[blockIdx.x];
    float* part1 = scratch + q.s_off + q.rows;
    float* part2 = part1 + q.nb1;
    wdg_sn_finish_unnorm_block(part1, q.nchunks, part2, q.nchunks, q.cols, q.u, part2 + (size_t)q.nchunks * q.cols);
}
__global__ void __launch_bounds__(256) wdg_prep_rowdot_kernel(const WdgPrepLayer* __restrict__ L, int n, float* scratch) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 1);
    const WdgPrepLayer q = L[l];
    float* vraw = scratch + q.s_off;
    wdg_sn_rowdot_block(blockIdx.x - q.b1, q.w, q.u, q.rows, q.cols, vraw, vraw + q.rows);
}
__global__ void __launch_bounds__(256) wdg_prep_colpart_kernel(const WdgPrepLayer* __restrict__ L, int n, float* scratch) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 2);
    const WdgPrepLayer q = L[l];
    float* vraw = scratch + q.s_off;
    float* part1 = vraw + q.rows;
    wdg_sn_colpart_block(blockIdx.x - q.b2, q.w, vraw, part1, q.nb1, q.rows, q.cols, part1 + q.nb1);
}
__global__ void __launch_bounds__(1024) wdg_prep_finish_kernel(const WdgPrepLayer* __restrict__ L, float* scratch) {
    const WdgPrepLayer q = L[blockIdx.x];
    float* part2 = scratch + q.s_off + q.rows + q.nb1;
    wdg_sn_finish_block(part2, q.nchunks, q.cols, q.u, part2 + (size_t)q.nchunks * q.cols);
}
__global__ void __launch_bounds__(256) wdg_prep_scale_kernel(const WdgPrepLayer* __restrict__ L, int n, const float* scratch) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 4);
    const WdgPrepLayer q = L[l];
    const float k = scratch[q.s_off + q.rows + q.nb1 + (size_t)q.nchunks * q.cols];
    const long long tot = (long long)q.rows * q.cols;
    for (long long i = (long long)(blockIdx.x - q.b4) * 256 + threadIdx.x; i < tot; i += (long long)q.nb4 * 256) q.w[i] *= k;
}
// master HWIO -> wF [Cout][taps][Cin_p] and wD [taps][Cin][Cout_p] (same maps as wdg_weight_pack_kernel)
__global__ void __launch_bounds__(256) wdg_prep_pack_kernel(const WdgPrepLayer* __restrict__ L, int n) {
    const int l = wdg_prep_find(L, n, blockIdx.x, 5);
    const WdgPrepLayer q = L[l];
    const int Cin_p = (q.cin + 3) & ~3, Cout_p = (q.cout + 3) & ~3;
    const long long nF = q.wF ? (long long)q.cout * q.taps * Cin_p : 0;
    const long long nD = q.wD ? (long long)q.taps * q.cin * Cout_p : 0;
    for (long long idx = (long long)(blockIdx.x - q.b5) * 256 + threadIdx.x; idx < nF + nD; idx += (long long)q.nb5 * 256) {
        if (idx < nF) {
            const int ci = (int)(idx % Cin_p);
            const long long r = idx / Cin_p;
            const int tap = (int)(r % q.taps);
            const int co = (int)(r / q.taps);
            q.wF[idx] = ci < q.cin ? q.w[((long long)tap * q.cin + ci) * q.cout + co] : 0.f;
        } else {
            const long long j = idx - nF;
            const int co = (int)(j % Cout_p);
            const long long r = j / Cout_p;  // tap*Cin + ci
            q.wD[j] = co < q.cout ? q.w[r * q.cout + co] : 0.f;
        }
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
    int b1 = 0, b2 = 0, b4 = 0, b5 = 0;
    for (int idx : order) {
        const wdg_prep_layer& s = layers[idx];
        if (!s.w || s.taps <= 0 || s.cin <= 0 || s.cout <= 0 || (s.sn && (!s.u || s.rows <= 0 || s.cols <= 0))) {
            delete b;
            wdg_set_error("wdg_prep_batch_create: bad layer %d", idx);
            return WDG_ERR_ARG;
        }
        WdgPrepLayer q;
        memset(&q, 0, sizeof(q));
        q.w = s.w; q.u = s.u; q.wF = s.wF; q.wD = s.wD;
        q.rows = s.rows; q.cols = s.cols; q.taps = s.taps; q.cin = s.cin; q.cout = s.cout; q.sn = s.sn != 0;
        q.s_off = soff;
        q.b1 = b1; q.b2 = b2; q.b3 = b->n_sn; q.b4 = b4; q.b5 = b5;
        if (q.sn) {
            q.nb1 = (q.rows + 31) / 32;
            q.nchunks = (q.rows + 63) / 64;
            const long long tot = (long long)q.rows * q.cols;
            q.nb4 = (int)std::max<long long>(1, std::min<long long>((tot + 1023) / 1024, 512));
            soff += (long long)wdg_sn_scratch_floats(q.rows, q.cols);
            b1 += q.nb1; b2 += q.nchunks; b4 += q.nb4;
            b->n_sn++;
        }
        const long long Cin_p = (q.cin + 3) & ~3, Cout_p = (q.cout + 3) & ~3;
        const long long np = (q.wF ? (long long)q.cout * q.taps * Cin_p : 0) + (q.wD ? (long long)q.taps * q.cin * Cout_p : 0);
        q.nb5 = np ? (int)std::max<long long>(1, std::min<long long>((np + 1023) / 1024, 512)) : 0;
        b5 += q.nb5;
        if (q.sn) b->g5_sn = b5;
        b->h.push_back(q);
    }
    b->n = n;
    b->g1 = b1; b->g2 = b2; b->g4 = b4; b->g5_all = b5;
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
        // [u W^T per row chunk + the chunk's share of v W, one pass over W] -> [norms, u, 1/sigma] -> [w / sigma] -> [repack]
        WDG_CHECK_ARG(scratch, "scratch required");
        hipLaunchKernelGGL(wdg_prep_chunk_kernel, dim3(b->g2), dim3(256), 0, st, b->d, b->n_sn, scratch);
        hipLaunchKernelGGL(wdg_prep_finish2_kernel, dim3(b->n_sn), dim3(1024), 0, st, b->d, scratch);
        WDG_LAUNCH_CHECK();
    }
    if (sn) {
        hipLaunchKernelGGL(wdg_prep_scale_kernel, dim3(b->g4), dim3(256), 0, st, b->d, b->n_sn, scratch);
        WDG_LAUNCH_CHECK();
    }
    const int g5 = (flags & WDG_PREP_PACK_ALL) ? b->g5_all : (sn ? b->g5_sn : 0);
    if (g5 > 0) {
        hipLaunchKernelGGL(wdg_prep_pack_kernel, dim3(g5), dim3(256), 0, st, b->d, (flags & WDG_PREP_PACK_ALL) ? b->n : b->n_sn);
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
