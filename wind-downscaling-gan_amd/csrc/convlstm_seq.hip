// convlstm_seq.hip — ConvLSTM2D over a whole sequence (n_timesteps > 1) in ONE persistent launch, for the few-channel
// full-resolution layers of the discriminator (/root/reference/src/downscaling/gan/models.py:93,101: 2 -> 2 and 5 -> 16
// features; Keras ConvLSTM2D: gates i,f,c,o = conv(x_t, K) + b + conv(h_{t-1}, R), hard-sigmoid / tanh, h_0 = c_0 = 0).
//
// Launched step by step that recurrence is 2 tiny kernels per timestep (a 3x3 convolution over 16 channels of a
// 96 x 96 x 8 map and the cell update: 40 + 6 us, a tenth of the machine) — 2,300 launches per train step at the shipped
// sequence length of 24.  Here one workgroup owns a 4 x 32 pixel tile of one image for ALL timesteps:
//   * per step it stages the x_t halo and the h_{t-1} halo (6 x 34 pixels) in LDS, forms the four gates of its pixels
//     (thread = pixel x feature half; packed fp32 FMAs with wave-uniform weights, as convlstm1.hip), updates the cell and
//     writes h_t (+ the gate pre-activations and c_t the backward pass wants);
//   * the only inter-workgroup dependence is the one-pixel ring of h_{t-1} that belongs to the (up to 8) neighbouring tiles:
//     each tile publishes "step t done" in a per-tile counter (stores -> every wave drains -> barrier -> agent-scope
//     release -> counter), a consumer polls its neighbours' counters with relaxed agent-scope loads, then one agent-scope
//     acquire, a drain and a barrier precede its plain loads (MI355X_MICROARCH.md, workgroup hand-off recipe).  Tiles are
//     128-byte-line aligned in h (32 pixels x >= 4 floats), so no line is shared between writers.
//   * the grid is sized to be fully resident (every workgroup must make progress while others wait); a workgroup that
//     owns several tiles walks them in the same order every step, which keeps the wait graph acyclic.  Every spin is
//     bounded: on a timeout the kernel raises *err and stops waiting (the host checks it).
// The backward kernel walks the sequence in reverse with the same hand-off on dgates_{t+1}: per step it pulls the
// recurrent gradient R_t = conv^T(dgates_{t+1}, R) for its pixels from the staged halo, adds the incoming dh_t, runs the
// cell backward with dc carried in registers, and publishes dgates_t.  Kernel / bias / input gradients are then ordinary
// full-sequence launches over the dgates tensor (conv_wgrad / colsum / conv_dgrad), as before.
#include "common.h"
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SQ_TH = 4, SQ_TW = 32;             // centre tile: 128 pixels, 2 threads (feature halves) per pixel
constexpr int SQ_HH = SQ_TH + 2, SQ_HW = SQ_TW + 2;
constexpr int SQ_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ float sq_tanh(float x) {
    const float ax = fabsf(x);
    const float x2 = x * x;
    const float poly = x * fmaf(x2, fmaf(x2, fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    const float e = __expf(-2.f * ax);
    const float big = copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), x);
    return ax < 0.1f ? poly : big;
}
__device__ __forceinline__ float sq_hsig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float sq_hsig_grad(float x) {
    const float v = 0.2f * x + 0.5f;
    return (v >= 0.f && v <= 1.f) ? 0.2f : 0.f;
}

struct WdgSeq {
    const float* X;      // [T*B, H, W, ldx] time-major (image n = t*B + b)
    float* Hs;           // [T*B, H, W, ldh] hidden states (forward: out; backward: unused)
    float* G;            // [T*B, H, W, 4F] gate pre-activations i|f|c|o (forward: optional out; backward: in)
    float* C;            // [T*B, H, W, F] cell states (forward: out; backward: in)
    const float* dH;     // backward: incoming gradient of every h_t [T*B, H, W, lddh]
    float* dG;           // backward: dgates out [T*B, H, W, 4F]
    int* flags;          // [B * tiles_h * tiles_w] steps completed per tile (zeroed before the launch)
    int* err;            // set to 1 when a wait timed out
    long long isX, isH, isDH;   // image strides
    int B, T, H, W, ldx, ldh, lddh;
    int tiles_h, tiles_w, ntiles;
};

// wait until every existing neighbour tile of (b, ty, tx) has completed `need` steps; one lane polls, then the
// consumer side of the hand-off (acquire -> drain -> barrier) for the whole workgroup
__device__ __forceinline__ void sq_wait_neighbours(const WdgSeq& p, int b, int ty, int tx, int need) {
    if (threadIdx.x == 0) {
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int ny = ty + dy, nx = tx + dx;
                if ((dy | dx) == 0 || ny < 0 || nx < 0 || ny >= p.tiles_h || nx >= p.tiles_w) continue;
                const int* f = p.flags + ((long long)b * p.tiles_h + ny) * p.tiles_w + nx;
                int spins = 0;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > SQ_SPIN_LIMIT) {
                        __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// producer side: every wave drains its stores, barrier, one lane releases at agent scope and bumps the tile's counter
__device__ __forceinline__ void sq_publish(const WdgSeq& p, int tile, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.flags + tile, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------------
template <int CIN, int F>
__global__ void __launch_bounds__(256) wdg_convlstm_seq_fwd_kernel(const WdgSeq p, const float* __restrict__ Wx,
                                                                   const float* __restrict__ Wh, const float* __restrict__ bias) {
    constexpr int FH = F >= 2 ? F / 2 : 1;
    constexpr int C4 = (CIN + 3) / 4;
    constexpr int HS = F | 1;                        // odd pixel stride of the h halo: per-pixel b32 reads of a wave hit distinct banks
    __shared__ __attribute__((aligned(16))) f32x4 xs[SQ_HH * SQ_HW * C4];
    __shared__ float hs[SQ_HH * SQ_HW * HS];
    const int t = threadIdx.x;
    const int half = __builtin_amdgcn_readfirstlane(t >> 7);
    const int f0 = half * FH;
    const bool half_on = !(F < 2 && half);
    const int cp = t & 127, py = cp >> 5, px = cp & 31;

    for (int step = 0; step < p.T; ++step) {
        for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
            int r = tile;
            const int tx = r % p.tiles_w;
            r /= p.tiles_w;
            const int ty = r % p.tiles_h;
            const int b = r / p.tiles_h;
            const int oy0 = ty * SQ_TH, ox0 = tx * SQ_TW;
            const long long n = (long long)step * p.B + b;
            // x_t halo (zero outside the image: the conv's 'same' padding)
            const float* Ximg = p.X + n * p.isX;
            for (int idx = t; idx < SQ_HH * SQ_HW * C4; idx += 256) {
                const int c4 = idx % C4, pix = idx / C4;
                const int hy = pix / SQ_HW, hx = pix - hy * SQ_HW;
                const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                    v = *reinterpret_cast<const f32x4*>(Ximg + ((long long)gy * p.W + gx) * p.ldx + 4 * c4);
                xs[c4 * (SQ_HH * SQ_HW) + pix] = v;
            }
            if (step > 0) {
                sq_wait_neighbours(p, b, ty, tx, step);       // their h_{step-1} is visible from here on
                const float* Hprev = p.Hs + (n - p.B) * p.isH;
                for (int idx = t; idx < SQ_HH * SQ_HW * F; idx += 256) {
                    const int ch = idx % F, pix = idx / F;
                    const int hy = pix / SQ_HW, hx = pix - hy * SQ_HW;
                    const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
                    float v = 0.f;
                    if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = Hprev[((long long)gy * p.W + gx) * p.ldh + ch];
                    hs[pix * HS + ch] = v;
                }
            }
            __syncthreads();
            const int gy = oy0 + py, gx = ox0 + px;
            if (half_on && gy < p.H && gx < p.W) {
                __attribute__((aligned(8))) float g4[4][FH];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int f = 0; f < FH; ++f) g4[g][f] = bias[g * F + f0 + f];
#pragma unroll 1
                for (int tap = 0; tap < 9; ++tap) {
                    const int hp = (py + tap / 3) * SQ_HW + px + tap % 3;
#pragma unroll
                    for (int c4 = 0; c4 < C4; ++c4) {
                        const f32x4 xv = xs[c4 * (SQ_HH * SQ_HW) + hp];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = 4 * c4 + j;
                            if (c < CIN) {
                                const float* w = Wx + (tap * CIN + c) * 4 * F + f0;
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    if constexpr (FH % 2 == 0) {
#pragma unroll
                                        for (int f = 0; f < FH; f += 2)
                                            *(f32x2*)&g4[g][f] = __builtin_elementwise_fma((f32x2){xv[j], xv[j]}, *(const f32x2*)&w[g * F + f], *(f32x2*)&g4[g][f]);
                                    } else {
#pragma unroll
                                        for (int f = 0; f < FH; ++f) g4[g][f] = fmaf(xv[j], w[g * F + f], g4[g][f]);
                                    }
                                }
                            }
                        }
                    }
                    if (step > 0) {
#pragma unroll
                        for (int ch = 0; ch < F; ++ch) {
                            const float hv = hs[hp * HS + ch];
                            const float* w = Wh + (tap * F + ch) * 4 * F + f0;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                if constexpr (FH % 2 == 0) {
#pragma unroll
                                    for (int f = 0; f < FH; f += 2)
                                        *(f32x2*)&g4[g][f] = __builtin_elementwise_fma((f32x2){hv, hv}, *(const f32x2*)&w[g * F + f], *(f32x2*)&g4[g][f]);
                                } else {
#pragma unroll
                                    for (int f = 0; f < FH; ++f) g4[g][f] = fmaf(hv, w[g * F + f], g4[g][f]);
                                }
                            }
                        }
                    }
                }
                const long long pix = (long long)gy * p.W + gx;
                float* gout = p.G ? p.G + (n * p.H * p.W + pix) * 4 * F + f0 : nullptr;
                float* cout_ = p.C + (n * p.H * p.W + pix) * F + f0;
                const float* cprev = p.C + ((n - p.B) * p.H * p.W + pix) * F + f0;
                float* hout = p.Hs + n * p.isH + pix * p.ldh + f0;
#pragma unroll
                for (int f = 0; f < FH; ++f) {
                    float c = sq_hsig(g4[0][f]) * sq_tanh(g4[2][f]);
                    if (step > 0) c += sq_hsig(g4[1][f]) * cprev[f];
                    cout_[f] = c;
                    hout[f] = sq_hsig(g4[3][f]) * sq_tanh(c);
                    if (gout) {
                        gout[f] = g4[0][f];
                        gout[F + f] = g4[1][f];
                        gout[2 * F + f] = g4[2][f];
                        gout[3 * F + f] = g4[3][f];
                    }
                }
            }
            if (step + 1 < p.T) sq_publish(p, tile, step + 1);
            else __syncthreads();
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// dgates_t for every t (written to dG); thread = (pixel, feature half); dc is carried in registers when the workgroup owns
// one tile, through LDS-free global scratch otherwise (DC, [B, H, W, F]).
template <int CIN, int F>
__global__ void __launch_bounds__(256) wdg_convlstm_seq_bwd_kernel(const WdgSeq p, const float* __restrict__ WhT, float* DC) {
    constexpr int FH = F >= 2 ? F / 2 : 1;
    constexpr int GS = (4 * F) | 1;                  // odd pixel stride of the dgates halo
    __shared__ float dgs[SQ_HH * SQ_HW * GS];
    const int t = threadIdx.x;
    const int half = __builtin_amdgcn_readfirstlane(t >> 7);
    const int f0 = half * FH;
    const bool half_on = !(F < 2 && half);
    const int cp = t & 127, py = cp >> 5, px = cp & 31;

    for (int k = 0; k < p.T; ++k) {
        const int step = p.T - 1 - k;
        for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
            int r = tile;
            const int tx = r % p.tiles_w;
            r /= p.tiles_w;
            const int ty = r % p.tiles_h;
            const int b = r / p.tiles_h;
            const int oy0 = ty * SQ_TH, ox0 = tx * SQ_TW;
            const long long n = (long long)step * p.B + b;
            const long long HW = (long long)p.H * p.W;
            if (k > 0) {
                sq_wait_neighbours(p, b, ty, tx, k);          // their dgates_{step+1} is visible from here on
                const float* Gn = p.dG + (n + p.B) * HW * 4 * F;
                for (int idx = t; idx < SQ_HH * SQ_HW * 4 * F; idx += 256) {
                    const int ch = idx % (4 * F), pix = idx / (4 * F);
                    const int hy = pix / SQ_HW, hx = pix - hy * SQ_HW;
                    const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
                    float v = 0.f;
                    if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = Gn[((long long)gy * p.W + gx) * 4 * F + ch];
                    dgs[pix * GS + ch] = v;
                }
                __syncthreads();
            }
            const int gy = oy0 + py, gx = ox0 + px;
            if (half_on && gy < p.H && gx < p.W) {
                const long long pix = (long long)gy * p.W + gx;
                __attribute__((aligned(8))) float dh[FH];
                const float* dhin = p.dH + n * p.isDH + pix * p.lddh + f0;
#pragma unroll
                for (int f = 0; f < FH; ++f) dh[f] = dhin[f];
                if (k > 0) {
                    // recurrent gradient: dh[f] += sum_tap sum_g dgates_{t+1}[pixel + (1 - th, 1 - tw)][g] * Wh[tap][f][g]
                    // (WhT [tap][g][f]: the transposed recurrent kernel, so a wave-uniform run of FH weights is one scalar load)
#pragma unroll 1
                    for (int tap = 0; tap < 9; ++tap) {
                        const float* dgp = &dgs[((py + 2 - tap / 3) * SQ_HW + px + 2 - tap % 3) * GS];
#pragma unroll 4
                        for (int g = 0; g < 4 * F; ++g) {
                            const float v = dgp[g];
                            const float* w = WhT + (tap * 4 * F + g) * F + f0;
                            if constexpr (FH % 2 == 0) {
#pragma unroll
                                for (int f = 0; f < FH; f += 2)
                                    *(f32x2*)&dh[f] = __builtin_elementwise_fma((f32x2){v, v}, *(const f32x2*)&w[f], *(f32x2*)&dh[f]);
                            } else {
#pragma unroll
                                for (int f = 0; f < FH; ++f) dh[f] = fmaf(v, w[f], dh[f]);
                            }
                        }
                    }
                }
                const float* gin = p.G + (n * HW + pix) * 4 * F + f0;
                const float* cin_ = p.C + (n * HW + pix) * F + f0;
                const float* cprev = p.C + ((n - p.B) * HW + pix) * F + f0;
                float* dcp = DC + ((long long)b * HW + pix) * F + f0;
                float* dg = p.dG + (n * HW + pix) * 4 * F + f0;
#pragma unroll
                for (int f = 0; f < FH; ++f) {
                    const float xi = gin[f], xf = gin[F + f], xc = gin[2 * F + f], xo = gin[3 * F + f];
                    const float gi = sq_hsig(xi), gf = sq_hsig(xf), gc = sq_tanh(xc), go = sq_hsig(xo);
                    const float cprev_v = step > 0 ? cprev[f] : 0.f;
                    const float tc = sq_tanh(cin_[f]);
                    float dc = dh[f] * go * (1.f - tc * tc);
                    if (k > 0) dc += dcp[f];
                    dg[f] = dc * gc * sq_hsig_grad(xi);
                    dg[F + f] = dc * cprev_v * sq_hsig_grad(xf);
                    dg[2 * F + f] = dc * gi * (1.f - gc * gc);
                    dg[3 * F + f] = dh[f] * tc * sq_hsig_grad(xo);
                    dcp[f] = dc * gf;
                }
            }
            if (k + 1 < p.T) sq_publish(p, tile, k + 1);
            else __syncthreads();
        }
    }
}

// WhT[tap][g][f] = Wh[tap][f][g]
__global__ void wdg_convlstm_seq_transpose_kernel(const float* __restrict__ Wh, float* WhT, int F) {
    const int n = 9 * F * 4 * F;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int g = i % (4 * F), f = (i / (4 * F)) % F, tap = i / (4 * F * F);
        WhT[(tap * 4 * F + g) * F + f] = Wh[i];
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
extern "C" int wdg_convlstm_seq_supported(int cin, int F) { return (cin == 2 && F == 2) || (cin == 5 && F == 16); }

static int sq_resident_blocks(const void* kernel) {
    int dev = 0, per_cu = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    // the occupancy query can over-report by one block per CU (MI355X_MICROARCH.md, residency): keep a margin; two
    // resident workgroups per CU are plenty for this kernel
    per_cu = std::max(1, std::min(per_cu - 1, 2));
    return cus * per_cu;
}

extern "C" size_t wdg_convlstm_seq_scratch_bytes(int B, int H, int W, int F) {
    const size_t tiles = (size_t)B * ((H + SQ_TH - 1) / SQ_TH) * ((W + SQ_TW - 1) / SQ_TW);
    return tiles * sizeof(int) + 64 + 256 + (size_t)B * H * W * F * sizeof(float) + (size_t)9 * 4 * F * F * sizeof(float);   // flags | err | dc | WhT
}

static int sq_fill(WdgSeq& p, int B, int T, int H, int W, int F, void* scratch, hipStream_t st) {
    p.B = B; p.T = T; p.H = H; p.W = W;
    p.tiles_h = (H + SQ_TH - 1) / SQ_TH;
    p.tiles_w = (W + SQ_TW - 1) / SQ_TW;
    p.ntiles = B * p.tiles_h * p.tiles_w;
    p.flags = (int*)scratch;
    p.err = p.flags + p.ntiles;
    WDG_HIP(hipMemsetAsync(scratch, 0, (size_t)p.ntiles * sizeof(int) + 64, st));
    return WDG_OK;
}

extern "C" int wdg_convlstm_seq_fwd(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* wh,
                                    const float* bias, float* h, int ldh, int64_t img_stride_h, float* gates, float* c,
                                    int B, int T, int H, int W, int cin, int F, void* scratch, size_t scratch_bytes,
                                    wdg_stream stream) {
    WDG_CHECK_ARG(x && wx && wh && bias && h && c && scratch, "null argument");
    WDG_CHECK_ARG(wdg_convlstm_seq_supported(cin, F), "unsupported (cin, F)");
    WDG_CHECK_ARG(((uintptr_t)x & 15) == 0 && ldx % 4 == 0 && ldx >= wdg_round_up(cin, 4) && ldh >= F, "x alignment / ld");
    WDG_CHECK_ARG((W * ldh) % 32 == 0 && ((uintptr_t)h & 127) == 0 && img_stride_h % 32 == 0,
                  "h rows must start on 128-byte lines (tiles must not share lines)");
    WDG_CHECK_ARG(scratch_bytes >= wdg_convlstm_seq_scratch_bytes(B, H, W, F), "scratch too small");
    hipStream_t st = (hipStream_t)stream;
    WdgSeq p;
    memset(&p, 0, sizeof(p));
    int rc = sq_fill(p, B, T, H, W, F, scratch, st);
    if (rc != WDG_OK) return rc;
    p.X = x; p.Hs = h; p.G = gates; p.C = c;
    p.isX = img_stride_x; p.isH = img_stride_h; p.ldx = ldx; p.ldh = ldh;
    const void* k = cin == 2 ? (const void*)&wdg_convlstm_seq_fwd_kernel<2, 2> : (const void*)&wdg_convlstm_seq_fwd_kernel<5, 16>;
    const int grid = std::min(p.ntiles, sq_resident_blocks(k));
    if (cin == 2)
        hipLaunchKernelGGL((wdg_convlstm_seq_fwd_kernel<2, 2>), dim3(grid), dim3(256), 0, st, p, wx, wh, bias);
    else
        hipLaunchKernelGGL((wdg_convlstm_seq_fwd_kernel<5, 16>), dim3(grid), dim3(256), 0, st, p, wx, wh, bias);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

extern "C" int wdg_convlstm_seq_bwd(const float* gates, const float* c, const float* wh, const float* dh, int lddh,
                                    int64_t img_stride_dh, float* dgates, int B, int T, int H, int W, int cin, int F,
                                    void* scratch, size_t scratch_bytes, wdg_stream stream) {
    WDG_CHECK_ARG(gates && c && wh && dh && dgates && scratch, "null argument");
    WDG_CHECK_ARG(wdg_convlstm_seq_supported(cin, F), "unsupported (cin, F)");
    WDG_CHECK_ARG(((uintptr_t)dgates & 127) == 0 && (W * 4 * F) % 32 == 0, "dgates rows must start on 128-byte lines");
    WDG_CHECK_ARG(scratch_bytes >= wdg_convlstm_seq_scratch_bytes(B, H, W, F), "scratch too small");
    hipStream_t st = (hipStream_t)stream;
    WdgSeq p;
    memset(&p, 0, sizeof(p));
    int rc = sq_fill(p, B, T, H, W, F, scratch, st);
    if (rc != WDG_OK) return rc;
    p.G = const_cast<float*>(gates); p.C = const_cast<float*>(c); p.dH = dh; p.dG = dgates;
    p.isDH = img_stride_dh; p.lddh = lddh;
    char* base = reinterpret_cast<char*>(scratch) + (((size_t)p.ntiles * sizeof(int) + 64 + 255) & ~(size_t)255);
    float* dc = reinterpret_cast<float*>(base);
    float* whT = dc + (size_t)B * H * W * F;
    hipLaunchKernelGGL(wdg_convlstm_seq_transpose_kernel, dim3(4), dim3(256), 0, st, wh, whT, F);
    const void* k = cin == 2 ? (const void*)&wdg_convlstm_seq_bwd_kernel<2, 2> : (const void*)&wdg_convlstm_seq_bwd_kernel<5, 16>;
    const int grid = std::min(p.ntiles, sq_resident_blocks(k));
    if (cin == 2)
        hipLaunchKernelGGL((wdg_convlstm_seq_bwd_kernel<2, 2>), dim3(grid), dim3(256), 0, st, p, whT, dc);
    else
        hipLaunchKernelGGL((wdg_convlstm_seq_bwd_kernel<5, 16>), dim3(grid), dim3(256), 0, st, p, whT, dc);
    WDG_LAUNCH_CHECK();
    return WDG_OK;
}

// 1 when a wait of the last launches on this scratch timed out (results invalid); resets the flag
extern "C" int wdg_convlstm_seq_check(void* scratch, int B, int H, int W, wdg_stream stream) {
    const size_t tiles = (size_t)B * ((H + SQ_TH - 1) / SQ_TH) * ((W + SQ_TW - 1) / SQ_TW);
    int v = 0;
    if (hipMemcpyAsync(&v, (int*)scratch + tiles, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return -1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return v;
}
