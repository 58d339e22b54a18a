// convlstm2.h — the two-feature ConvLSTM recurrent steps (models.py:93 at n_timesteps > 1) as device functions of a pixel index:
// launched on their own (conv_halo.hip: wdg_convlstm_step / wdg_convlstm_bwd_step) and as extra workgroups of the 16-feature layer's
// step launches (convlstm16.hip: wdg_convlstm16_pair_step / _pair_bwd_step — one launch per timestep for BOTH recurrent layers).
#pragma once
#include "common.h"

typedef float wdg_f32x2 __attribute__((ext_vector_type(2)));

// forward arguments of one step of the two-feature layer
struct WdgCl2F {
    const float* h_prev; int ldx; long long imgStrideX;
    const float* wF; float* gates; const float* c_prev; float* c_out; int ldc; float* h_out; int ldh; int n_img, H, W;
};
// backward arguments
struct WdgCl2B {
    const float* dg_next; const float* wD; float* dh_prev; int ldx; long long imgStrideX;
    const float* gates_t; const float* c_prev; const float* c_cur; const float* dc_in; float* dgates_out; float* dc_out; int ldc;
    int n_img, H, W;
};

__device__ __forceinline__ void wdg_convlstm2_fwd_body(long long idx, const float* __restrict__ h_prev, int ldx, long long imgStrideX,
                                                                const float* __restrict__ wF, float* gates,
                                                                const float* __restrict__ c_prev, float* c_out, int ldc,
                                                                float* h_out, int ldh, int n_img, int H, int W) {
    const long long P = (long long)n_img * H * W;
    if (idx >= P) return;
    const int img = (int)(idx / ((long long)H * W));
    const int rem = (int)(idx - (long long)img * H * W);
    const int y = rem / W, x = rem - y * W;
    float* gp = gates + idx * 8;
    f32x4 z0 = *reinterpret_cast<const f32x4*>(gp), z1 = *reinterpret_cast<const f32x4*>(gp + 4);
    float z[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
    const float* hb = h_prev + (long long)img * imgStrideX;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            wdg_f32x2 hv = (wdg_f32x2){0.f, 0.f};
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                hv = *reinterpret_cast<const wdg_f32x2*>(hb + ((long long)yy * W + xx) * ldx);
            const int tap = dy * 3 + dx;
#pragma unroll
            for (int g = 0; g < 8; ++g) z[g] = fmaf(hv[1], wF[(g * 9 + tap) * 4 + 1], fmaf(hv[0], wF[(g * 9 + tap) * 4], z[g]));
        }
    *reinterpret_cast<f32x4*>(gp) = (f32x4){z[0], z[1], z[2], z[3]};
    *reinterpret_cast<f32x4*>(gp + 4) = (f32x4){z[4], z[5], z[6], z[7]};
    auto hs = [](float v) { return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f); };
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float cn = hs(z[2 + f]) * c_prev[idx * ldc + f] + hs(z[f]) * wdg_tanh(z[4 + f]);
        c_out[idx * ldc + f] = cn;
        h_out[idx * ldh + f] = hs(z[6 + f]) * wdg_tanh(cn);
    }
}


__device__ __forceinline__ void wdg_convlstm2_bwd_body(long long idx, const float* __restrict__ dg_next, const float* __restrict__ wD,
                                                                float* dh_prev, int ldx, long long imgStrideX,
                                                                const float* __restrict__ gates_t, const float* __restrict__ c_prev,
                                                                const float* __restrict__ c_cur, const float* __restrict__ dc_in,
                                                                float* dgates_out, float* dc_out, int ldc, int n_img, int H, int W) {
    const long long P = (long long)n_img * H * W;
    if (idx >= P) return;
    const int img = (int)(idx / ((long long)H * W));
    const int rem = (int)(idx - (long long)img * H * W);
    const int y = rem / W, x = rem - y * W;
    float* dhp = dh_prev + (long long)img * imgStrideX + ((long long)y * W + x) * ldx;
    float dh[2] = {dhp[0], dhp[1]};
    const float* db = dg_next + (long long)img * H * W * 8;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int qy = y - (dy - 1), qx = x - (dx - 1);     // the output pixel whose tap (dy, dx) read this pixel
            if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W) {
                const float* q = db + ((long long)qy * W + qx) * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
                const int tap = dy * 3 + dx;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const float* w = wD + (tap * 2 + ci) * 8;
                    dh[ci] += (a[0] * w[0] + a[1] * w[1]) + (a[2] * w[2] + a[3] * w[3]) + (b[0] * w[4] + b[1] * w[5]) + (b[2] * w[6] + b[3] * w[7]);
                }
            }
        }
    dhp[0] = dh[0];
    dhp[1] = dh[1];
    auto hs = [](float v) { return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f); };
    auto hsg = [](float v) { const float u = 0.2f * v + 0.5f; return (u >= 0.f && u <= 1.f) ? 0.2f : 0.f; };
    const float* g = gates_t + idx * 8;
    float* dg = dgates_out + idx * 8;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float xi = g[f], xf = g[2 + f], xc = g[4 + f], xo = g[6 + f];
        const float gi = hs(xi), gf = hs(xf), gc = wdg_tanh(xc), go = hs(xo);
        const float cp = c_prev ? c_prev[idx * ldc + f] : 0.f;
        const float tc = wdg_tanh(c_cur[idx * ldc + f]);
        const float dc = dh[f] * go * (1.f - tc * tc) + dc_in[idx * ldc + f];
        dg[f] = dc * gc * hsg(xi);
        dg[2 + f] = dc * cp * hsg(xf);
        dg[4 + f] = dc * gi * (1.f - gc * gc);
        dg[6 + f] = dh[f] * tc * hsg(xo);
        if (dc_out) dc_out[idx * ldc + f] = dc * gf;
    }
}
