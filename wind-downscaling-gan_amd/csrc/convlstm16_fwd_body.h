// convlstm16_fwd_body.h — body of the 16-feature forward step (convlstm16.hip), included into the kernels that run it: the kernel parameter `p`
// (WdgLstm16, read through the kernel-argument segment: as an argument of a device function it is copied to private memory —
// 168 registers + 232 bytes of scratch instead of 84 + 0) and `int bid` (the tile index) are in scope.

    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    f32x4* lds_a = smem;                  // [4 kg][208 pixels]
    f32x4* lds_w = smem + 4 * L_NPIX;     // [9 taps][4 kg][64 gate columns]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h, img = bid / p.tiles_h;
    const int oy0 = ty * L_TH, ox0 = tx * L_TW;
    const float* Aimg = p.A + (long long)img * p.imgStrideA;

    // ---- requests in the order of first use: halo of h_{t-1}, weights, then the tile's own operands.  ALL of them branch-free
    // (buffer loads; padding and the ragged edge get bit 31 of their offset set arithmetically -> out of the descriptor's
    // range -> zeros): as `if (inside) v = load` / `inside ? load : 0` the compiler emitted exec-masked blocks with two full
    // s_waitcnt vmcnt(0) between them — three round trips in sequence where this chain of launches can afford one.
    const HaloSlots hs = l_halo_slots(t, oy0 - 1, ox0 - 1, p.H, p.W);
    const long long pimg = (long long)img * p.H * p.W;
    const wdg_srd srdA = wdg_make_srd(Aimg), srdG = wdg_make_srd(p.gates + pimg * 64), srdC = wdg_make_srd(p.c_prev + pimg * p.ldc);
    f32x4 hv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        hv[u] = wdg_buffer_load_f32x4(srdA, l_halo_byte_off(hs.off[u], p.ldA));
    f32x4 wv[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) wv[u] = p.Wl[u * 256 + t];
    const int oy = oy0 + wave;
    f32x4 old[2][4], cprev[2];
    bool ok[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ox = ox0 + a * 16 + li;
        ok[a] = oy < p.H && ox < p.W;
        const unsigned bad = (unsigned)((p.H - 1 - oy) | (p.W - 1 - ox)) & 0x80000000u;
        const int pl = oy * p.W + ox;
#pragma unroll
        for (int b = 0; b < 4; ++b) old[a][b] = wdg_buffer_load_f32x4(srdG, (unsigned)((pl * 64 + b * 16 + 4 * lg) * 4) | bad);
        cprev[a] = wdg_buffer_load_f32x4(srdC, (unsigned)((pl * p.ldc + 4 * lg) * 4) | bad);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) lds_w[u * 256 + t] = wv[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        lds_a[hs.lds[u]] = hv[u];
    __syncthreads();

#pragma unroll
    for (int a = 0; a < 2; ++a) {
        f32x4 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int th = tap / 3, tw = tap % 3;
            const f32x4 af = lds_a[lg * L_NPIX + (wave + th) * L_HW + a * 16 + li + tw];
            f32x4 bf[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = lds_w[(tap * 4 + lg) * 64 + b * 16 + li];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][j], af[j], acc[b], 0, 0, 0);
        }
        // ---- this fragment's epilogue: complete pre-activations (kept for the backward pass), cell update (Keras hard_sigmoid /
        // tanh: c = f c_prev + i c~, h = o tanh(c) — the arithmetic of wdg_lstm_fwd, pointwise.hip)
        if (ok[a]) {
            const long long pix = pimg + (long long)oy * p.W + ox0 + a * 16 + li;
            f32x4 v[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                v[b] = acc[b] + old[a][b];
                *reinterpret_cast<f32x4*>(p.gates + pix * 64 + b * 16 + 4 * lg) = v[b];
            }
            f32x4 cn, hn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cn[r] = l_hs(v[1][r]) * cprev[a][r] + l_hs(v[0][r]) * wdg_tanh(v[2][r]);
                hn[r] = l_hs(v[3][r]) * wdg_tanh(cn[r]);
            }
            *reinterpret_cast<f32x4*>(p.c_out + pix * p.ldc + 4 * lg) = cn;
            *reinterpret_cast<f32x4*>(p.h_out + pix * p.ldh + 4 * lg) = hn;
        }
    }
