// convlstm16_bwd_body.h — body of the 16-feature backward step (convlstm16.hip), included into the kernels that run it: the kernel parameter `p`
// (WdgLstm16, read through the kernel-argument segment: as an argument of a device function it is copied to private memory —
// 168 registers + 232 bytes of scratch instead of 84 + 0) and `int bid` (the tile index) are in scope.

    extern __shared__ __attribute__((aligned(16))) f32x4 smem[];
    f32x4* lds_a = smem;                  // [4 kg][208 pixels] of the current 16-channel slice of dgates_t
    f32x4* lds_w = smem + 4 * L_NPIX;     // [4 slices][9 taps][4 kg][16 features]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    const int tx = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int ty = bid % p.tiles_h, img = bid / p.tiles_h;
    const int oy0 = ty * L_TH, ox0 = tx * L_TW;
    const float* Aimg = p.A + (long long)img * p.imgStrideA;

    // (every request branch-free, see the forward kernel: here the compiler had put a full wait between the request of the
    // next dgates slice and the MFMAs that were meant to cover it)
    const HaloSlots hs = l_halo_slots(t, oy0 - 1, ox0 - 1, p.H, p.W);
    const long long pimg = (long long)img * p.H * p.W;
    const wdg_srd srdA = wdg_make_srd(Aimg), srdD = wdg_make_srd(p.dh_prev + (long long)img * p.imgStrideDh),
                  srdG = wdg_make_srd(p.gates_t + pimg * 64), srdCp = wdg_make_srd((p.c_prev ? p.c_prev : p.c_cur) + pimg * p.ldc),
                  srdCc = wdg_make_srd(p.c_cur + pimg * p.ldc), srdDc = wdg_make_srd(p.dc_in + pimg * p.ldc);
    unsigned hoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        hoff[u] = l_halo_byte_off(hs.off[u], p.ldA);          // (+ 64 ck below: padding stays at 0x80000000 + 64 ck, out of range)
    f32x4 hv[4];
    auto halo_request = [&](int ck) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) hv[u] = wdg_buffer_load_f32x4(srdA, hoff[u] + ck * 64);
    };
    halo_request(0);
    f32x4 wv[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) wv[u] = p.Wl[u * 256 + t];
    const int oy = oy0 + wave;
    f32x4 old[2], bin[2][7];
    bool ok[2];
    const unsigned no_cprev = p.c_prev ? 0u : 0x80000000u;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int ox = ox0 + a * 16 + li;
        ok[a] = oy < p.H && ox < p.W;
        const unsigned bad = (unsigned)((p.H - 1 - oy) | (p.W - 1 - ox)) & 0x80000000u;
        const int pl = oy * p.W + ox;
        old[a] = wdg_buffer_load_f32x4(srdD, (unsigned)((pl * p.ld_dh + 4 * lg) * 4) | bad);
#pragma unroll
        for (int q = 0; q < 4; ++q) bin[a][q] = wdg_buffer_load_f32x4(srdG, (unsigned)((pl * 64 + q * 16 + 4 * lg) * 4) | bad);
        const unsigned co = (unsigned)((pl * p.ldc + 4 * lg) * 4) | bad;
        bin[a][4] = wdg_buffer_load_f32x4(srdCp, co | no_cprev);
        bin[a][5] = wdg_buffer_load_f32x4(srdCc, co);
        bin[a][6] = wdg_buffer_load_f32x4(srdDc, co);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) lds_w[u * 256 + t] = wv[u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        lds_a[hs.lds[u]] = hv[u];
    __syncthreads();

    f32x4 acc[2];
    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ck = 0; ck < 4; ++ck) {
        if (ck < 3) halo_request(ck + 1);          // in flight under this slice's MFMAs
        __builtin_amdgcn_sched_barrier(0);         // (... which the scheduler otherwise moves behind most of them)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // data gradient: tap (th, tw) reads dgates at (y + 1 - th, x + 1 - tw) -> halo row wave + 2 - th, column + 2 - tw
            const int th = tap / 3, tw = tap % 3;
            const f32x4 bf = lds_w[((ck * 9 + tap) * 4 + lg) * 16 + li];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 af = lds_a[lg * L_NPIX + (wave + 2 - th) * L_HW + a * 16 + li + 2 - tw];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[j], acc[a], 0, 0, 0);
            }
        }
        if (ck < 3) {
            __syncthreads();                       // every wave is done with this slice
#pragma unroll
            for (int u = 0; u < 4; ++u)
                lds_a[hs.lds[u]] = hv[u];
            __syncthreads();
        }
    }

    // ---- epilogue: complete dh_{t-1}, then the cell backward (the arithmetic of wdg_lstm_bwd, pointwise.hip)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (!ok[a]) continue;
        const long long pl = (long long)oy * p.W + ox0 + a * 16 + li, pix = pimg + pl;
        const f32x4 dhv = acc[a] + old[a];
        *reinterpret_cast<f32x4*>(p.dh_prev + (long long)img * p.imgStrideDh + pl * p.ld_dh + 4 * lg) = dhv;
        const f32x4 xi = bin[a][0], xf = bin[a][1], xc = bin[a][2], xo = bin[a][3], cp = bin[a][4], cc = bin[a][5], dci = bin[a][6];
        f32x4 di, df, dcc, dob, dcp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gi = l_hs(xi[r]), gf = l_hs(xf[r]), gc = wdg_tanh(xc[r]), go = l_hs(xo[r]);
            const float tc = wdg_tanh(cc[r]);
            const float dc = dhv[r] * go * (1.f - tc * tc) + dci[r];
            di[r] = dc * gc * l_hsg(xi[r]);
            df[r] = dc * cp[r] * l_hsg(xf[r]);
            dcc[r] = dc * gi * (1.f - gc * gc);
            dob[r] = dhv[r] * tc * l_hsg(xo[r]);
            dcp[r] = dc * gf;
        }
        float* dg = p.dgates_out + pix * 64 + 4 * lg;
        *reinterpret_cast<f32x4*>(dg) = di;
        *reinterpret_cast<f32x4*>(dg + 16) = df;
        *reinterpret_cast<f32x4*>(dg + 32) = dcc;
        *reinterpret_cast<f32x4*>(dg + 48) = dob;
        if (p.dc_out) *reinterpret_cast<f32x4*>(p.dc_out + pix * p.ldc + 4 * lg) = dcp;
    }
