// conv_plan.h — host-side convolution plan shared by the implicit-GEMM and halo-tile kernels.
#pragma once
#include "common.h"
#include <vector>

struct WdgPhase {
    int Pa, Pb;            // output sub-grid of this phase (rows, cols per image)
    int a_off_h, a_off_w;  // A coord = pa * a_mul + a_off + tap displacement
    int o_off_h, o_off_w;  // Out coord = pa * o_mul + o_off
    int K4;                // k4 groups (padded to a multiple of 8 with invalid entries)
    int tab_off;           // first table entry of this phase
    wdg_fastdiv div_papb, div_pb;   // row index -> (image, pa, pb) by multiply-high (wdg_phase_finish fills them)
    // 2-D row tiles (t2_w > 0): the BM rows of a tile are a t2_h x t2_w patch of output pixels instead of BM consecutive
    // pixels of one output row — fewer distinct input rows per tile, so neighbouring taps hit L1 / L2 instead of the fabric.
    // t2_h * t2_w <= BM: a patch that does not fill the tile (6 x 21 = 126 of 128 rows on an 84 x 84 map) leaves idle rows,
    // the row space of the launch is then tiles * BM ("virtual" rows), not the pixel count
    int t2_w, t2_h, t2_rows, t2_tiles_w, t2_tpi;
    wdg_fastdiv div_t2_img, div_t2_w, div_t2_ml;   // tile index -> (image, tile row, tile column); row in tile -> (ly, lx)
};
static inline void wdg_phase_finish(WdgPhase& ph) {
    ph.div_papb = wdg_fastdiv_make((unsigned)(ph.Pa * ph.Pb > 0 ? ph.Pa * ph.Pb : 1));
    ph.div_pb = wdg_fastdiv_make((unsigned)(ph.Pb > 0 ? ph.Pb : 1));
    ph.t2_w = ph.t2_h = ph.t2_rows = ph.t2_tiles_w = ph.t2_tpi = 0;
    ph.div_t2_img = ph.div_t2_w = ph.div_t2_ml = wdg_fastdiv_make(1u);
}
// enable 2-D row tiles for tiles of bm rows: the t2_h x t2_w patch (t2_h | Pa, t2_w | Pb) with the most pixels that fits bm
// rows, if it fills at least 97 % of them; among equals the one with the smallest perimeter, wider than tall (a pixel row of a
// patch is one contiguous run of the output).  Returns the virtual row count of the phase per image (tiles * bm), 0 = linear.
static inline int wdg_phase_tile2d(WdgPhase& ph, int bm) {
    int best_h = 0, best_w = 0;
    for (int h = 1; h <= ph.Pa && h <= bm; ++h) {
        if (ph.Pa % h) continue;
        for (int w = 1; w <= ph.Pb && h * w <= bm; ++w) {
            if (ph.Pb % w) continue;
            const int a = h * w, b = best_h * best_w;
            if (a > b || (a == b && (h + w < best_h + best_w || (h + w == best_h + best_w && w > best_w)))) { best_h = h; best_w = w; }
        }
    }
    if (best_h * best_w * 100 < bm * 97 || best_h < 2) return 0;
    ph.t2_h = best_h; ph.t2_w = best_w; ph.t2_rows = best_h * best_w;
    ph.t2_tiles_w = ph.Pb / best_w;
    ph.t2_tpi = (ph.Pa / best_h) * (ph.Pb / best_w);
    ph.div_t2_img = wdg_fastdiv_make((unsigned)ph.t2_tpi);
    ph.div_t2_w = wdg_fastdiv_make((unsigned)ph.t2_tiles_w);
    ph.div_t2_ml = wdg_fastdiv_make((unsigned)best_w);
    return ph.t2_tpi * bm;
}

struct wdg_conv_plan {
    wdg_conv_geom g;
    int Cin_p, Cout_p, taps;
    int cus;
    // > 0 (wdg_conv_plan_create_sliced): the plan covers a RANGE of output channels of a layer whose HWIO weight tensor has w_ld of
    // them; the data- and weight-gradient entry points then address that tensor (pointers offset to the range's first channel) with
    // this channel stride instead of the plan's own channel count
    int w_ld = 0;
    // forward
    int4* d_tab_fwd = nullptr;
    int2* d_wrow = nullptr;
    int K4_fwd = 0;  // padded to 8
    // dgrad
    int4* d_tab_dgrad = nullptr;
    std::vector<WdgPhase> ph_dgrad;
    int K4_dgrad_max = 0;
    size_t ws_bytes = 0;
    // launch configs (chosen at creation)
    int fwd_split = 1, dgrad_split = 1, wgrad_split = 1;
    // halo-tile kernel (stride 1, few output channels): per-tap tables {dh, dw, b_off0, 0}
    int4* d_taps_fwd = nullptr;
    int4* d_taps_dgrad = nullptr;
    int halo_fwd_nt = 0, halo_dgrad_nt = 0;   // 0 = not eligible; else 16-column tiles per block (1, 2, 4)
    int halo_auto = 0;                        // conv_fwd / conv_dgrad dispatch to the halo kernel by themselves ...
    int halo_auto_fwd = 0, halo_auto_dgrad = 0;   // ... per direction (shallow reductions only)
};


// dgrad_patch_s3.hip
void wdg_dgrad_s3_set(int v);
bool wdg_dgrad_s3_ok(const wdg_conv_plan* pl, int c0, int C, int ldy_act);
size_t wdg_dgrad_s3_ws_bytes(const wdg_conv_plan* pl);
#if defined(__HIPCC__)
int wdg_dgrad_s3_launch(const wdg_conv_plan* pl, const float* dy, const float* wD, float* dx, const float* y, int ldy_act,
                        int64_t img_stride_act, const float* mean_rstd, const float* gamma, int c0, int C, float act_slope, float* dgamma, float* dbeta, float* dbias,
                        void* ws, hipStream_t stream);
#endif

// conv_halo.hip
int wdg_halo_plan_init(wdg_conv_plan* pl);
void wdg_halo_plan_free(wdg_conv_plan* pl);
void wdg_halo_set_wg(int v);
void wdg_upconv_set_gather_xcd(int v);
void wdg_bn_set_bwd_blocks(int v);
void wdg_halo_set_persistent(int v);
void wdg_halo_set_stage(int v);
void wdg_halo_bf16_set_thin(int v);   // conv_halo_bf16.hip: the specialised 16 -> (<= 4) output-conv kernel on / off
void wdg_halo_set_th4(int v);
void wdg_halo_set_max_cin(int v);
void wdg_h16_set_small_tiles(int v);   // conv_igemm_bf16.hip
void wdg_h16_set_lstm_fused(int v);
void wdg_patch_h16_set_lstm_small(int v);
void wdg_lstm16_set_step(int v);       // convlstm16.hip
void wdg_cl2_set_thin(int v);          // convlstm1.hip
bool wdg_lstm2_geom(const wdg_conv_plan* pl);   // conv_halo.hip: the two-feature layer on its pixel-per-thread step kernels?
struct WdgHaloLstm {   // ConvLSTM cell update / cell backward in the epilogue of the recurrent convolution (conv_halo.hip)
    int F, ldc, ldh;
    const float* c_prev;
    float* c_out;
    float* h_out;
    int bwd;
    const float* gates_t;
    const float* c_cur;
    const float* dc_in;
    float* dgates_out;
    float* dc_out;
};
void wdg_halo_set_lstm_fused(int v);
struct WdgHaloLn {     // LayerNormalization of the 16 output channels in the persistent 3x3 kernel's epilogue (conv_halo.hip)
    float* z;
    int ldz;
    long long img_stride_z;
    const float* gamma;
    const float* beta;
    float eps;
    float* mean_rstd;
};
bool wdg_halo_ln_eligible(const wdg_conv_plan* pl);
void wdg_halo_set_ln(int v);
int wdg_halo_launch(const wdg_conv_plan* pl, bool dgrad, const float* A, int ldA, long long imgStrideA, int upsample,
                    const float* Bw, const float* bias, float* Out, int act, float slope, int accumulate,
                    hipStream_t st, const WdgHaloLstm* cell = nullptr, const WdgHaloLn* ln = nullptr);

// conv_patch_h16.hip
void wdg_patch_h16_set(int v);
void wdg_patch_h16_set_budget(int kib);
void wdg_patch_h16_set_dbg(int v);
void wdg_patch_h16_set_nloop(int v);
void wdg_patch_h16_set_flat(int v);
int wdg_patch_h16_eligible(const wdg_conv_plan* pl);
int wdg_patch_h16_eligible_t(const wdg_conv_plan* pl);
// ConvLSTM gate columns: F features, columns interleaved (gate n & 3 of feature n >> 2).  c_out == NULL: plain convolution
// that writes its columns interleaved (the input part of the gates); else the recurrent step with the cell update in the epilogue.
struct WdgPatchGates {
    int F;
    const float* gates_x;
    const float* c_prev;
    float* c_out;
    int ldc;
    float* h_out;              // fp32 h [pixel][ldh], or NULL when only the 16-bit copy is wanted
    int ldh;
    int skip_k;
    void* h16_out;             // optional: h in the 16-bit operand format [pixel][ldh16]
    int ldh16;
};
int wdg_patch_h16_eligible_s(const wdg_conv_plan* pl);
int wdg_patch_h16_launch(const wdg_conv_plan* pl, int transposed1x1, const float* x, const void* w16, const float* bias,
                         const float* affine, float* y, int act, float slope, int accumulate, int fmt, hipStream_t st,
                         const WdgPatchGates* gx = nullptr, int out16 = 0, int in16 = 0);

// convlstm1.hip
void wdg_convlstm1_set_mfma(int v);

// wgrad_halo.hip
int wdg_wgrad_halo_eligible(const wdg_conv_plan* pl);
size_t wdg_wgrad_halo_ws_bytes(const wdg_conv_plan* pl);
int wdg_wgrad_thin_eligible(const wdg_conv_plan* pl);
int wdg_wgrad_thin_has_bias_row(const wdg_conv_plan* pl);
size_t wdg_wgrad_thin_ws_bytes(const wdg_conv_plan* pl);
void wdg_wgrad_thin_enable(int v);
int wdg_wgrad_thin_launch(const wdg_conv_plan* pl, const float* x, const float* dy, float* dw, float* dbias,
                          int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int wdg_wgrad_halo_launch(const wdg_conv_plan* pl, const float* x, const float* dy, float* dw, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st);
