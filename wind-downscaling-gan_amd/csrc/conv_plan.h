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
    // 2-D row tiles (t2_w > 0): the BM rows of a tile are a (BM / t2_w) x t2_w patch of output pixels instead of BM consecutive
    // pixels of one output row — fewer distinct input rows per tile, so neighbouring taps hit L1 / L2 instead of the fabric
    int t2_w, t2_wshift, t2_tiles_w;
    wdg_fastdiv div_t2_img, div_t2_w;   // tile index -> (image, tile row, tile column)
};
static inline void wdg_phase_finish(WdgPhase& ph) {
    ph.div_papb = wdg_fastdiv_make((unsigned)(ph.Pa * ph.Pb > 0 ? ph.Pa * ph.Pb : 1));
    ph.div_pb = wdg_fastdiv_make((unsigned)(ph.Pb > 0 ? ph.Pb : 1));
    ph.t2_w = ph.t2_wshift = ph.t2_tiles_w = 0;
    ph.div_t2_img = ph.div_t2_w = wdg_fastdiv_make(1u);
}
// enable 2-D row tiles of bm rows (bm, t2w powers of two) when the phase's pixel grid divides evenly
static inline bool wdg_phase_tile2d(WdgPhase& ph, int bm, int t2w) {
    const int t2h = bm / t2w;
    if (t2w <= 0 || t2h <= 0 || t2w * t2h != bm || ph.Pa % t2h || ph.Pb % t2w) return false;
    int sh = 0;
    while ((1 << sh) < t2w) ++sh;
    ph.t2_w = t2w; ph.t2_wshift = sh; ph.t2_tiles_w = ph.Pb / t2w;
    ph.div_t2_img = wdg_fastdiv_make((unsigned)((ph.Pa / t2h) * (ph.Pb / t2w)));
    ph.div_t2_w = wdg_fastdiv_make((unsigned)(ph.Pb / t2w));
    return true;
}

struct wdg_conv_plan {
    wdg_conv_geom g;
    int Cin_p, Cout_p, taps;
    int cus;
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


// conv_halo.hip
int wdg_halo_plan_init(wdg_conv_plan* pl);
void wdg_halo_plan_free(wdg_conv_plan* pl);
void wdg_halo_set_wg(int v);
void wdg_halo_set_persistent(int v);
void wdg_halo_set_th4(int v);
void wdg_halo_set_max_cin(int v);
void wdg_h16_set_small_tiles(int v);   // conv_igemm_bf16.hip
void wdg_h16_set_lstm_fused(int v);
void wdg_patch_h16_set_lstm_small(int v);
void wdg_lstm16_set_step(int v);       // convlstm16.hip
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
    float* h_out;
    int ldh;
    int skip_k;
};
int wdg_patch_h16_launch(const wdg_conv_plan* pl, int transposed1x1, const float* x, const void* w16, const float* bias,
                         const float* affine, float* y, int act, float slope, int accumulate, int fmt, hipStream_t st,
                         const WdgPatchGates* gx = nullptr, int out16 = 0);

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
