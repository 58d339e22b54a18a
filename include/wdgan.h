/*
 * wdgan.h — C-ABI of libwdgan.so: the MI355X (gfx950) native operator library beneath the
 * `downscaling` generator / discriminator / GAN train step.
 *
 * The reference (OpheliaMiralles/wind-downscaling-gan) has no FFI: its hot path sits behind the
 * Keras layer API and bottoms out in TensorFlow 2.4.3 kernels.  Every entry point below therefore
 * replaces a *Keras layer call site* of the reference; the citation after each declaration names
 * the call sites (file:line under /root/reference/src/downscaling) whose arithmetic it performs.
 *
 * Conventions
 *   - all tensors fp32, channels-last; an activation is addressed as
 *         base + img * img_stride + (h * W + w) * ld + c        (elements, not bytes)
 *     so channel-concatenations are zero-copy views (ld > C) of one wider buffer;
 *   - every activation view handed to a conv entry point has ld % 4 == 0, a 16-byte aligned base,
 *     and zero-filled pad channels up to the next multiple of 4;
 *   - raw device pointers + sizes, caller-owned buffers, no torch types, no hidden allocation on
 *     the launch path (plans allocate their small index tables at creation time);
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it;
 *   - return value: 0 on success, a negative wdg_status otherwise.  Nothing falls back to the CPU.
 */
#ifndef WDGAN_H
#define WDGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wdg_stream;

enum wdg_status {
    WDG_OK = 0,
    WDG_ERR_ARG = -1,      /* bad geometry / alignment / null pointer */
    WDG_ERR_WORKSPACE = -2,/* workspace too small: call wdg_conv_ws_bytes */
    WDG_ERR_HIP = -3       /* a HIP runtime call failed: see wdg_last_error */
};

/* Human-readable text of the last failure on the calling thread. */
const char* wdg_last_error(void);
/* Library version / target string, e.g. "wdgan 0.1 gfx950". */
const char* wdg_version(void);
/* Performance knobs for A/B measurements (results are identical for every setting):
 *   "igemm_pipe": 0 single LDS stage + two barriers per K-step (default), 1 double-buffered LDS + one barrier,
 *                 2 = 1 + fragment prefetch;
 *   "halo_weights_global": 1 (default) halo-tile kernel reads weight fragments from global memory, 0 stages
 *                 them in LDS;
 *   "xcd_swizzle": 1 (default) XCD-aware workgroup -> tile remap in the implicit-GEMM kernel, 0 off. */
int wdg_set_tuning(const char* key, int value);
int wdg_tuning_epoch(void);   /* number of wdg_set_tuning calls so far: key of launch sequences cached by the caller (HIP graphs) */
/* Number of compute units of the current device (used by the host-side split-K heuristic). */
int wdg_device_cus(void);

/* ------------------------------------------------------------------------------------------
 * Convolution family (implicit GEMM on v_mfma_f32_16x16x4_f32, exact fp32).
 *
 * One geometry describes the Keras Conv2D   y = conv(x, w; stride, pad)          [models.py:33,39,
 * 49,70,95,103,114,123,134; tf_utils.py:20,29] with x:(n_img,H,W,Cin), y:(n_img,Ho,Wo,Cout),
 * master weights HWIO [kh][kw][Cin][Cout] (the TF checkpoint layout).  The same three kernels
 * serve Conv2DTranspose [models.py:55,64] whose TF kernel (kh,kw,out,in) is the HWIO kernel of the
 * conv it is the adjoint of:  convT forward = wdg_conv_dgrad (+bias/activation epilogue),
 * convT input-gradient = wdg_conv_fwd, convT weight-gradient = wdg_conv_wgrad with x/dy swapped.
 * ConvLSTM2D input and recurrent convolutions [models.py:45,93,101] are plain 3x3 same convs.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_img;               /* B*T (TimeDistributed flattens the time axis) */
    int32_t H, W, Cin;           /* conv-input spatial size and logical channel count */
    int32_t ldx;                 /* pixel stride of the conv-input view (elements, %4==0) */
    int64_t img_stride_x;        /* image stride of the conv-input view (elements) */
    int32_t Ho, Wo, Cout;        /* conv-output spatial size and logical channel count */
    int32_t ldy;                 /* pixel stride of the conv-output view */
    int64_t img_stride_y;        /* image stride of the conv-output view */
    int32_t kh, kw, stride;      /* kernel size and stride (same in both directions) */
    int32_t pad_h, pad_w;        /* symmetric zero padding (ZeroPadding2D / 'same') */
} wdg_conv_geom;

typedef struct wdg_conv_plan wdg_conv_plan;

/* Builds the device-side index tables (tap/channel offsets, dgrad phases). Not on the launch path. */
int wdg_conv_plan_create(wdg_conv_plan** plan, const wdg_conv_geom* geom);
/* A plan for a RANGE of output channels [n0, n0 + geom->Cout) of a layer whose HWIO weight tensor has w_ld output channels:
 * wdg_conv_dgrad / wdg_conv_wgrad then take wD + n0 / dW + n0 of the FULL tensors and the y / dy views start at channel n0
 * (forward needs no special plan: the rows of wF are contiguous per output channel).  Use: keras ConvLSTM2D at
 * n_timesteps = 1 (gan/models.py:45, 93, 101) — c_0 = 0, the forget gate is never read and its gradient is zero, so its
 * quarter of the input convolution is dead in all three directions.  Implicit-GEMM layers only. */
int wdg_conv_plan_create_sliced(wdg_conv_plan** plan, const wdg_conv_geom* geom, int w_ld);
int wdg_conv_plan_destroy(wdg_conv_plan* plan);
/* Bytes of split-K scratch the plan may use (max over fwd/dgrad/wgrad). */
size_t wdg_conv_ws_bytes(const wdg_conv_plan* plan);
/* Launch configuration chosen for the plan (for profiling labels):
 * info[0..7] = {fwd BM, fwd BN, fwd split, dgrad BM, dgrad BN, dgrad split, wgrad BN, wgrad split};
 * BM == 0 / wgrad BN == 0 mark the halo-tile kernels (conv_halo.hip, wgrad_halo.hip). */
int wdg_conv_plan_info(const wdg_conv_plan* plan, int32_t* info);

/* y = act(conv(x, wF) + bias) [+ y if accumulate].
 *   wF: forward-packed weights [Cout][kh*kw][roundup4(Cin)] (see wdg_weight_pack).
 *   bias may be NULL; act: 0 linear, 1 LeakyReLU(slope).                     models.py:33,39,49,70,95,103,114 */
int wdg_conv_fwd(const wdg_conv_plan* plan, const float* x, const float* wF, const float* bias,
                 float* y, int act, float slope, int accumulate,
                 void* ws, size_t ws_bytes, wdg_stream stream);

/* dx = act(conv_transpose(dy, wD) + bias) [+ dx if accumulate]   (gradient w.r.t. the conv input;
 * also the *forward* of Conv2DTranspose, where bias/act are used).
 *   wD: data-gradient-packed weights [kh*kw][Cin][roundup4(Cout)] (== master when Cout%4==0).
 *                                                                            models.py:55,64; ganbase.py:35,60 */
int wdg_conv_dgrad(const wdg_conv_plan* plan, const float* dy, const float* wD, const float* bias,
                   float* dx, int act, float slope, int accumulate,
                   void* ws, size_t ws_bytes, wdg_stream stream);

/* The same two launches as PRODUCERS OF A BatchNormalization INPUT (models.py:33-34, 39-40, 49-50, 55-56: conv ->
 * bias -> LeakyReLU -> BatchNormalization), with the norm's first pass folded into the epilogue:
 *   stats  != NULL (training):  y = act(conv + bias), and the replica slabs stats[stats_rep][2][C] (fp64, zeroed by the
 *           caller, C = output channels rounded up to a multiple of 4) receive sum_p y and sum_p y^2 per channel — no separate read of y for
 *           wdg_bn_stats; finish with wdg_bn_finalize_train(stats, stats_rep, ...) and wdg_bn_apply;
 *   affine != NULL (inference): y = act(conv + bias) * affine[c] + affine[C + c], affine = the [scale | shift] of
 *           wdg_bn_finalize_infer — neither wdg_bn_stats nor wdg_bn_apply runs.
 * Exactly one of the two.  Launches that cannot carry the hook (split-K second stage, halo-tile kernel) run the
 * standalone pass behind the convolution, so the results do not depend on the route. */
int wdg_conv_fwd_bn(const wdg_conv_plan* plan, const float* x, const float* wF, const float* bias, float* y, int act,
                    float slope, double* stats, int stats_rep, const float* affine, void* ws, size_t ws_bytes,
                    wdg_stream stream);
int wdg_conv_dgrad_bn(const wdg_conv_plan* plan, const float* dy, const float* wD, const float* bias, float* dx, int act,
                      float slope, double* stats, int stats_rep, const float* affine, void* ws, size_t ws_bytes,
                      wdg_stream stream);

/* Data gradient of a convolution whose INPUT was produced by conv -> bias -> LeakyReLU -> LayerNormalization (the backward of
 * models.py:97,105,116,125 chained to the data gradient of the layer that reads the normalised tensor: models.py:113-114,
 * 122-123; ganbase.py:35,46,60):
 *   dx = dgrad(dy)                                               (as wdg_conv_dgrad, no bias / activation / accumulate)
 *   channels [c0, c0 + C) of dx are dz, the gradient w.r.t. the norm's OUTPUT; they are replaced IN PLACE by
 *   dpre = rstd * (dz g - mean_c(dz g) - xh * mean_c(dz g xh)) * lrelu'(y),   xh = (y - mean) * rstd,
 *   the gradient w.r.t. the producer's pre-activation (act_slope < 0: no activation derivative), and
 *   dgamma += sum_p dz xh, dbeta += sum_p dz, dbias += sum_p dpre      (each optional).
 * y: the norm's input [pixel][ldy_act] (C channels, own pixel / image stride), mean_rstd [pixel][2] as written by the forward.
 * par_ws: zero-initialised scratch of wdg_conv_dgrad_lnbwd_par_floats(C) floats (needed with parameter gradients; the call
 * leaves it zeroed).  One launch where an implicit-GEMM tile owns complete pixels — the norm's two reductions then run on the
 * accumulators and the standalone pass over dz (read dz + y, write dpre) disappears —, wdg_conv_dgrad + wdg_ln_bwd
 * otherwise: the results do not depend on the route.  The 7 x 7 stride-3 layer with 32 input and 64 output channels (the
 * discriminator's first block, models.py:105-116) runs on a kernel of its own, csrc/dgrad_patch_s3.hip: the dy patch of a
 * 24 x 24 block of dx pixels in LDS, all nine residue classes of the stride in one workgroup; its parameter sums are the same
 * bits on every run (no atomics) and it does not touch par_ws.  wdg_conv_dgrad_lnbwd_route tells which one a call would take
 * (profiling labels): 0 data gradient + wdg_ln_bwd, 1 implicit-GEMM epilogue, 2 the patch kernel. */
int wdg_conv_dgrad_lnbwd(const wdg_conv_plan* plan, const float* dy, const float* wD, float* dx, const float* y, int ldy_act,
                         int64_t img_stride_act, const float* mean_rstd, const float* gamma, int c0, int C, float act_slope,
                         float* dgamma, float* dbeta, float* dbias, float* par_ws, void* ws, size_t ws_bytes, wdg_stream stream);
int64_t wdg_conv_dgrad_lnbwd_par_floats(int C);
int wdg_conv_dgrad_lnbwd_route(const wdg_conv_plan* plan, int c0, int C, int ldy_act, int with_param_grads, size_t ws_bytes);

/* conv -> bias -> LeakyReLU -> LayerNormalization in one call (the discriminator's blocks, models.py:113-116, 122-125,
 * 134-136; the shortcut branch, tf_utils.py:29-31; the encoder, autoencoder.py:27-30):
 *   y = act(conv(x) + bias)            (kept: the backward pass needs the pre-norm activation)
 *   z = (y - mean) * rstd * gamma + beta   over the channel axis, eps inside the square root
 *   mean_rstd[2 * pixel] = (mean, rstd)    (optional)
 * z has the layout of y (same ld / image stride).  The normalisation runs in whichever epilogue owns complete output rows
 * (the implicit-GEMM tile when it spans all channels, the split-K second stage otherwise); every other route runs
 * wdg_ln_fwd behind the convolution.  Same results on every route. */
int wdg_conv_fwd_ln(const wdg_conv_plan* plan, const float* x, const float* wF, const float* bias, float* y, float* z,
                    const float* gamma, const float* beta, float eps, float* mean_rstd, int act, float slope,
                    void* ws, size_t ws_bytes, wdg_stream stream);
/* The same with z in a view of its own (pixel stride ldz, image stride img_stride_z): the discriminator writes the normalised
 * branch straight into its half of the [hr | mix] concatenation (models.py:102-108) while y keeps its own buffer.  The thin
 * full-resolution 16 -> 16 layer runs the norm in the epilogue of the halo-tile kernel; elsewhere a z view different from y's
 * takes the convolution followed by wdg_ln_fwd. */
int wdg_conv_fwd_ln_strided(const wdg_conv_plan* plan, const float* x, const float* wF, const float* bias, float* y, float* z,
                            int ldz, int64_t img_stride_z, const float* gamma, const float* beta, float eps, float* mean_rstd,
                            int act, float slope, void* ws, size_t ws_bytes, wdg_stream stream);

/* Fused UpSampling2D(2,'bilinear') + Conv2DTranspose forward: y = act(convT(upsample2x(x_low), wD) + bias).
 * `plan` is the transposed conv's plan on the UPSAMPLED grid (conv-output side 2H x 2W, stride 1, k <= 5,
 * Cin <= 64); x_low is the H x W tensor with pixel stride ld_low.  The upsampled tensor is never
 * written to memory.                                                                    models.py:62-64 */
int wdg_upconv_fwd(const wdg_conv_plan* plan, const float* x_low, int ld_low, int64_t img_stride_low,
                   const float* wD, const float* bias, float* y, int act, float slope, wdg_stream stream);

/* ---- inference precision (BASELINE configs[3]: bf16 tiled inference) --------------------------------
 * bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 activations in memory (rounded to bf16
 * while being staged).  w*16 are bf16 copies (wdg_convert_bf16) of the packed fp32 layouts wF / wD.
 * `affine` (optional, [2*C]: scale | shift) is applied after the activation: the inference-mode
 * BatchNormalization that follows every generator conv (models.py:34,40,50,56) fused into the epilogue.
 * Needs roundup4(Cin) % 8 == 0 (forward) / roundup4(Cout) % 8 == 0 (transposed).  The reference is fp32 only;
 * the tolerance of this path is defined by the build (tests/test_bf16_gpu.py). */
int wdg_convert_bf16(const float* src, void* dst_bf16, int64_t n, wdg_stream stream);
int wdg_conv_fwd_bf16(const wdg_conv_plan* plan, const float* x, const void* wF16, const float* bias,
                      const float* affine, float* y, int act, float slope, int accumulate, wdg_stream stream);
int wdg_conv_dgrad_bf16(const wdg_conv_plan* plan, const float* dy, const void* wD16, const float* bias,
                        const float* affine, float* dx, int act, float slope, int accumulate, wdg_stream stream);

/* bf16 halo-tile kernels for the generator's last two layers (stride 1, <= 64 output channels, channels % 8 == 0):
 * the thin forward conv, and the fused UpSampling2D(bilinear) + Conv2DTranspose forward (two-stage LDS staging). */
int wdg_conv_halo_fwd_bf16(const wdg_conv_plan* plan, const float* x, const void* wF16, const float* bias,
                           const float* affine, float* y, int act, float slope, wdg_stream stream);
int wdg_upconv_fwd_bf16(const wdg_conv_plan* plan, const float* x_low, int ld_low, int64_t img_stride_low,
                        const void* wD16, const float* bias, const float* affine, float* y, int act, float slope,
                        wdg_stream stream);

/* The same five entry points with IEEE fp16 operands (v_mfma_f32_16x16x32_f16; BASELINE configs[4], the stochastic
 * ensemble): identical kernels instantiated for the other 16-bit format, w*16 = fp16 copies (wdg_convert_f16). */
int wdg_convert_f16(const float* src, void* dst_f16, int64_t n, wdg_stream stream);
int wdg_conv_fwd_f16(const wdg_conv_plan* plan, const float* x, const void* wF16, const float* bias,
                     const float* affine, float* y, int act, float slope, int accumulate, wdg_stream stream);

/* ConvLSTM2D recurrent step in ONE launch, fp32 (gan/models.py:93,101 with n_timesteps > 1; the discriminator's 2- and
 * 16-feature ConvLSTMs): gates += conv3x3(h_prev, wF) (gates: the input part on entry, the pre-activation sums [i|f|c~|o] the
 * backward pass reads on exit), then c_out = hsig(f) c_prev + hsig(i) tanh(c~), h_out = hsig(o) tanh(c_out) on the
 * accumulators.  supported(): the plan's forward runs the halo-tile kernel with F = 16 or F = 2 (otherwise:
 * wdg_conv_fwd(accumulate) + wdg_lstm_fwd). */
int wdg_convlstm_step_supported(const wdg_conv_plan* plan, int F);
int wdg_convlstm_step(const wdg_conv_plan* plan, const float* h_prev, const float* wF, float* gates, const float* c_prev,
                      float* c_out, int ldc, float* h_out, int ldh, int F, wdg_stream stream);
/* The same step for layers the implicit GEMM runs (the generator's 128-feature ConvLSTM, models.py:45): gates += conv(h_prev),
 * cell update in the GEMM's epilogue, one launch per timestep instead of three.  wF_il: the packed forward weights
 * [4F][taps][Cin_p] with GATE-INTERLEAVED rows — row n holds gate n & 3 of feature n >> 2, i.e. row (n & 3) * F + (n >> 2) of the
 * standard layout — so that a lane's four accumulator registers are i, f, c~, o of one feature of one pixel.  gates keeps the
 * standard column order [i | f | c~ | o] (input part in, pre-activation sums out).  F % 16 == 0. */
int wdg_convlstm_step_gemm_supported(const wdg_conv_plan* plan, int F);
int wdg_convlstm_step_gemm(const wdg_conv_plan* plan, const float* h_prev, const float* wF_il, float* gates, const float* c_prev,
                           float* c_out, int ldc, float* h_out, int ldh, int F, wdg_stream stream);

/* Backward counterpart (BPTT of the same layers): dh_prev += conv_transpose(dgates_next, wD) completes the gradient of
 * h_{t-1}; the epilogue then differentiates the cell of timestep t-1 (gates_t, c_prev [NULL at t-1 = 0], c_cur, dc_in ->
 * dgates_out and, unless NULL, dc_out).  Replaces wdg_conv_dgrad(accumulate) + wdg_lstm_bwd of the next loop iteration. */
int wdg_convlstm_bwd_step_supported(const wdg_conv_plan* plan, int F);
int wdg_convlstm_bwd_step(const wdg_conv_plan* plan, const float* dgates_next, const float* wD, float* dh_prev,
                          const float* gates_t, const float* c_prev, const float* c_cur, const float* dc_in,
                          float* dgates_out, float* dc_out, int ldc, int F, wdg_stream stream);

/* The same two steps for the 16-feature layer (models.py:101) in kernels of their own (convlstm16.hip: 3 x 3, 16 -> 64, every
 * request ordered by first use; 33 -> 2x us forward, 44 -> 3x us backward per timestep at batch 8).  The recurrent kernel
 * [3][3][16][64] is handed over in the kernels' LDS layouts, written by wdg_convlstm16_pack (9216 floats each).  `plan` as for
 * wdg_convlstm_step; wdg_convlstm16_supported: 1 when the plan is that layer (else use wdg_convlstm_step / _bwd_step). */
int wdg_convlstm16_supported(const wdg_conv_plan* plan);
int wdg_convlstm16_pack(const float* w_hwio, float* wl_fwd, float* wl_bwd, wdg_stream stream);
/* BOTH recurrent layers of the discriminator (models.py:93 ConvLSTM2D(2) and :101 ConvLSTM2D(16): independent chains of T - 1
 * dependent steps) in ONE launch per timestep and direction: the workgroups of the two-feature layer's pixel-per-thread step ride
 * behind those of the 16-feature step.  plan16 / first argument group as wdg_convlstm16_step / _bwd_step; plan2 / second group as
 * wdg_convlstm_step / wdg_convlstm_bwd_step with F = 2 (wF2 / wD2: that layer's packed forward / data-gradient weights). */
int wdg_convlstm16_pair_supported(const wdg_conv_plan* plan16, const wdg_conv_plan* plan2);
int wdg_convlstm16_pair_step(const wdg_conv_plan* plan16, const float* h_prev, const float* wl_fwd, float* gates, const float* c_prev,
                             float* c_out, int ldc, float* h_out, int ldh, const wdg_conv_plan* plan2, const float* h_prev2,
                             const float* wF2, float* gates2, const float* c_prev2, float* c_out2, int ldc2, float* h_out2, int ldh2,
                             wdg_stream stream);
int wdg_convlstm16_pair_bwd_step(const wdg_conv_plan* plan16, const float* dgates_next, const float* wl_bwd, float* dh_prev,
                                 const float* gates_t, const float* c_prev, const float* c_cur, const float* dc_in, float* dgates_out,
                                 float* dc_out, int ldc, const wdg_conv_plan* plan2, const float* dgates_next2, const float* wD2,
                                 float* dh_prev2, const float* gates_t2, const float* c_prev2, const float* c_cur2, const float* dc_in2,
                                 float* dgates_out2, float* dc_out2, int ldc2, wdg_stream stream);
int wdg_convlstm16_step(const wdg_conv_plan* plan, const float* h_prev, const float* wl_fwd, float* gates, const float* c_prev,
                        float* c_out, int ldc, float* h_out, int ldh, wdg_stream stream);
int wdg_convlstm16_bwd_step(const wdg_conv_plan* plan, const float* dgates_next, const float* wl_bwd, float* dh_prev,
                            const float* gates_t, const float* c_prev, const float* c_cur, const float* dc_in,
                            float* dgates_out, float* dc_out, int ldc, wdg_stream stream);

/* 16-bit ConvLSTM2D inference (gan/models.py:45; TimeDistributed ConvLSTM2D(F, 3, padding='same', return_sequences=True)):
 * the input part of the gates for all timesteps, written with INTERLEAVED gate columns (column n = gate n & 3 of feature
 * n >> 2; Keras order i, f, c, o), then one launch per timestep: recurrent 3x3 convolution of h_{t-1} + cell update
 * (hard_sigmoid / tanh) in the epilogue -> c_t, h_t.  h_prev == c_prev == NULL is t = 0.  fmt: 0 bf16, 1 fp16.
 * `plan` is the 3x3 plan of the gate convolution (Cout = 4F, dense output); supported() tells whether it runs here
 * (otherwise: wdg_conv_fwd_bf16 + accumulate + wdg_lstm_fwd). */
int wdg_convlstm_h16_supported(const wdg_conv_plan* plan, int F);
int wdg_conv_fwd_h16_gates(const wdg_conv_plan* plan, const float* x, const void* wF16, const float* bias, float* gates_x,
                           int F, int fmt, wdg_stream stream);
int wdg_convlstm_step_h16(const wdg_conv_plan* plan, const float* h_prev, const void* wF16, const float* gates_x,
                          const float* c_prev, float* c_out, int ldc, float* h_out, int ldh, int F, int fmt,
                          wdg_stream stream);
/* The same two with activations in the 16-bit operand format (see wdg_conv_fwd_h16_act16): x16 / h_prev16 != 0: the layer input /
 * the previous hidden state holds 16-bit elements of `fmt` (plan strides in elements); h_out16: a copy of h_t in that format
 * [pixel][ldh16] — additional (h_out != NULL) or the only one (h_out == NULL).  The readers of h in inference (the next step,
 * the next convolution) round it to the operand format while staging: the same bits, half the bytes.  models.py:45 */
int wdg_conv_fwd_h16_gates_x16(const wdg_conv_plan* plan, const void* x, int x16, const void* wF16, const float* bias, float* gates,
                               int F, int fmt, wdg_stream stream);
int wdg_convlstm_step_h16x(const wdg_conv_plan* plan, const void* h_prev, int h_prev16, const void* wF16, const float* gates_x,
                           const float* c_prev, float* c_out, int ldc, float* h_out, int ldh, void* h_out16, int ldh16, int F, int fmt,
                           wdg_stream stream);
int wdg_conv_dgrad_f16(const wdg_conv_plan* plan, const float* dy, const void* wD16, const float* bias,
                       const float* affine, float* dx, int act, float slope, int accumulate, wdg_stream stream);
int wdg_conv_halo_fwd_f16(const wdg_conv_plan* plan, const float* x, const void* wF16, const float* bias,
                          const float* affine, float* y, int act, float slope, wdg_stream stream);
int wdg_upconv_fwd_f16(const wdg_conv_plan* plan, const float* x_low, int ld_low, int64_t img_stride_low,
                       const void* wD16, const float* bias, const float* affine, float* y, int act, float slope,
                       wdg_stream stream);

/* UpSampling2D(2,'bilinear') + Conv2DTranspose(5x5, stride 1, 'same') (models.py:62-64) evaluated as four 4x4
 * convolutions on the low-resolution grid with composite kernels (csrc/upconv4.hip): 16 instead of 25 taps per
 * output pixel, results equal to the two-step definition up to fp32 re-association (~1e-7 relative).
 * w_hwio: the layer's kernel as stored, [5][5][N][C] (N <= 16 layer outputs, C layer inputs);
 * wc: wdg_upconv4_weight_floats(N, C) floats, rebuilt by wdg_upconv4_pack whenever w changes.
 * x_low: [n_img][H][W][ld_low] (C channels used), y: [n_img][2H][2W][ldy] (N channels written). */
size_t wdg_upconv4_weight_floats(int N, int C);
int wdg_upconv4_supported(int N, int C, int H, int W);
int wdg_upconv4_pack(const float* w_hwio, int N, int C, float* wc, wdg_stream stream);
int wdg_upconv4_fwd(const float* x_low, int ld_low, int64_t img_stride_low, int n_img, int H, int W, int C,
                    const float* wc, const float* bias, float* y, int ldy, int64_t img_stride_y, int N,
                    int act, float slope, wdg_stream stream);

/* dw[kh][kw][Cin][Cout] (+)= sum_pixels x (*) dy  — HWIO, the master layout.   ganbase.py:46,60 */
int wdg_conv_wgrad(const wdg_conv_plan* plan, const float* x, const float* dy, float* dw,
                   int accumulate, void* ws, size_t ws_bytes, wdg_stream stream);
/* Same, plus the bias gradient dbias[Cout] += sum_pixels dy (always accumulated; NULL to skip): the
 * ConvLSTM2D kernel/bias gradients of models.py:93,101 in one pass over dgates. */
int wdg_conv_wgrad_bias(const wdg_conv_plan* plan, const float* x, const float* dy, float* dw, float* dbias,
                        int accumulate, void* ws, size_t ws_bytes, wdg_stream stream);

/* Repack master HWIO weights into the two kernel layouts.
 * wD may be NULL when Cout%4==0 (the master is then used directly as wD). */
int wdg_weight_pack(const float* w_hwio, float* wF, float* wD, int taps, int Cin, int Cout,
                    wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * tfa.layers.SpectralNormalization — one power iteration, in place.          models.py:33,39,49,55,
 * W = reshape(w, [rows, cols]) (cols = last kernel axis);                    95,103,114,123,134
 *   v = l2n(u W^T); u' = l2n(v W); sigma = v W u'^T; w <- w / sigma; u <- u'
 * scratch: >= wdg_sn_scratch_floats(rows, cols) floats.  All reductions run in a fixed order so
 * that replicated weights stay bit-identical across data-parallel ranks.
 * ------------------------------------------------------------------------------------------ */
size_t wdg_sn_scratch_floats(int rows, int cols);
int wdg_sn_power_iter(float* w, float* u, int rows, int cols, float* scratch, wdg_stream stream);

/* Host utility: CRC-32C (Castagnoli) continued from `crc` (0 to start) — the checksum of the TF tensor-bundle
 * checkpoint format that GAN.save_weights / load_weights of the reference go through (ganbase.py:132-140). */
uint32_t wdg_crc32c(const void* data, size_t n, uint32_t crc);

/* Batched weight preparation of one network: what every `training=True` call of a Keras model with
 * SpectralNormalization wrappers does before its first layer runs (all power iterations + in-place
 * w <- w / sigma, models.py:33-134 through tfa), plus the refresh of the kernel-layout copies — one
 * launch per stage for ALL layers instead of five launches per layer.  Per-layer arithmetic and
 * summation order equal wdg_sn_power_iter / wdg_weight_pack (bit-identical results). */
typedef struct wdg_prep_layer {
    float* w;            /* master HWIO weights                                               */
    float* u;            /* sn_u [cols] (NULL when sn == 0)                                   */
    float* wF;           /* forward layout copy, or NULL                                      */
    float* wD;           /* data-gradient layout copy, or NULL (Cout % 4 == 0: master is used) */
    int32_t rows, cols;  /* SN matrix view of w (cols = last kernel axis)                     */
    int32_t taps, cin, cout;
    int32_t sn;          /* 1: layer is wrapped in SpectralNormalization                      */
} wdg_prep_layer;
typedef struct wdg_prep_batch wdg_prep_batch;
#define WDG_PREP_SN 1        /* power iteration + w /= sigma on the SN layers, then repack them */
#define WDG_PREP_PACK_ALL 2  /* repack every layer (after an optimizer step / weight load)      */
int wdg_prep_batch_create(wdg_prep_batch** out, const wdg_prep_layer* layers, int n);
size_t wdg_prep_batch_scratch_floats(const wdg_prep_batch* b);
int wdg_prep_batch_run(const wdg_prep_batch* b, float* scratch, int flags, wdg_stream stream);
int wdg_prep_batch_destroy(wdg_prep_batch* b);

/* ------------------------------------------------------------------------------------------
 * BatchNormalization (axis -1, eps 1e-3, momentum 0.99) over P = B*T*H*W pixels.   models.py:34,40,50,56,69
 * Statistics are accumulated in fp64 so the SyncBN all-reduce can be inserted between
 * wdg_bn_stats and wdg_bn_finalize by the host.
 * ------------------------------------------------------------------------------------------ */
/* stats[0:C] += sum_p x, stats[C:2C] += sum_p x^2   (fp64; caller zeroes `stats` first) */
int wdg_bn_stats(const float* x, int64_t P, int C, int ldx, double* stats, wdg_stream stream);
/* training: mean/var from stats over `count` pixels -> scale/shift, saved mean/invstd, moving
 * stats update (moving = momentum*moving + (1-momentum)*batch).  scale_shift: [2*C]; saved: [2*C]. */
/* `stats` = `replicas` slabs [2][C] (summed here): wdg_bn_stats fills slab 0, the fused producers below spread
 * their atomics over all of them. */
int wdg_bn_finalize_train(const double* stats, int replicas, double count, const float* gamma, const float* beta,
                          float* moving_mean, float* moving_var, float momentum, float eps,
                          float* scale_shift, float* saved_mean_invstd, int C, wdg_stream stream);
/* stats[0] = sum over the replica slabs (in place): collapse before a data-parallel all-reduce of the statistics, then
 * finalize with replicas = 1. */
int wdg_bn_collapse(double* stats, int replicas, int C, wdg_stream stream);
/* inference: scale/shift from the moving statistics. */
int wdg_bn_finalize_infer(const float* gamma, const float* beta, const float* moving_mean,
                          const float* moving_var, float eps, float* scale_shift, int C,
                          wdg_stream stream);
/* z = x*scale + shift */
int wdg_bn_apply(const float* x, int ldx, const float* scale_shift, float* z, int ldz,
                 int64_t P, int C, wdg_stream stream);
/* backward, pass 1: red[0:C] += sum dz, red[C:2C] += sum dz*xhat  (fp64, caller zeroes) */
int wdg_bn_bwd_reduce(const float* dz, int lddz, const float* y, int ldy,
                      const float* saved_mean_invstd, int64_t P, int C, double* red,
                      wdg_stream stream);
/* backward, pass 2: dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) with the means taken
 * from red_mean/count (the SyncBN-all-reduced sums); dgamma += red_param[C:2C], dbeta +=
 * red_param[0:C] (this rank's own sums, so the later gradient all-reduce does not double count;
 * NULL skips them); if act_slope >= 0 the LeakyReLU derivative (sign of y) is fused:
 * dpre = dy * (y > 0 ? 1 : slope) and dbias += colsum(dpre).  dpre may alias dz. */
int wdg_bn_bwd_apply(const float* dz, int lddz, const float* y, int ldy,
                     const float* saved_mean_invstd, const float* gamma, const double* red_mean,
                     const double* red_param, double count, float act_slope, float* dpre,
                     int lddpre, float* dgamma, float* dbeta, float* dbias, int64_t P, int C,
                     wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * LayerNormalization (axis -1 only, eps 1e-3) per pixel.            models.py:97,105,116,125,136
 * ------------------------------------------------------------------------------------------ */
int wdg_ln_fwd(const float* y, int ldy, const float* gamma, const float* beta, float eps,
               float* z, int ldz, float* mean_rstd /* [P][2] or NULL */, int64_t P, int C,
               wdg_stream stream);
/* dpre = LN-backward(dz) * lrelu'(y) (act_slope < 0: no activation); dgamma/dbeta/dbias
 * accumulate (+=).  */
int wdg_ln_bwd(const float* dz, int lddz, const float* y, int ldy, const float* mean_rstd,
               const float* gamma, float act_slope, float* dpre, int lddpre,
               float* dgamma, float* dbeta, float* dbias, int64_t P, int C, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * ConvLSTM2D cell pointwise part, Keras gate order i,f,c,o, hard_sigmoid = clip(0.2x+0.5,0,1).
 * gates: [P][4F] pre-activations (x-conv + bias + h-conv).                   models.py:45,93,101
 * c_prev may be NULL (t = 0: c_0 = 0).
 * ------------------------------------------------------------------------------------------ */
int wdg_lstm_fwd(const float* gates, int ldg, const float* c_prev, int ldcp, float* c, int ldc,
                 float* h, int ldh, int64_t P, int F, wdg_stream stream);
/* Given dh (total gradient w.r.t. h_t) and dc_in (gradient flowing into c_t from t+1, may be
 * NULL), produce dgates [P][4F] and dc_prev (gradient w.r.t. c_{t-1}; may be NULL at t = 0). */
int wdg_lstm_bwd(const float* gates, int ldg, const float* c_prev, int ldcp, const float* c,
                 int ldc, const float* dh, int lddh, const float* dc_in, int lddci,
                 float* dgates, int lddg, float* dc_prev, int lddcp, int64_t P, int F,
                 wdg_stream stream);

/* Single-timestep ConvLSTM2D with few channels, fused (models.py:93,101 at n_timesteps = 1, where
 * h_0 = c_0 = 0 removes the recurrent conv and the forget path):  h = hs(o)*tanh(hs(i)*tanh(c~)) with
 * (i,f,c~,o) = conv3x3_same(x, wx) + bias.  wx is the master HWIO kernel [3][3][cin][4F].
 * Supported (cin, F): see wdg_convlstm1_supported (the discriminator's (2,2) and (5,16)).
 * The backward recomputes the gates from x instead of storing them; it writes the dense dgates tensor
 * [n_img*H*W][4F] (forget-gate slots zero) only when `dgates` is non-NULL (the weight gradient then is
 * wdg_conv_wgrad(x, dgates) and the bias gradient wdg_colsum(dgates)), and dx only when non-NULL. */
int wdg_convlstm1_supported(int cin, int F);
int wdg_convlstm1_fwd(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                      float* h, int ldh, int64_t img_stride_h, int n_img, int H, int W, int cin, int F,
                      wdg_stream stream);

/* The same three calls for the ConvLSTM2D that reads the concatenation [low_res | high_res] (models.py:100-101) WITHOUT the
 * concatenated tensor: the last x2_n (1 or 2) of the cin = 5 input channels are read from a second tensor x2 [n][H][W][ldx2]
 * (the caller's high-res tensor, in place), the others from x — the low-res part is constant over a train step, the high-res
 * part changes with every discriminator pass (twelve two-channel copies per step otherwise).  5 -> 16 layer only
 * (wdg_convlstm1_x2_supported); dw == NULL in the backward: no weight gradient. */
int wdg_convlstm1_x2_supported(int cin, int F, int x2_n);
int wdg_convlstm1_fwd_x2(const float* x, int ldx, int64_t img_stride_x, const float* x2, int ldx2, int64_t img_stride_x2,
                         int x2_n, const float* wx, const float* bias, float* h, int ldh, int64_t img_stride_h, int n_img,
                         int H, int W, int cin, int F, wdg_stream stream);
int wdg_convlstm1_bwd_x2(const float* x, int ldx, int64_t img_stride_x, const float* x2, int ldx2, int64_t img_stride_x2, int x2_n,
                         const float* wx, const float* bias, const float* dh, int lddh, int64_t img_stride_dh, float* dx,
                         int lddx, int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                         float* dw, float* dbias, void* ws, size_t ws_bytes, wdg_stream stream);

/* n_timesteps > 1 (models.py:101): the input part of the 5 -> 16-feature layer's gate pre-activations for ALL timesteps in one
 * launch, gates[n, y, x, i | f | c~ | o] = conv(x, kernel) + bias — the tensor the recurrent steps (wdg_convlstm16_step)
 * accumulate onto.  Same matrix-pipe kernel as wdg_convlstm1_fwd (45-row reduction, weights in registers), storing the four
 * accumulators instead of running the cell.  wx: the kernel [3][3][5][64] (HWIO); gates: dense [n][H][W][64], 16-byte aligned. */
int wdg_convlstm_gates_x_supported(int cin, int F);
int wdg_convlstm_gates_x(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias, float* gates,
                         int n_img, int H, int W, int cin, int F, wdg_stream stream);
/* Data gradient of wdg_convlstm_gates_x for the 2 -> 2-feature layer (models.py:93 under ganbase.py:35,60):
 * dx[..., 0:2] (+)= conv_transpose(dgates, wx); dgates dense [n][H][W][8], wx the kernel [3][3][2][8] (HWIO) as stored.
 * (wdg_convlstm_gates_x covers this layer too: one pixel per thread on the vector unit, both directions.) */
int wdg_convlstm_gates_dx_supported(int cin, int F);
int wdg_convlstm_gates_dx(const float* dgates, const float* wx, float* dx, int lddx, int64_t img_stride_dx, int accumulate,
                          int n_img, int H, int W, int cin, int F, wdg_stream stream);
int wdg_convlstm1_bwd(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                      const float* dh, int lddh, int64_t img_stride_dh, float* dgates, float* dx, int lddx,
                      int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                      wdg_stream stream);

/* Input gradient of the LAST cin - dx_c0 input channels only: dx[..., 0 : cin - dx_c0] (+)= d x[..., dx_c0 : cin].  The
 * discriminator's 5 -> 16 layer reads concat(low, high) (models.py:100-101); only the gradient of the two high-resolution
 * channels is ever used (gradient penalty ganbase.py:35, generator step :60) — with dx = the buffer that already holds the
 * 2 -> 2 layer's input gradient (models.py:93) and accumulate_dx = 1, d(high) is complete after this call.  dx_c0: 0, or 3
 * with cin = 5. */
int wdg_convlstm1_bwd_dx_from(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                              const float* dh, int lddh, int64_t img_stride_dh, float* dx, int lddx,
                              int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                              int dx_c0, wdg_stream stream);

/* wdg_convlstm1_bwd plus the layer's weight and bias gradient in the same pass: dw [3][3][cin][4F] += and
 * dbias [4F] += (MFMA over the LDS-resident x / dgates tiles, persistent blocks, block partials summed in block
 * order by a second kernel) — no dense dgates tensor is written.  ws: wdg_convlstm1_wgrad_ws_bytes() of scratch. */
size_t wdg_convlstm1_wgrad_ws_bytes(int n_img, int H, int W, int cin, int F);
int wdg_convlstm1_bwd_wgrad(const float* x, int ldx, int64_t img_stride_x, const float* wx, const float* bias,
                            const float* dh, int lddh, int64_t img_stride_dh, float* dx, int lddx,
                            int64_t img_stride_dx, int accumulate_dx, int n_img, int H, int W, int cin, int F,
                            float* dw, float* dbias, void* ws, size_t ws_bytes, wdg_stream stream);

/* Thin 3x3 'same' conv + bias + LeakyReLU + LayerNormalization fused (models.py:94-97,102-105).
 * forward : y = lrelu(conv(x, w) + bias) (kept for the backward), z = LN(y)*gamma + beta, mean_rstd [P][2].
 * backward: dpre = LN'(dz)*lrelu'(y) (dense [P][cout], must not alias dz), dx = conv^T(dpre) (optional),
 *           dgamma/dbeta/dbias += (all three or none).  w_hwio is the master kernel [3][3][cin][cout].
 * Supported (cin, cout): wdg_convln_supported ((2,16): the high-res-only branch). */
int wdg_convln_supported(int cin, int cout);
int wdg_convln_fwd(const float* x, int ldx, int64_t isx, const float* w_hwio, const float* bias,
                   const float* gamma, const float* beta, float eps, float slope, float* y, int ldy,
                   int64_t isy, float* z, int ldz, int64_t isz, float* mean_rstd, int n_img, int H, int W,
                   int cin, int cout, wdg_stream stream);
int wdg_convln_bwd(const float* dz, int lddz, int64_t isdz, const float* y, int ldy, int64_t isy,
                   const float* mean_rstd, const float* w_hwio, const float* gamma, float slope,
                   float* dpre, float* dx, int lddx, int64_t isdx, float* dgamma, float* dbeta,
                   float* dbias, int n_img, int H, int W, int cin, int cout, wdg_stream stream);

/* Backward of the same fused block from x instead of y / (mean, rstd): y and its LayerNorm statistics are recomputed
 * per pixel (the input has 2 channels, the saved tensors 18 floats per pixel), dpre stays in LDS, dx (optional),
 * dgamma / dbeta / dbias += (all or none) and — when dw is given — the kernel gradient dw [3][3][cin][cout] += come out
 * of the same launch.  Pair with wdg_convln_fwd(y = NULL, mean_rstd = NULL).  ws: wdg_convln_wgrad_ws_bytes(). */
size_t wdg_convln_wgrad_ws_bytes(int n_img, int H, int W, int cin);
int wdg_convln_bwd_x(const float* dz, int lddz, int64_t isdz, const float* x, int ldx, int64_t isx,
                     const float* w_hwio, const float* bias, const float* gamma, float eps, float slope,
                     float* dx, int lddx, int64_t isdx, float* dgamma, float* dbeta, float* dbias, float* dw,
                     void* ws, size_t ws_bytes, int n_img, int H, int W, int cin, int cout, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * UpSampling2D(2, 'bilinear'): half-pixel centres, edge clamp.                     models.py:62
 * ------------------------------------------------------------------------------------------ */
int wdg_upsample2x_fwd(const float* x, int ldx, int64_t img_stride_x, float* y, int ldy,
                       int64_t img_stride_y, int n_img, int H, int W, int C, wdg_stream stream);
int wdg_upsample2x_bwd(const float* dy, int lddy, int64_t img_stride_dy, float* dx, int lddx,
                       int64_t img_stride_dx, int n_img, int H, int W, int C, int accumulate,
                       wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Backward of "UpSampling2D(2,'bilinear') -> Conv2DTranspose(5x5)" (models.py:60-64) on the low-resolution
 * grid (csrc/upconv_col.hip): col [n, Hl, Wl, 25*C] = the output gradient pulled through the tap shifts and the
 * adjoint of the clamped bilinear stencil (C = the layer's output channels: 4, 8 or 16; dy is [n, 2Hl, 2Wl, >=C]);
 * the two GEMMs dx = col * W and dW += col^T * x then run through wdg_conv_fwd / wdg_conv_wgrad as 1x1
 * convolutions with the layer's weight tensor viewed as [25*C][C_in].  A quarter of the multiply-adds of the
 * full-resolution formulation (tf.GradientTape through the two Keras layers, ganbase.py:55-61).
 * ------------------------------------------------------------------------------------------ */
int wdg_upconv_col_supported(int C);
int wdg_upconv_col(const float* dy, int ldy, int64_t img_stride_dy, float* col, int n_img, int Hl, int Wl, int C,
                   wdg_stream stream);
/* The forward in the same form: z [n, Hl, Wl, 25*C] = x * W as a 1x1 GEMM
 * (wdg_conv_dgrad / wdg_conv_dgrad_bf16 with the weights viewed as [25*C][C_in]), then
 * y = affine(act(bias + gather(z))): the adjoint of wdg_upconv_col with the clamped-bilinear coefficients,
 * y [n, 2Hl, 2Wl, >= C]; affine = optional [2*C] scale | shift (fused inference BatchNorm). */
int wdg_upconv_gather(const float* z, const float* bias, const float* affine, float* y, int ldy,
                      int64_t img_stride_y, int n_img, int Hl, int Wl, int C, int act, float slope,
                      double* stats, int stats_rep, wdg_stream stream);
/* The same pair with z in the 16-bit operand format (inference precision only): the column GEMM rounds its fp32 sums to bf16 /
 * fp16 on store (z16 [n, Hl, Wl, 25 * C] 16-bit elements), the gather widens them on load — z is the largest tensor of the 16-bit
 * forward and has exactly one writer and one reader.  `plan`: 1 x 1 plan with x side = z, y side = x_low (as for
 * wdg_conv_dgrad_bf16).  fmt: 0 bf16, 1 fp16. */
int wdg_upconv_colgemm_h16_supported(const wdg_conv_plan* plan);   /* 1 when the patch kernel takes this map; else use the fp32 z route */
int wdg_upconv_colgemm_h16(const wdg_conv_plan* plan, const float* x_low, const void* wD16, void* z16, int fmt, wdg_stream stream);
int wdg_upconv_gather_h16(const void* z16, int fmt, const float* bias, const float* affine, float* y, int ldy, int64_t img_stride_y,
                          int n_img, int Hl, int Wl, int C, int act, float slope, wdg_stream stream);
/* Both stages in ONE launch (inference precision; models.py:62-64 behind api.py:130-138): a workgroup stages the 12 x 12 window of
 * x_low of its 8 x 8 low-resolution tile once (rounded to the operand format), forms the window's z slice per tap row on the
 * 16-bit MFMA in LDS and runs the gather passes on it — the column tensor (400 floats per low-resolution pixel) never reaches
 * memory.  x_low [n, Hl, Wl, ldx >= Cin] fp32; w16: the layer's weight tensor [25 * C][Cin] in the operand format (fmt 0 bf16,
 * 1 fp16); y [n, 2 Hl, 2 Wl, ldy >= C] fp32 = affine(act(bias + ...)).  Supported: Cin 160, C 16 (the shipped generator). */
int wdg_upconv_fused_h16_supported(int Cin, int C);
int wdg_upconv_fused_h16(const void* x_low, int ldx, int64_t img_stride_x, const void* w16, int fmt, const float* bias,
                         const float* affine, void* y, int ldy, int64_t img_stride_y, int n_img, int Hl, int Wl, int Cin, int C,
                         int act, float slope, int out16, int in16, wdg_stream stream);
/* out16 != 0: y holds 16-bit elements of the operand format (ldy / img_stride_y in elements, ldy % 8 == 0) — for the reader below,
 * which rounds to that format anyway: the same values, half the bytes.  in16 != 0: x_low likewise (ldx % 8 == 0).
 * wdg_conv_thin16_fwd_h16: the generator's output conv (models.py:70: 3 x 3, stride 1, 16 (padded) input channels, <= 4 output
 * channels at a pixel stride of 4 floats) reading x16 [n, H, W, ldx16] in the operand format — bit for bit the result of
 * wdg_conv_halo_fwd_bf16 / _f16 on the fp32 tensor x16 was rounded from.  `plan`: the layer's forward plan. */
int wdg_conv_thin16_fwd_h16(const wdg_conv_plan* plan, const void* x16, int ldx16, int64_t img_stride_x16, const void* wF16, int fmt,
                            const float* bias, const float* affine, float* y, int act, float slope, wdg_stream stream);

/* Activations in the 16-bit operand format between two layers of the inference-precision forward.  Every such layer rounds its
 * input to the operand format while staging it; when all readers of a tensor are such layers the producer can store it rounded —
 * the values multiplied are the same bits, the tensor has half the bytes.  in16 / out16: x / y hold 16-bit elements (bf16 for
 * fmt 0, IEEE fp16 for fmt 1) and the plan's ldx / ldy / image strides count elements of the tensors as stored.  Only the
 * input-patch kernel takes this route; wdg_conv_h16_act16_supported says whether it would (transposed 0: the plan's forward conv
 * = wdg_conv_fwd_bf16 / _f16; 1: its transposed direction = wdg_conv_dgrad_bf16 / _f16: 1 x 1, or k x k stride-k). */
int wdg_conv_h16_act16_supported(const wdg_conv_plan* plan, int transposed, int in16, int out16);
int wdg_conv_fwd_h16_act16(const wdg_conv_plan* plan, const void* x, int in16, const void* wF16, int fmt, const float* bias,
                           const float* affine, void* y, int out16, int act, float slope, wdg_stream stream);
int wdg_conv_dgrad_h16_act16(const wdg_conv_plan* plan, const void* dy, int in16, const void* wD16, int fmt, const float* bias,
                             const float* affine, void* dx, int out16, int act, float slope, wdg_stream stream);


/* ------------------------------------------------------------------------------------------
 * Flatten + Dense(1) per timestep + GlobalAveragePooling1D over T.             models.py:137-140
 * x: [T*B][K] rows, time-major (row = t*B + b, ld = K), w: [K], b: [1], score: [B].
 * ------------------------------------------------------------------------------------------ */
int wdg_dense_gap_fwd(const float* x, const float* w, const float* b, float* score, int B, int T,
                      int K, wdg_stream stream);
/* dscore: [B]; dx[(t,b)][k] = dscore[b]/T * w[k]; dw += sum x*dscore/T; db += sum dscore.
 * dw/db may be NULL (input-gradient-only pass, ganbase.py:35,60). */
int wdg_dense_gap_bwd(const float* x, const float* w, const float* dscore, float* dx, float* dw,
                      float* db, int B, int T, int K, wdg_stream stream);
/* The same head backward chained with the backward of the LayerNormalization (+ LeakyReLU) whose OUTPUT is the head's input x
 * (models.py:125/136 under :137-140): dx receives dpre, the gradient w.r.t. that norm's producer's pre-activation, instead of
 * dz = dscore / T * w (which never leaves the registers); y [rows * K / C][C] is the norm's input, mean_rstd as written by the
 * forward, dgamma / dbeta / dbias accumulate (optional; par_ws: zeroed scratch of wdg_conv_dgrad_lnbwd_par_floats(C) floats).
 * Equals wdg_dense_gap_bwd followed by wdg_ln_bwd in place. */
int wdg_dense_gap_bwd_ln(const float* x, const float* w, const float* dscore, float* dx, float* dw, float* db, int B, int T,
                         int K, const float* y, const float* mean_rstd, const float* gamma, int C, float act_slope,
                         float* dgamma, float* dbeta, float* dbias, float* par_ws, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / reductions used by GAN.train_step.
 * ------------------------------------------------------------------------------------------ */
/* dst[img][q][c] (+)= src[img][q][c], c < C, q < pixels_per_img  (channel pack/unpack and
 * batch<->time-major permutation between strided views; any alignment) */
int wdg_copy_channels(const float* src, int lds, int64_t img_stride_src, float* dst, int ldd,
                      int64_t img_stride_dst, int n_img, int64_t pixels_per_img, int C,
                      int accumulate, wdg_stream stream);
/* The same with a two-level image index (image = outer * n_inner + inner, separate strides per level on both sides):
 * one launch for the (B,T) <-> (T,B) permutation between the API layout and the time-major activations. */
int wdg_copy_channels_2level(const float* src, int lds, int64_t inner_stride_src, int64_t outer_stride_src,
                             float* dst, int ldd, int64_t inner_stride_dst, int64_t outer_stride_dst,
                             int n_outer, int n_inner, int64_t pixels_per_img, int C, int accumulate,
                             wdg_stream stream);
/* Window patches of a strided grid — shortcut_convolution (tf_utils.py:15-32; stride >= kernel, so the windows are
 * disjoint): out[n][oy][ox][(ky*k + kx)*C + c] = x[n][oy*stride - pad + ky][ox*stride - pad + kx][c] (0 outside), t x t
 * windows per image; the conv becomes a 1x1 convolution on `out` with the weights viewed as [k*k*C][Cout].
 * wdg_patch_scatter is the adjoint: dx (+)= the window gradient at each pixel's (unique) window position, 0 elsewhere. */
int wdg_patch_gather(const float* x, int ldx, int64_t img_stride_x, float* out, int n_img, int H, int W, int C,
                     int k, int stride, int pad, int t, wdg_stream stream);
int wdg_patch_scatter(const float* dpatch, float* dx, int lddx, int64_t img_stride_dx, int n_img, int H, int W, int C,
                      int k, int stride, int pad, int t, int accumulate, wdg_stream stream);
/* out[c] (+)= sum_p x[p][c]                                      (bias gradients) */
int wdg_colsum(const float* x, int ldx, int64_t P, int C, float* out, int accumulate,
               wdg_stream stream);
/* Internal activations are TIME-MAJOR: image index = t*B + b (ConvLSTM steps then touch
 * contiguous slabs).  out = eps[b]*a + (1-eps[b])*b_, b = (p / pixels_per_img) % B   ganbase.py:30-31 */
int wdg_lerp_batch(const float* a, int lda, const float* b_, int ldb, const float* eps,
                   float* out, int ldo, int64_t P, int64_t pixels_per_img, int B, int C,
                   wdg_stream stream);
/* out[b][c] = sum over t and the pixels of batch element b of x^2          ganbase.py:36 */
int wdg_sumsq_batch_ch(const float* x, int ldx, int64_t pixels_per_img, int T, int B, int C,
                       float* out, wdg_stream stream);
/* out[i] = mean(x[off[2i]:off[2i+1]]^2)   (g_gradient_param / d_gradient_param, ganbase.py:80-81) */
int wdg_segment_meansq(const float* x, const int64_t* off, int nseg, float* out,
                       wdg_stream stream);

/* Tile bookkeeping of the inference driver (api.py:96-151, `predict`) on the device.
 * wdg_tiles_gather_normalise: keys4[n] = {sx, row0, k, 0}; tile n = field[k*T .. k*T+T-1][row0, row0-1, ...][sx .. sx+S-1][:]
 *   (latitude flipped, api.py:119) -> tiles [N][T][S][S][C]; NaN-aware mean / population std per (column inside the tile,
 *   channel) over all tiles, timesteps and rows (np.nanmean / np.nanstd over axes (0,1,2), api.py:126-129; fp64 sums, fp32
 *   results in mean_std [S*C][2]); tiles are normalised in place.  stats_scratch: replicas * S*C * 3 doubles.
 * wdg_tiles_blend: the first n_real predictions of a group [B][T][S][S][ldp] (2 channels used), cropped by `crop` pixels per
 *   side, are added into acc [NT][LAT][LON][2] (fp64) and counted in cnt [NT][LAT][LON] (api.py:139-150). */
int wdg_tiles_gather_normalise(const float* field, int LAT, int LON, int C, const int32_t* keys4, int N, int T, int S, float* tiles,
                               double* stats_scratch, int replicas, float* mean_std, wdg_stream stream);
int wdg_tiles_blend(const float* pred, int ldp, const int32_t* keys4, int n_real, int T, int S, int crop, int LAT, int LON, double* acc,
                    int32_t* cnt, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Generator evaluation metrics (gan/metrics.py; compiled into the GAN by api.py:77-81 and updated by
 * every train step, ganbase.py:71).  real / fake: dense [B][T][H][W][2] wind fields.
 *
 * wdg_metrics_pointwise: one pass, out6[b][0..5] (fp64) = per-sample sums of
 *   0 tau*((u^-beta u)^2+(v^-beta v)^2)   wind_speed_weighted_rmse  metrics.py:32-45  (sqrt(sum/THW))
 *   1 (|w|-|w^|)^2                        wind_speed_rmse           metrics.py:79-88
 *   2 acos(clip(cos,-1,1))/pi             angular_cosine_distance   metrics.py:94-101
 *   3 0.5*(1-cos)                         opposite_cosine_similarity metrics.py:103-105
 *   4 u^2+v^2, 5 u^2(u-u^)^2+v^2(v-v^)^2  extreme_weighted_rmse     metrics.py:66-73   (sqrt(sum5_b / sum_b sum4_b))
 * NaN terms count as zero where the reference masks them (tf.where(is_nan)). */
int wdg_metrics_pointwise(const float* real, const float* fake, int64_t pixels_per_sample, int B, double* out6,
                          wdg_stream stream);
/* log_spectral_distance (metrics.py:121-137): out[b] = sum over the bins of the two rfft2d spectra (interleaved
 * complex64, [B][bins_per_sample]) of (10*log10((|R|^2+eps)/(|F|^2+eps)))^2; lsd = sqrt(out/bins). */
int wdg_lsd_reduce(const float* spec_real, const float* spec_fake, int64_t bins_per_sample, int B, float eps,
                   double* out, wdg_stream stream);
/* spatially_convolved_ks_stat (metrics.py:155-187): out[(H-patch+1)*(W-patch+1)] (fp64) = mean over (time x channel)
 * and batch of max_k |ECDF_real(p_k) - ECDF_fake(p_k)| on every stride-1 patch, p = points100 (the 100 evaluation
 * points of ks_stat_on_patch, ascending).  scratch: wdg_spatial_ks_scratch_bytes bytes.  patch <= 48. */
size_t wdg_spatial_ks_scratch_bytes(int B, int T, int H, int W, int C);
int wdg_spatial_ks(const float* real, const float* fake, int B, int T, int H, int W, int C, int patch,
                   const float* points100, void* scratch, double* out, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * FlexibleNoiseGenerator: N(0, std^2) from Philox4x32-10 + Box-Muller.   data_generator.py:319-335
 * Element e of the (dense, row-major) logical tensor [P][C] uses counter (offset + e/4), lane e%4.
 * out[p*ldo + c] = (add ? add[p*lda + c] : 0) + std * z       (instance noise: ganbase.py:40,42)
 * ------------------------------------------------------------------------------------------ */
int wdg_philox_normal(float* out, int ldo, const float* add, int lda, int64_t P, int C,
                      uint64_t seed, uint64_t offset, float std, wdg_stream stream);

/* Generator input in one pass (models.py:28: concatenate([low-res image, noise])): out [rows, ld] with rows = T' * B * XY in
 * time-major order (row = (t * B + b) * XY + r) receives [image[b, t, r, 0:CI] | std * N(0, 1) x CN | zeros]; image element
 * (b, t, r, c) sits at image[b * img_stride_b + t * img_stride_t + r * CI + c].  The noise is exactly the stream
 * wdg_philox_normal(out + CI, ld, NULL, 0, rows, CN, seed, offset, std) writes.  Supported: CI 3, CN 20, ld 24. */
int wdg_input_assemble_supported(int CI, int CN, int ld);
int wdg_input_assemble(const float* image, int64_t img_stride_b, int64_t img_stride_t, int CI, float* out, int ld, int64_t rows,
                       int B, int XY, int CN, uint64_t seed, uint64_t offset, float std, wdg_stream stream);
/* The same for B of the Bo batch slots of a larger time-major buffer — destination row (t * Bo + b0 + b) * XY + r; image source,
 * Philox counters and values exactly those of wdg_input_assemble on the dense [T' * B * XY, ld] view.  One predict() group of 16
 * tiles with its own noise draw (api.py:132-137) inside a forward pass that carries several groups. */
int wdg_input_assemble_slots(const float* image, int64_t img_stride_b, int64_t img_stride_t, int CI, float* out, int ld, int64_t rows,
                             int B, int XY, int CN, uint64_t seed, uint64_t offset, float std, int Bo, int b0, wdg_stream stream);
/* ... with the rows in the 16-bit operand format of the inference-precision layers (fmt 0 bf16, 1 fp16; out holds 16-bit
 * elements, ld in elements): the values wdg_input_assemble_slots writes, rounded to nearest even — what the generator's first
 * 16-bit layer would round them to while staging (wdg_conv_fwd_h16_act16 with in16). */
int wdg_input_assemble_h16(const float* image, int64_t img_stride_b, int64_t img_stride_t, int CI, void* out, int ld, int64_t rows,
                           int B, int XY, int CN, uint64_t seed, uint64_t offset, float std, int fmt, int Bo, int b0, wdg_stream stream);
/* U[0,1) for the interpolation coefficients eps (ganbase.py:30). */
int wdg_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * tf.keras.optimizers.Adam update, TF form (epsilon outside the bias correction):   train.py:34-35,57-58
 *   m += (1-b1)(g-m); v += (1-b2)(g^2-v); p -= lr_t * m / (sqrt(v) + eps),
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the host.  g is pre-scaled by grad_scale
 *   (1/world_size after an RCCL sum all-reduce).
 * ------------------------------------------------------------------------------------------ */
int wdg_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                float beta2, float eps, float grad_scale, wdg_stream stream);
/* Zero 1..16 element ranges [begin, end) of a flat fp32 buffer in one launch (begin_end: host array of n pairs, every value a
 * multiple of 4): what `optimizer.zero_grad` leaves of the gradient buffer when the big kernel gradients are stored by their first
 * weight-gradient launch instead of being zero-filled (ganbase.py:39,50: a fresh tape per update). */
int wdg_zero_ranges(float* base, const int64_t* begin_end, int n, wdg_stream stream);

/* ------------------------------------------------------------------------------------------
 * Measurement only (no reference counterpart): a stand-in for a collective's kernel on a one-GPU box.  `blocks` workgroups copy
 * `bytes` from src (wrapping around src_bytes) to dst and then idle until min_us microseconds have passed since the kernel
 * started — what a link-bound RCCL ring all-reduce on its own stream looks like to the rest of the chip.  bytes = 0: only the
 * wait (a latency-bound small collective).  engine/trainer.py DistSync with WDG_DP_PROXY=1; bench.py dp_fifth_queue_proxy.
 * ------------------------------------------------------------------------------------------ */
int wdg_dp_proxy(const void* src, void* dst, int64_t src_bytes, int64_t bytes, int blocks, float min_us, wdg_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* WDGAN_H */
