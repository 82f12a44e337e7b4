/*
 * ptpp.h -- C ABI of libptpp_hip.so: the MI355X (gfx950) kernels behind the
 * PromptTTS++ mel-synthesis hot path.
 *
 * The reference (line/promptttspp) is pure Python on PyTorch and has NO native
 * interface of its own; every entry point below replaces a
 * cluster of torch ops at the reference's L1->L0 seam.  Each declaration cites
 * the reference code (file:line under /root/reference) whose arithmetic it
 * implements.  INTEGRATION.md shows the ctypes binding a reference maintainer
 * would add.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (torch tensors:
 *     Tensor.data_ptr()).  The library allocates nothing persistent.
 *   - Activations are channels-last: a (B, T, C) tensor is B*T rows of C
 *     contiguous elements with an explicit row stride `ld*` (in elements).
 *     The Python boundary transposes the reference's (B, C, T) tensors.
 *   - `dtype`: PTPP_F32 (exact f32 MFMA path, parity mode) or PTPP_BF16
 *     (bf16 storage + bf16 MFMA, f32 accumulate).  Bias, LayerNorm/Snake
 *     parameters, masks/lengths and all statistics are always f32 / i32.
 *   - `lengths` (int32[B], may be NULL): per-utterance valid length; it
 *     replaces the reference's float/int mask tensors (sequence_mask,
 *     utils/model.py:30-34).  Row t of utterance b is "valid" iff
 *     t < lengths[b].
 *   - All launches are asynchronous on `stream` (a hipStream_t passed as
 *     void*); no call synchronises.  Re-entrant across streams.
 *   - Return value: 0 on success, negative PTPP_E* otherwise; never throws.
 *     ptpp_last_error() returns a thread-local message for the last failure.
 */
#ifndef PTPP_H_
#define PTPP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTPP_F32 0
#define PTPP_BF16 1

#define PTPP_OK 0
#define PTPP_EINVAL (-1)   /* bad argument (shape / alignment / dtype) */
#define PTPP_ELAUNCH (-2)  /* HIP launch error */
#define PTPP_ENOTSUP (-3)  /* configuration not built */

/* activation codes used by the fused epilogues */
#define PTPP_ACT_NONE 0
#define PTPP_ACT_RELU 1
#define PTPP_ACT_GELU 2 /* erf form, torch.nn.GELU() default (frame_prior.py:64) */
#define PTPP_ACT_SWISH 3
#define PTPP_ACT_TANH 4
#define PTPP_ACT_MISH 5

const char* ptpp_last_error(void);
int ptpp_version(void);

/* ------------------------------------------------------------------ *
 * Weight packing (run once per weight update; folds the cast to the
 * compute dtype that torch.autocast would otherwise do per use).
 * ------------------------------------------------------------------ */

/* Padded per-tap channel count of a packed weight for `dtype`. */
int ptpp_conv_cin_padded(int cin, int dtype);

/* torch Conv1d weight w[Cout][Cin][ks] (f32) -> wp[Cout][ks][CinP] (dtype),
 * K-contiguous, zero padded.  mode 0: forward operand.  mode 1: operand of
 * the data-gradient convolution, wp[Cin][ks][CoutP] with taps flipped. */
int ptpp_pack_conv_weight(const float* w, void* wp, int cout, int cin, int ks,
                          int mode, int dtype, void* stream);

/* ------------------------------------------------------------------ *
 * Conv1d / Linear as an MFMA implicit GEMM with fused epilogue.
 *   y[b,t,:] = res[b,t,:] + out_scale * mask_out(act(W * x_masked + bias))
 * Replaces nn.Conv1d / nn.Linear + the elementwise ops around them in
 *   esp/transformer/multi_layer_conv.py:52-67   (k=9, ReLU, masks)
 *   modules/variance_adaptor.py:31-36           (k=3/5, ReLU)
 *   modules/frame_prior.py:85-89                (k=17, GELU)
 *   modules/denoiser.py:58-64,69-83             (k=3 dilated, 1x1)
 *   esp/transformer/attention.py:32-35,229      (Linear 256x256)
 *   modules/mdn.py:50-78, modules/prompt_encoder.py:45-51 (Linear heads/MLP)
 *   vocoders/bigvgan.py:24-47,84-118            (dilated Conv1d, and
 *       ConvTranspose1d re-expressed as a 3-tap conv with Cout*stride
 *       outputs -- see promptttspp_amd/vocoders/bigvgan.py)
 * ------------------------------------------------------------------ */
typedef struct {
  const void* x;      /* (B*T, Cin) rows, stride ldx                */
  const void* wp;     /* packed weight, see ptpp_pack_conv_weight   */
  const float* bias;  /* (Cout) or NULL                             */
  const void* res;    /* (B*T, Cout) rows, stride ldr, or NULL      */
  void* y;            /* (B*T, Cout) rows, stride ldy               */
  const int32_t* lengths; /* (B) or NULL                            */
  int32_t B, T, Cin, Cout, ks, dil, pad;
  int32_t ldx, ldy, ldr;
  int32_t act;        /* PTPP_ACT_*                                 */
  int32_t in_mask;    /* treat x rows t >= lengths[b] as zero       */
  int32_t out_mask;   /* zero the conv term of rows t >= lengths[b] */
  float out_scale;
  int32_t dtype;
} ptpp_conv1d_args;

int ptpp_conv1d_fwd(const ptpp_conv1d_args* a, void* stream);

/* Same, with a second residual and a scale on the first:
 *   y = res_scale*res + res2 + out_scale*mask_out(act(...))
 * (lets the last conv of each AMP block accumulate the 3-block mean of
 *  vocoders/bigvgan.py:124-128 without an extra pass). */
int ptpp_conv1d_fwd_ex(const ptpp_conv1d_args* a, const void* res2, int ldr2,
                       float res_scale, void* stream);

/* Weight gradient of the convolution above (f32 accumulate, f32 output):
 *   dw[Cout][Cin][ks] (+)= sum_{b,t} dy[b,t,co] * x[b, t + j*dil - pad, ci]
 *   dbias[Cout]       (+)= sum_{b,t} dy[b,t,co]
 * `accumulate` != 0 adds to dw/dbias (gradient accumulation), else overwrites
 * via a zero-fill the caller performs (dw must be zeroed by the caller when
 * accumulate == 0; the kernel always uses atomic adds across row splits). */
int ptpp_conv1d_wgrad(const void* x, const void* dy, float* dw, float* dbias,
                      const int32_t* lengths, int B, int T, int Cin, int Cout,
                      int ks, int dil, int pad, int ldx, int lddy, int in_mask,
                      int dtype, void* stream);

/* ------------------------------------------------------------------ *
 * LayerNorm over the channel (last) dimension, biased variance.
 *   y = (x [+ res] - mean) * rsqrt(var + eps) * gamma + beta   [* mask]
 * Replaces the three LayerNorm variants of SURVEY.md F13:
 *   esp/transformer/layer_norm.py:12-33 (eps 1e-12),
 *   layers/norm.py:19-32 (eps 1e-5, (B,C,T) layout),
 *   modules/frame_prior.py:22-34 (eps 1e-5).
 * mean/rstd (f32, one per row) are written when non-NULL (for backward).
 * ------------------------------------------------------------------ */
int ptpp_layernorm_fwd(const void* x, const void* res, const float* gamma,
                       const float* beta, void* y, void* sum_out, float* mean,
                       float* rstd, const int32_t* lengths, int B, int T, int C,
                       float eps, int out_mask, int dtype, void* stream);

/* dx (and optionally the same gradient to `res`, which is dx) ; dgamma/dbeta
 * are accumulated with atomics into f32 buffers the caller zeroed. */
int ptpp_layernorm_bwd(const void* dy, const void* xsum, const float* gamma,
                       const float* mean, const float* rstd, void* dx,
                       float* dgamma, float* dbeta, const int32_t* lengths,
                       int B, int T, int C, int out_mask, int dtype,
                       void* stream);

/* ------------------------------------------------------------------ *
 * Anti-aliased Snake activation, one fused pass (layers/activations.py:22-44,
 * 74-138): replicate-pad -> x2 polyphase Kaiser-sinc up-FIR (12 taps, gain 2)
 * -> x + sin^2(x e^alpha)/(e^alpha + 1e-9) -> 12-tap low-pass, stride 2.
 *   x, y: (B, T, C) channels-last; log_alpha: (C) f32 (log domain, as
 *   stored in the state dict); filt_up/filt_down: HOST pointers to the 12 f32
 *   taps of `up.filter` / `down.lowpass.filter` (passed as kernel arguments;
 *   they are identical in the reference).
 * ------------------------------------------------------------------ */
int ptpp_aa_snake_fwd(const void* x, void* y, const float* log_alpha,
                      const float* filt_up, const float* filt_down, int B,
                      int T, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ *
 * Small fused elementwise / reduction kernels on channels-last rows.
 * ------------------------------------------------------------------ */

/* y = (a + b + c) * scale  (b, c nullable) -- AMP block mean
 * (vocoders/bigvgan.py:124-128). */
int ptpp_add3_scale(const void* a, const void* b, const void* c, void* y,
                    float scale, int64_t n, int dtype, void* stream);

/* Final BigVGAN stage (vocoders/bigvgan.py:129-131): conv_post with ONE
 * output channel (k taps over C channels) followed by tanh.
 *   x: (B, T, C) ; w: (ks, C) f32 ; y: (B, T) f32 */
int ptpp_conv_post_tanh(const void* x, const float* w, float bias, float* y,
                        int B, int T, int C, int ks, int dtype, void* stream);

/* Layout bridges between the reference's (B, C, T) f32 tensors and the
 * library's channels-last (B, T, C) `dtype` tensors. */
int ptpp_bct_to_btc(const float* x, void* y, int B, int C, int T, int dtype,
                    void* stream);
int ptpp_btc_to_bct(const void* x, float* y, int B, int T, int C, int dtype,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTPP_H_ */
