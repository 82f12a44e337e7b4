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
 *     Tensor.data_ptr()) unless a parameter says HOST.  The library allocates
 *     nothing persistent.
 *   - Activations are channels-last: a (B, T, C) tensor is B*T rows of C
 *     contiguous elements, with an explicit row stride `ld*` (in elements)
 *     where noted.  The Python boundary transposes the reference's (B, C, T).
 *   - `dtype`: PTPP_F32 (exact f32 MFMA path, parity mode) or PTPP_BF16
 *     (bf16 storage + bf16 MFMA, f32 accumulate).  Bias, LayerNorm/Snake
 *     parameters, lengths, statistics and gradients of parameters are f32/i32.
 *   - `lengths` (int32[B], may be NULL): per-utterance valid length; it
 *     replaces the reference's mask tensors (sequence_mask,
 *     utils/model.py:30-34).  Row t of utterance b is valid iff t < lengths[b].
 *   - Dropout is counter based: element keep-mask = f(seed, element index), so
 *     the backward pass regenerates it from the same (p, seed); p == 0 disables.
 *   - All launches are asynchronous on `stream` (a hipStream_t passed as
 *     void*); no call synchronises.  Re-entrant across streams.
 *   - Return value: 0 on success, negative PTPP_E* otherwise; never throws.
 *     ptpp_last_error() returns a thread-local message for the last failure.
 */
#ifndef PTPP_H_
#define PTPP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTPP_F32 0
#define PTPP_BF16 1
#define PTPP_F16 2 /* IEEE half, f32 accumulation: the vocoder's inference kernels (BigVGAN, BASELINE configs 4 / 5; the reference's AMP
                      is fp16, trainers/tts.py:92,203-211) -- ptpp_pack_conv_weight (modes 0 / 1), ptpp_conv1d_fwd*, the layout bridges,
                      ptpp_aa_snake_fwd, ptpp_amp_layer_fwd, ptpp_snake_conv1d_fwd, ptpp_snake_conv_post_tanh, ptpp_conv_post_tanh,
                      ptpp_add3_scale, ptpp_cast_from_f32; every other entry point refuses it */

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
#define PTPP_ACT_GATE 6 /* conv epilogue only, bf16: DiffNet gate sigmoid(a)*tanh(b) fused (see ptpp_conv1d_fwd) */

/* attention variants */
#define PTPP_ATTN_RELPOS_NEW 0    /* esp/transformer/attention.py:207-305 */
#define PTPP_ATTN_RELPOS_LEGACY 1 /* esp/transformer/attention.py:111-206 */
#define PTPP_ATTN_PLAIN 2         /* no positional term (BERT self-attention) */

const char* ptpp_last_error(void);
int ptpp_version(void);
/* Stream fork/join: everything enqueued on `waiter` after this call runs after everything enqueued on
 * `signaler` before it (hipEventRecord + hipStreamWaitEvent on a pooled event; both raw hipStream_t). */
int ptpp_stream_wait(void* waiter, void* signaler);

/* ------------------------------------------------------------------ *
 * Weight packing (once per weight version; folds the cast to the compute
 * dtype that torch.autocast would otherwise do per use).
 * ------------------------------------------------------------------ */

/* Padded per-tap channel count of a packed weight for `dtype`. */
int ptpp_conv_cin_padded(int cin, int dtype);

/* torch Conv1d weight w[Cout][Cin][ks] (f32) -> wp[Cout][ks][CinP] (dtype),
 * K-contiguous, zero padded.  mode 0: forward operand.  mode 1: operand of
 * the data-gradient convolution, wp[Cin][ks][CoutP] with taps flipped.  mode 2: mode 0 for a (gate | filter) weight
 * (Cout = 2C, Cout % 8 == 0) with its rows in the interleaved order of the fused DiffNet gate epilogue (PTPP_ACT_GATE):
 * packed row 8g + e = gate row 4g + e (e < 4) / filter row C + 4g + e - 4 (e >= 4).  modes 3 / 4 (bf16, 256 operand rows,
 * inner % 64 == 0): the elements of mode 0 / 1 re-ordered into the OPERAND STREAM of ptpp_conv1d_rt_fwd -- 2 * ks * inner / 64
 * stages of 16 KiB in consumption order, each the LDS image of its MFMA fragments (csrc/pack.hip stream_index). */
int ptpp_pack_conv_weight(const float* w, void* wp, int cout, int cin, int ks,
                          int mode, int dtype, void* stream);

/* nn.Conv2d weight w[Cout][Cin][3][3] (f32) -> both operands of the im2col GEMM of ptpp_im2col3x3s2 rows
 * (K index = (kh*3 + kw)*cinq + ci, channels zero-padded from Cin to cinq):  wp_fwd = mode 0 and wp_bwd (nullable) = mode 1
 * of ptpp_pack_conv_weight applied to that (Cout, 9*cinq) matrix, in one launch (modules/reference_encoder.py:61-82). */
int ptpp_pack_conv2d_3x3(const float* w, void* wp_fwd, void* wp_bwd, int cout, int cin,
                         int cinq, int dtype, void* stream);

/* Re-pack MANY operands in one launch (after an optimiser step).  table: device array of n_entries rows
 * of 10 int64: {src f32 (cout,cin,ks), dst base, cout, cin, ks, mode, dtype, innerp of dst,
 * row (mode 0) / column (mode 1) offset inside dst, first block}; block b serves row block_map[b]
 * (block_map NULL: the last row whose "first block" <= b, by binary search), one 32 x 32
 * (cout x cin) tile with all taps per block; total_blocks = sum of ceil(cout/32) * ceil(cin/32).
 * Several sources may share one dst (fused projections).  dst padding is NOT written (zero it once). */
int ptpp_pack_conv_weights_batched(const int64_t* table, int n_entries,
                                   const int32_t* block_map, int total_blocks,
                                   void* stream);

/* ------------------------------------------------------------------ *
 * Conv1d / Linear as an MFMA implicit GEMM with fused epilogue.
 *   y = res_scale*res + res2
 *       + out_scale * drop(mask_out(act(W * x_masked + bias)))
 * Replaces nn.Conv1d / nn.Linear + the elementwise ops around them in
 *   esp/transformer/multi_layer_conv.py:52-67   (k=9, ReLU, masks, dropout)
 *   modules/variance_adaptor.py:31-36           (k=3/5, ReLU)
 *   modules/frame_prior.py:85-89                (k=17)
 *   modules/denoiser.py:58-64,69-83             (k=3 dilated, 1x1)
 *   esp/transformer/attention.py:32-35,229      (Linear 256x256)
 *   modules/mdn.py:50-78, modules/prompt_encoder.py:45-51 (Linear heads/MLP)
 *   vocoders/bigvgan.py:24-47,84-118            (dilated Conv1d, and
 *       ConvTranspose1d re-expressed as a 3-tap conv with Cout*stride
 *       outputs -- see promptttspp_amd/vocoders/bigvgan.py)
 * The data gradient is the same kernel on a mode-1 packed weight.
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

/* act == PTPP_ACT_GATE (bf16, Cout % 8 == 0, 16-byte aligned rows): the weight rows / bias / residual are
 * laid out as interleaved groups [4 gate channels | their 4 filter channels]; the epilogue writes
 * y[.., Cout/2] = sigmoid(gate) * tanh(filter) of (conv + bias + res) -- modules/denoiser.py:69-83 without
 * materialising the 2C-channel pre-activation (inference). */
int ptpp_conv1d_fwd(const ptpp_conv1d_args* a, void* stream);

/* Full form: second residual, residual scale (AMP-block mean of
 * vocoders/bigvgan.py:124-128) and fused dropout on the conv term. */
int ptpp_conv1d_fwd_ex(const ptpp_conv1d_args* a, const void* res2, int ldr2,
                       float res_scale, float drop_p, uint64_t drop_seed,
                       void* stream);
/* Row-tile form for the frame-level layers with 256 output channels (bf16, Cin % 64 == 0, ks >= 3, act none / ReLU, no
 * dropout / second residual; csrc/conv1d_rt.hip): a workgroup owns 64-128 rows of one utterance with all channels, the x
 * window of a channel chunk is fetched once and the weights arrive as one contiguous operand stream (`wstream`: the weight in
 * pack mode 3, or mode 4 for the data gradient -- a->wp is ignored).  Bit-identical to ptpp_conv1d_fwd_ex(a, NULL, 0,
 * res_scale, 0, 0) on the mode-0 / mode-1 operand.  Serves modules/frame_prior.py:85-89 (k = 17), modules/variance_adaptor.py:31-36
 * (k = 5) and the data gradient of the DiffNet dilated conv, modules/denoiser.py:58-64. */
int ptpp_conv1d_rt_supported(int cin, int cout, int ks, int dil, int act, int dtype);
int ptpp_conv1d_rt_fwd(const ptpp_conv1d_args* a, const void* wstream, float res_scale, void* stream);
/* ... with a second output from the same epilogue: aux[row * ldaux + n] = dtype(y[row][n] * aux_scale) computed from the ROUNDED
 * y, zero on rows past an utterance's end when a->lengths is given (the DiffNet backward takes the next layer's
 * output-projection gradient operand gx / sqrt2 from here instead of a separate pass over gx). */
int ptpp_conv1d_rt_fwd_aux(const ptpp_conv1d_args* a, const void* wstream, float res_scale, void* aux, int ldaux, float aux_scale,
                           void* stream);
/* ... and with per-utterance column sums of the ROUNDED output from the same epilogue (round 6; the DiffNet backward needs
 * sum_t gx[l][b, t, :] of every layer for the step-projection gradients, modules/denoiser.py:72 differentiated -- a 307 MB pass
 * over the stored gradients before): colpart is (B, ceil(T / 32), 256) f32, NOT initialised by the caller; the tile that starts at
 * row t0 writes its sum to slot t0 / 32 and zeros to the other slots it covers, so ptpp_colsum_batch(colpart, out, B,
 * ceil(T / 32), 256, PTPP_F32) gives the sums in a fixed order (bit-reproducible).  Only where the launch takes the
 * global-weights form: ptpp_conv1d_rt_colpart_supported. */
int ptpp_conv1d_rt_colpart_supported(int cin, int ks, int dil, int B, int T);
int ptpp_conv1d_rt_fwd_cs(const ptpp_conv1d_args* a, const void* wstream, float res_scale, void* aux, int ldaux, float aux_scale,
                          float* colpart, void* stream);

/* Row-tile form for the phone-level feed-forward convs of the Conformer blocks (modules/esp/transformer/multi_layer_conv.py:52-67:
 * Conv1d 256 -> 1024 -> 256, k = 9; round 6): bf16, Cout %% 256 == 0 (column groups of 256 over one x window), Cin %% 128 == 0, ks = 9,
 * act none / ReLU, masks, residual, and dropout on the conv term (the epilogue of ptpp_conv1d_fwd_ex).  wstream: the weight in pack
 * mode 3 (forward) / 4 (data gradient); more than 256 operand rows are laid out as one stream per group of 256, back to back.
 * workspace (optional, as ptpp_conv1d_fwd_ws): with few blocks and a long Cin the input channels are split over blocks, f32
 * partial sums go through it and a finishing launch applies the epilogue.  Within a split the accumulation order is the tile
 * kernel's; across splits it is this kernel's own (results agree with ptpp_conv1d_fwd_ws to f32 summation order, not bit for bit). */
int ptpp_conv1d_rt_ex_supported(int cin, int cout, int ks, int dil, int act, int dtype);
/* Row count from which the whole-unit drivers (ptpp_conv_ln_stack_*, ptpp_diffnet_stack_*) take the row-tile kernel for a launch
 * whose operand stream was handed over; the Python layer pushes ITS threshold here so that both layers decide alike (default:
 * PTPP_CONV_RT_MIN_ROWS or 24 576). */
int ptpp_conv_rt_set_min_rows(long long rows);
int ptpp_conv1d_rt_fwd_ex(const ptpp_conv1d_args* a, const void* wstream, float res_scale, float drop_p, uint64_t drop_seed,
                          void* workspace, size_t workspace_bytes, void* stream);
/* ... where y is only an intermediate of a backward chain (the Conformer feed-forward: gradient w.r.t. the hidden activation ->
 * ReLU / dropout backward -> next data gradient): dz = [saved > 0 and t < lengths[b]] * bf16(y) / (1 - drop_p), the arithmetic of
 * ptpp_epilogue_bwd(y, saved, dz, lengths, .., 1, 1, 1, drop_p) from the conv's epilogue -- a->y is scratch (written only when the
 * launch had to be split).  a: no bias / residual / activation / output mask, ldy = Cout, lengths required. */
int ptpp_conv1d_rt_fwd_ex_relu_bwd(const ptpp_conv1d_args* a, const void* wstream, const void* saved, void* dz, float drop_p,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* ... without the finishing pass of a launch that is split over Cin: on return *nsplit > 1 means `workspace` holds the raw f32
 * partial sums [nsplit][B][T][Cout] and a->y was NOT written (the consumer sums them: ptpp_layernorm_bwd_add_splitk);
 * *nsplit == 1: a->y is complete.  a: no bias / residual / activation / scale (the output mask is the consumer's to apply). */
int ptpp_conv1d_rt_fwd_ex_partial(const ptpp_conv1d_args* a, const void* wstream, void* workspace, size_t workspace_bytes,
                                  int* nsplit, void* stream);

/* The same with an optional scratch (16-byte aligned device memory, NULL = none): layers with few
 * output tiles and a long K (Conformer FFN k = 9, BERT FFN) are then split over K -- f32 partial sums
 * of B*T*Cout elements per split go through the scratch and a second launch adds them and applies the
 * epilogue.  Results are deterministic; 64 MiB covers every shape of the path (the conv1d_wgrad
 * workspace of the same stream can be shared: the calls are ordered on the stream). */
int ptpp_conv1d_fwd_ws(const ptpp_conv1d_args* a, const void* res2, int ldr2,
                       float res_scale, float drop_p, uint64_t drop_seed,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Weight / bias gradient (f32 accumulate, f32 output; ACCUMULATES: the
 * caller zero-fills dw/dbias or hands in a gradient buffer to add to):
 *   dw[Cout][Cin][ks] += sum_{b,t} dy[b,t,co] * x[b, t + j*dil - pad, ci]
 *   dbias[Cout]       += sum_{b,t} dy[b,t,co]                 (dbias nullable)
 * The rows are split across blocks.  workspace: optional 16-byte aligned
 * device scratch (>= 4*Cout*Cin*ks bytes to be used; 64 MiB is plenty): the
 * split partials are stored there and summed in a fixed order by a second
 * kernel (deterministic dw).  NULL -> partials combined with f32 atomics. */
int ptpp_conv1d_wgrad(const void* x, const void* dy, float* dw, float* dbias,
                      const int32_t* lengths, int B, int T, int Cin, int Cout,
                      int ks, int dil, int pad, int ldx, int lddy, int in_mask,
                      int dtype, void* workspace, size_t workspace_bytes,
                      void* stream);

/* The weight gradients of SEVERAL layers of one shape (the layers of a stack: B, T, Cin, Cout, ks, row strides, mask shared;
 * operands, dilation, padding and accumulation targets per problem; probs is a HOST array) after the stack's backward has
 * produced all their operands.  When the tiles of all problems fill the machine (bf16, Cin, Cout > 64) ONE launch computes
 * them with every dw element owned by exactly one block that walks all rows in a fixed order and adds its total into dw: no
 * split-K partials, no second pass, bit-reproducible.  Otherwise (f32 parity mode, small shapes) it is a loop of
 * ptpp_conv1d_wgrad calls with the workspace. */
typedef struct {
  const void* x;
  const void* dy;
  float* dw;       /* (Cout, Cin, ks) f32, accumulated into */
  float* dbias;    /* (Cout) f32 or NULL                     */
  int32_t dil, pad;
} ptpp_wgrad_problem;
int ptpp_conv1d_wgrad_batched(const ptpp_wgrad_problem* probs, int nprob, const int32_t* lengths,
                              int B, int T, int Cin, int Cout, int ks, int ldx, int lddy,
                              int in_mask, int dtype, void* workspace, size_t workspace_bytes,
                              void* stream);

/* Backward of the conv epilogue (contiguous (B,T,C)):
 *   dz = dy * scale * [t < len] * relu'(y) * dropmask           */
int ptpp_epilogue_bwd(const void* dy, const void* y, void* dz,
                      const int32_t* lengths, int B, int T, int C, float scale,
                      int relu, int out_mask, float drop_p, uint64_t seed,
                      int dtype, void* stream);

/* ------------------------------------------------------------------ *
 * LayerNorm over the channel (last) dimension, biased variance.
 *   s = drop_in(act_in(x)) + res ;  y = drop_out(LN(s)*gamma+beta) * mask
 * Replaces the three LayerNorm variants of SURVEY.md F13:
 *   esp/transformer/layer_norm.py:12-33 (eps 1e-12),
 *   layers/norm.py:19-32 (eps 1e-5, (B,C,T) layout),
 *   modules/frame_prior.py:22-34 (eps 1e-5),
 * and the elementwise ops fused around them (frame_prior.py:85-89:
 * LN(x + dropout(gelu(z))); variance_adaptor.py:31-36: dropout(LN(.))*mask).
 * act_in: PTPP_ACT_NONE or PTPP_ACT_GELU.  sum_out (s) / mean / rstd are
 * written when non-NULL (saved for backward).
 * ------------------------------------------------------------------ */
int ptpp_layernorm_fwd(const void* x, const void* res, const float* gamma,
                       const float* beta, void* y, void* sum_out, float* mean,
                       float* rstd, const int32_t* lengths, int B, int T, int C,
                       float eps, int out_mask, int act_in, float drop_in_p,
                       uint64_t drop_in_seed, float drop_out_p,
                       uint64_t drop_out_seed, int dtype, void* stream);

/* The weight gradients of SEVERAL layers of DIFFERENT shapes over few rows (the linear / conv layers of one Conformer block
 * at phone level: a few thousand rows, where one split-K launch + reduction per layer is ~30 us of fixed cost for ~2 us of
 * arithmetic), each ACCUMULATED into its dw / dbias.  probs is a HOST array.  bf16 problems with Cin, Cout > 64 and at most
 * 256 row chunks of 32: ONE launch per 16 problems, every dw tile owned by one block that walks all rows of its problem in a
 * fixed order (no partials, no second pass, bit-reproducible).  Anything else: a loop of ptpp_conv1d_wgrad.
 * lengths non-NULL = that problem's input mask (x rows at or after lengths[b] count as zero). */
typedef struct {
  const void* x;
  const void* dy;
  float* dw;       /* (Cout, Cin, ks) f32, accumulated into */
  float* dbias;    /* (Cout) f32 or NULL                     */
  const int32_t* lengths;
  int32_t B, T, Cin, Cout, ks, dil, pad, ldx, lddy;
} ptpp_wgrad_gproblem;
int ptpp_conv1d_wgrad_grouped(const ptpp_wgrad_gproblem* probs, int nprob, int dtype,
                              void* workspace, size_t workspace_bytes, void* stream);
/* The same with the 1 x 1 problems of the bf16 fast path on `stream2` when it is not NULL (the two launches of the grouped form
 * are independent: on two streams they share the chip instead of queueing).  Both streams must already be ordered after the
 * producers of the operands; the fallback (per-problem kernels, shared workspace) uses `stream` only. */
int ptpp_conv1d_wgrad_grouped2(const ptpp_wgrad_gproblem* probs, int nprob, int dtype,
                               void* workspace, size_t workspace_bytes, void* stream, void* stream2);

/* Reduction scratch (ptpp_layernorm_bwd, ptpp_col_reduce, ptpp_bn_act_bwd): f32 atomics from
 * many blocks on one cache line serialise on MI355X (~50 ns per block visit), so per-column sums
 * go through a caller-owned scratch: block b adds its totals into replica b % 32, and a small
 * finishing launch sums the replicas, delivers the result and zeroes the scratch again.
 *   - PTPP_RED_SCRATCH_BYTES(C) bytes of device memory, 16-byte aligned;
 *   - all ZERO before the first call (the kernels leave it zero);
 *   - one per stream: calls that may overlap must not share it, and it must not double as the
 *     conv1d_wgrad workspace. */
#define PTPP_RED_REPLICAS 32
#define PTPP_RED_SCRATCH_BYTES(C) (PTPP_RED_REPLICAS * 8 * (size_t)(C))

/* dsum = dL/ds (also the gradient of `res`); dz = dsum*dropmask_in*act_in'(z)
 * when act_in/drop_in were used (z = the forward `x`).  dgamma/dbeta (nullable)
 * are f32 buffers the totals are ADDED to (`+=`, ordered on the stream); scratch
 * is needed when either is non-NULL. */
int ptpp_layernorm_bwd(const void* dy, const void* xsum, const void* z,
                       const float* gamma, const float* mean, const float* rstd,
                       void* dsum, void* dz, float* dgamma, float* dbeta,
                       const int32_t* lengths, int B, int T, int C, int out_mask,
                       int act_in, float drop_in_p, uint64_t drop_in_seed,
                       float drop_out_p, uint64_t drop_out_seed, int dtype,
                       void* scratch, size_t scratch_bytes, void* stream);
/* ... with the neighbouring elementwise passes of a pre-norm residual block folded in (bit for bit what the separate
 * launches produce): dsum = add + T(dL/ds) (`add` = gradient of the residual branch, may be NULL; the LayerNorm input
 * gradient is rounded to the storage type first, like a stored intermediate), and dz = the backward of the epilogue that FED
 * the block (ptpp_epilogue_bwd on dsum as stored): dz = dsum * dz_scale * [t < len when dz_mask] * dropmask_in * act_in'(z). */
int ptpp_layernorm_bwd_add(const void* dy, const void* xsum, const void* z,
                           const float* gamma, const float* mean, const float* rstd,
                           void* dsum, void* dz, const void* add, float dz_scale, int dz_mask, float* dgamma, float* dbeta,
                           const int32_t* lengths, int B, int T, int C, int out_mask,
                           int act_in, float drop_in_p, uint64_t drop_in_seed,
                           float drop_out_p, uint64_t drop_out_seed, int dtype,
                           void* scratch, size_t scratch_bytes, void* stream);
/* ptpp_layernorm_bwd_add whose dy is not a tensor but the split-K partial sums of the conv that produces it (round 6; the
 * Conformer feed-forward backward: data gradient of the k = 9 conv 1024 -> 256, split over Cin, then the LayerNorm backward):
 * dy[row] = dtype(sum_k partials[k][row]), rows t >= partial_lengths[b] zero (NULL: no mask) -- conv_splitk_finish_kernel's
 * arithmetic, so the results are those of the two launches.  bf16 only; no act_in / drop_out. */
int ptpp_layernorm_bwd_add_splitk(const float* partials, int nsplit, const int32_t* partial_lengths, const void* xsum,
                                  const float* gamma, const float* mean, const float* rstd, void* dsum, void* dz, const void* add,
                                  float dz_scale, int dz_mask, float* dgamma, float* dbeta, const int32_t* lengths, int B, int T,
                                  int C, int out_mask, float drop_in_p, uint64_t drop_in_seed, int dtype, void* scratch,
                                  size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * Relative-position multi-head attention for short sequences
 * (esp/transformer/attention.py:63-93,142-206,237-305).
 *   q,k,v: (B,T,*) rows with stride ld, head h at columns [h*dk,(h+1)*dk)
 *   pos:   linear_pos(pos_emb): (2T-1, H*dk) new / (T, H*dk) legacy, stride ldpos
 *   bias_u/bias_v: (H, dk) f32;  ctx: (B,T,H*dk) rows with stride ldctx
 *   probs: (B,H,T,T) f32 softmax output BEFORE dropout (NULL in inference; saved for bwd)
 *   drop_p/drop_seed: dropout on the attention probabilities (BERT's
 *   attention_probs_dropout_prob, transformers BertSelfAttention; the reference trains
 *   encoder.layer[-1].attention, modules/prompt_encoder.py:29-31); 0 = off.  The keep-mask is a
 *   function of (seed, element index): ptpp_attention_bwd regenerates it from the same pair.
 * ------------------------------------------------------------------ */
int ptpp_attention_fwd(const void* q, const void* k, const void* v,
                       const void* pos, const float* bias_u, const float* bias_v,
                       void* ctx, float* probs, const int32_t* lengths, int B,
                       int T, int H, int dk, int ld, int ldpos, int ldctx,
                       int variant, float drop_p, uint64_t drop_seed, int dtype,
                       void* stream);

/* dS: (B,H,T,T) f32 workspace; dq/dk_out/dv_out: (B,T,*) rows, stride lddq;
 * dpos: (2T-1, H*dk) f32 (overwritten); du/dvb: (H*dk) f32 buffers the totals are
 * ADDED to (through the reduction scratch described at ptpp_layernorm_bwd, here
 * PTPP_RED_SCRATCH_BYTES(H*dk) bytes; needed for the NEW variant).  Variants: NEW and PLAIN. */
int ptpp_attention_bwd(const void* q, const void* k, const void* v,
                       const void* pos, const float* bias_u, const float* bias_v,
                       const float* probs, const void* dctx, float* dS, void* dq,
                       void* dk_out, void* dv_out, float* dpos, float* du,
                       float* dvb, const int32_t* lengths, int B, int T, int H,
                       int dk, int ld, int ldpos, int lddctx, int lddq,
                       int variant, float drop_p, uint64_t drop_seed, int dtype,
                       void* scratch, size_t scratch_bytes, void* stream);
/* Windowed relative-position attention of the FFT-block encoder plug-in (modules/transformer.py:59-137, Shaw et al.):
 *   score[i,j] = (q_i . k_j + [|j-i| <= w] q_i . emb_k[j-i+w]) / sqrt(dk);  ctx_i = sum_j P[i,j] v_j + sum_r P[i,i+r-w] emb_v[r]
 * emb_k / emb_v: (2w+1, dk) f32, shared by the heads.  Masking, probability output and dropout as ptpp_attention_fwd (PLAIN).
 * Backward: dS (B,H,T,T) f32 scratch/out, dq / dk / dv as ptpp_attention_bwd, demb_k / demb_v (2w+1, dk) f32 OVERWRITTEN. */
int ptpp_attention_win_fwd(const void* q, const void* k, const void* v, const float* emb_k, const float* emb_v, void* ctx,
                           float* probs, const int32_t* lengths, int B, int T, int H, int dk, int ld, int ldctx, int window,
                           float drop_p, uint64_t drop_seed, int dtype, void* stream);
int ptpp_attention_win_bwd(const void* q, const void* k, const void* v, const float* emb_k, const float* emb_v, const float* probs,
                           const void* dctx, float* dS, void* dq, void* dk_out, void* dv_out, float* demb_k, float* demb_v,
                           const int32_t* lengths, int B, int T, int H, int dk, int ld, int lddctx, int lddq, int window,
                           float drop_p, uint64_t drop_seed, int dtype, void* stream);

/* ------------------------------------------------------------------ *
 * Length regulator as a gather / segment-sum instead of the reference's
 * dense 0/1 path matmul (utils/model.py:37-47, variance_adaptor.py:129-131).
 *   cum: (B, Tp) int32 inclusive cumulative durations
 *   y[b,f,:] = x[b, p(f), :],  p(f) = #{p : cum[b,p] <= f}   (0 past the end)
 * ------------------------------------------------------------------ */
int ptpp_length_regulate_fwd(const void* x, const int32_t* cum, void* y, int B,
                             int Tp, int Tf, int C, int dtype, void* stream);
int ptpp_length_regulate_bwd(const void* dy, const int32_t* cum, void* dx, int B,
                             int Tp, int Tf, int C, int dtype, void* stream);

/* y = drop(x*scale + pe[t,:]) -- positional encodings (modules/embedding.py:91,
 * esp/transformer/embedding.py:255,326); pe (T,C) f32 or NULL. */
int ptpp_posenc_fwd(const void* x, const float* pe, void* y, int B, int T, int C,
                    float scale, float drop_p, uint64_t seed, int dtype,
                    void* stream);

/* ------------------------------------------------------------------ *
 * DiffNet residual block glue (modules/denoiser.py:69-83,136-140).
 * ------------------------------------------------------------------ */
/* g = sigmoid(a[:, :C]) * tanh(a[:, C:]) ; a: (rows, 2C), g: (rows, C) */
int ptpp_gate_fwd(const void* a, void* g, int64_t rows, int C, int dtype,
                  void* stream);
/* da (rows, row stride ldda >= 2C) from dg; lets the caller write each layer's
 * gradient straight into its slice of the batched conditioner gradient. */
int ptpp_gate_bwd(const void* a, const void* dg, void* da, int64_t rows, int C,
                  int ldda, int dtype, void* stream);
/* xn = o ? (x + o[:, :C])/sqrt2 : x ; skip(f32) = (init?0:skip) + o[:, C:] ;
 * yin = xn + dnext[b,:]  (o, yin nullable) */
int ptpp_diffnet_post_fwd(const void* o, const void* x, float* skip,
                          const float* dnext, void* xn, void* yin, int B, int T,
                          int C, int init, int dtype, void* stream);
/* The layer's 1 x 1 output projection AND the tail above in one launch (bf16): o = conv1x1(a->x) is never stored;
 * a->Cout = 2C, a->ks = 1; a->res / a->y are ignored (x and xn take their place).  Bit-identical to
 * ptpp_conv1d_fwd followed by ptpp_diffnet_post_fwd.  _supported: bf16, C % 128 == 0, Cin % 64 == 0.
 * Reference: modules/denoiser.py:78-83 (output_projection, chunk, residual / skip). */
int ptpp_conv1d_diffnet_post_supported(int C, int cin, int dtype);
int ptpp_conv1d_diffnet_post(const ptpp_conv1d_args* a, const void* x, float* skip,
                             const float* dnext, void* xn, void* yin, int init,
                             void* stream);
/* One reverse step of the DDPM sampler (modules/diffusion.py:283-302: predict_start_from_noise, clamp to [-1, 1],
 * q_posterior, + sigma_t * noise) on (B, per_b) f32 elements; t: (B) int64 step indices ON THE DEVICE; the five
 * schedule buffers are the module's f32 tables (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_mean_coef1, posterior_mean_coef2, posterior_log_variance_clipped); eps in eps_dtype; noise nullable. */
int ptpp_ddpm_step(const float* x, const void* eps, const float* noise, const int64_t* t,
                   const float* sra, const float* srm1, const float* c1, const float* c2,
                   const float* logvar, float* out, int B, int64_t per_b, int eps_dtype,
                   void* stream);
/* ... also writing out_lp = out rounded to eps_dtype (nullable): the denoiser's input of the next reverse step. */
int ptpp_ddpm_step_lp(const float* x, const void* eps, const float* noise, const int64_t* t,
                      const float* sra, const float* srm1, const float* c1, const float* c2,
                      const float* logvar, float* out, void* out_lp, int B, int64_t per_b, int eps_dtype,
                      void* stream);
/* Everything between two DiffNet stacks of the reverse-diffusion loop in ONE launch (bf16, C = 256, M % 16 == 0, M <= 96;
 * csrc/sampler_head.hip): skip projection + ReLU + output projection (modules/denoiser.py:147-152) -> eps, the reverse step
 * (modules/diffusion.py:283-302, as ptpp_ddpm_step) -> x_out, and -- when win_p is given -- the NEXT step's input projection
 * + ReLU (denoiser.py:131) -> h0 and its first layer's input yin0 = h0 + ds0[b] (denoiser.py:76).  Same operand order and bf16
 * rounding points as the seven launches it replaces; the update is the reference's unfused f32 sequence (ptpp_ddpm_step: equal). */
typedef struct {
  const void* s;            /* (B*T, C) bf16: skip sum / sqrt(L) (ptpp_diffnet_stack_fwd: skip_scaled)            */
  const void* ws_p;         /* skip projection C -> C, mode-0 operand (C, 1, C) bf16                             */
  const float* ws_b;        /* (C)                                                                               */
  const void* wo_p;         /* output projection C -> M, mode-0 operand (M, 1, C)                                */
  const float* wo_b;        /* (M)                                                                               */
  const float* x;           /* (B*T, M) f32: x_t                                                                 */
  const float* noise;       /* (B*T, M) f32 or NULL (t == 0)                                                     */
  const int64_t* t;         /* (B) step indices on the device                                                    */
  const float *sra, *srm1, *c1, *c2, *logvar; /* the schedule tables of ptpp_ddpm_step                           */
  float* x_out;             /* (B*T, M) f32: x_{t-1}                                                             */
  const void* win_p;        /* input projection M -> C, mode-0 operand (C, 1, M padded to 32), or NULL           */
  const float* win_b;       /* (C)                                                                               */
  const float* ds0;         /* (B, C) f32: layer 0's step projection at the NEXT step                            */
  void* h0;                 /* (B*T, C) bf16 out                                                                 */
  void* yin0;               /* (B*T, C) bf16 out                                                                 */
  int32_t B, T, C, M, dtype;
} ptpp_sampler_head_args;
int ptpp_sampler_head_supported(int C, int M, int dtype);
int ptpp_sampler_head(const ptpp_sampler_head_args* a, void* stream);
/* Backward of the same layer: dg = conv1x1(a->x = do, a->wp = W_out^T) (a->Cout = C, never stored) with
 * ptpp_gate_bwd in its epilogue -- da (row stride ldda >= 2C) from the saved pre-activation act (B, T, 2C).
 * Bit-identical to ptpp_conv1d_fwd followed by ptpp_gate_bwd.  _supported: bf16, C % 8 == 0, Cin % 64 == 0. */
int ptpp_conv1d_gate_bwd_supported(int C, int cin, int dtype);
int ptpp_conv1d_gate_bwd(const ptpp_conv1d_args* a, const void* act, void* da, int ldda,
                         void* stream);
/* ... on the row-tile engine (csrc/conv1d_rt.hip, the global-weights form): the projection weight as an operand stream
 * (`wstream`: the (2C, C, 1) weight in pack mode 4; a->wp is ignored), a->ks = 1, a->Cout = C = 256, a->Cin % 128 == 0, bf16.
 * Same arithmetic in the same order as ptpp_conv1d_gate_bwd: bit-identical da. */
int ptpp_conv1d_rt_gate_bwd_supported(int C, int cin, int dtype);
int ptpp_conv1d_rt_gate_bwd(const ptpp_conv1d_args* a, const void* wstream, const void* act, void* da, int ldda, void* stream);

/* The DiffNet dilated conv (+ conditioner slice as `res`) of the TRAINING forward with the gate in the epilogue and the
 * pre-activation kept for the backward (modules/denoiser.py:76-79): weights, bias and `res` in the gate-interleaved row order
 * (ptpp_pack_conv_weight mode 2: rows [4 gate | their 4 filter partners]); y = g (B, T, C) and a_out (B, T, 2C) in the
 * STANDARD [gate | filter] order, row stride lda -- bit for bit what ptpp_conv1d_fwd followed by ptpp_gate_fwd produce. */
int ptpp_conv1d_gate_fwd_save_supported(int C, int cin, int dtype);
int ptpp_conv1d_gate_fwd_save(const ptpp_conv1d_args* a, void* a_out, int lda, void* stream);
/* One DiffNet residual layer as ONE launch (bf16, C = 256; csrc/diffnet_layer.hip; reference modules/denoiser.py:69-83):
 *   a = dilated_conv_k3(yin) + cond;  g = sigmoid(a[:C]) * tanh(a[C:]);  o = output_projection(g) (masked);
 *   xn = (x + o[:C]) / sqrt2;  skip = (init ? 0 : skip) + o[C:];  yin_next = xn + dnext[b].
 * Bit-identical to ptpp_conv1d_gate_fwd_save (a_out / g_out given: training) or ptpp_conv1d_fwd with PTPP_ACT_GATE
 * (a_out = g_out = NULL: inference) followed by ptpp_conv1d_diffnet_post.  The layer's two weights arrive as ONE operand
 * stream (ptpp_diffnet_pack_wstream: 64 stages of 16 KiB in consumption order, each already the LDS image of its MFMA
 * fragments) built from the mode-2 operand of the dilated conv and the mode-0 operand of the output projection; dil_b and
 * cond (row stride ldc, this layer's 2C channels) are in the gate-interleaved order of mode 2. */
typedef struct {
  const void* yin;        /* (B, T, C) bf16: x + diffusion-step projection (input of the dilated conv)   */
  const void* x;          /* (B, T, C) bf16: residual input                                              */
  const void* cond;       /* (B, T, .) bf16, row stride ldc: conditioner slice of this layer (2C ch.)    */
  const void* wstream;    /* ptpp_diffnet_wstream_bytes(C) bytes of this layer                           */
  const float* dil_b;     /* (2C) f32, gate-interleaved order                                            */
  const float* out_b;     /* (2C) f32                                                                    */
  const float* dnext;     /* (B, C) f32 step projection of the NEXT layer, or NULL (then yin_next unused) */
  float* skip;            /* (B, T, C) f32, accumulated in place                                         */
  void* xn;               /* (B, T, C) bf16 out                                                          */
  void* yin_next;         /* (B, T, C) bf16 out or NULL                                                  */
  void* a_out;            /* (B, T, 2C) bf16 out, STANDARD [gate | filter] order, or NULL (inference)    */
  void* g_out;            /* (B, T, C) bf16 out, or NULL together with a_out                             */
  const int32_t* lengths; /* (B) or NULL: rows past an utterance's end have o = 0 (training)             */
  int32_t B, T, C, dil, ldc, init, dtype;
  void* skip_scaled;      /* (B, T, C) bf16 out or NULL: bf16(skip * skip_scale) after this layer's update -- on the last  */
  float skip_scale;       /* layer, the input of the skip projection (sum / sqrt(L), denoiser.py:150) without two launches  */
  const void* condx;      /* NULL, or the conditioner INPUT (B, T, 256) bf16, row stride ldcx: its 1 x 1 projection           */
  int32_t ldcx;           /* (denoiser.py:76) then runs inside the launch as 16 more stages of the first matrix pass --       */
                          /* `cond` is unused, `wstream` is the ptpp_diffnet_pack_wstream_cond form (80 stages) and `dil_b`   */
                          /* the SUM of the dilated-conv and conditioner biases.  The pre-activation is then accumulated in   */
                          /* f32 before its single rounding (the precomputed slice was rounded to bf16 first).                */
} ptpp_diffnet_layer_args;
int ptpp_diffnet_layer_supported(int C, int dtype);
int64_t ptpp_diffnet_wstream_bytes(int C);
/* dil_wp / out_wp: HOST arrays of L device pointers (mode-2 (2C, 3, C) and mode-0 (2C, 1, C) bf16 operands); wstream:
 * L * ptpp_diffnet_wstream_bytes(C) bytes, layer l at l * that. */
int ptpp_diffnet_pack_wstream(const void* const* dil_wp, const void* const* out_wp, void* wstream, int L, int C, void* stream);
/* ... with the conditioner projection's weights (mode-2 (2C, 1, 256) operands) as stages 48-63: the `condx` form of the layer */
int64_t ptpp_diffnet_wstream_bytes_cond(int C);
int ptpp_diffnet_pack_wstream_cond(const void* const* dil_wp, const void* const* cond_wp, const void* const* out_wp, void* wstream, int L,
                                   int C, void* stream);
int ptpp_diffnet_layer_fwd(const ptpp_diffnet_layer_args* a, void* stream);
/* diagnostics for tools/ only: dbg bit 0 = per-block clock stamps (6 x uint64 per block: start, first stage landed, end of the
 * dilated conv, end of the gate epilogue, end of the output projection, end), bits 1-3 switch work off (results invalid). */
int ptpp_diffnet_layer_fwd_dbg(const ptpp_diffnet_layer_args* a, int dbg, void* stamps, void* stream);
/* dout (rows, 2C) = [gx/sqrt2 | gskip], masked rows zero */
int ptpp_diffnet_post_bwd(const void* gx, const void* gskip, void* dout,
                          const int32_t* lengths, int B, int T, int C, int dtype,
                          void* stream);
/* The skip halves of ALL layers' dout in one launch: do_all[l][row][C:] = gskip[row] (masked) for l < L, and the residual half
 * of the LAST layer (whose incoming gx is zero) cleared; the other residual halves come from ptpp_conv1d_rt_fwd_aux. */
int ptpp_diffnet_post_bwd_fill(const void* gskip, void* do_all, const int32_t* lengths, int B, int T, int C, int L, int dtype,
                               void* stream);
/* Harmonic-plus-noise source of the F0-aware vocoder (promptttspp/vocoders/nsf.py:31-206: SineGen._f02sine + SineGen.forward +
 * SourceModuleHnNSF.forward's Linear + tanh) in three launches of one kernel.  f0: (B, L) f32 Hz at the sample rate (0 = unvoiced); rand_ini:
 * (B, dim) f32 initial phases (column 0 zero); noise: (B, L, dim) f32 standard-normal draws; w: (dim) f32 + bias: the merge
 * layer; out: (B, L) f32.  dim = harmonic_num + 1 in {9, 1} (ptpp_nsf_source_supported).  Both prefix sums run in a
 * fixed order (bit-reproducible; not torch.cumsum's order: csrc/nsf.hip explains why the output does not depend on it). */
int ptpp_nsf_source_supported(int dim);
size_t ptpp_nsf_source_scratch_bytes(int B, int L, int dim);   /* the table of range sums between the three launches */
int ptpp_nsf_source(const float* f0, const float* rand_ini, const float* noise, const float* w, float bias, float* out,
                    int B, int L, int dim, float sampling_rate, float sine_amp, float noise_std, float voiced_threshold,
                    void* scratch, size_t scratch_bytes, void* stream);

/* out[b,c] = sum_t x[b,t,c] (f32) */
int ptpp_colsum_batch(const void* x, float* out, int B, int T, int C, int dtype,
                      void* stream);

/* ------------------------------------------------------------------ *
 * BatchNorm with batch statistics + activation on channels-last rows, GLU,
 * depthwise Conv1d, 3x3/stride-2 im2col.  Conformer convolution module
 * (esp/conformer/convolution.py:58-85: GLU -> depthwise k7 -> BatchNorm1d ->
 * Swish) and the GST reference encoder (modules/reference_encoder.py:65-81:
 * Conv2d 3x3 s2 -> BatchNorm2d -> ReLU; the Conv2d itself is im2col + the
 * MFMA GEMM above).  `rows` = all rows, padded positions included, exactly
 * like the reference's train-mode BatchNorm.
 * ------------------------------------------------------------------ */
/* out[c] = sum_r x[r,c]  (mean == NULL)  or  sum_r (x[r,c]-mean[c])^2 ; out (C f32) is overwritten.
 * scratch: the reduction scratch described at ptpp_layernorm_bwd. */
int ptpp_col_reduce(const void* x, const float* mean, float* out, int64_t rows,
                    int C, int dtype, void* scratch, size_t scratch_bytes,
                    void* stream);
/* Training-mode statistics in one call (4 launches): mean[c], rstd[c] = rsqrt(biased var + eps) over
 * all rows (two passes: sums, then centred squares), and -- when the pointers are non-NULL -- the
 * running estimates updated in place like torch.nn.BatchNorm: running = (1-momentum)*running +
 * momentum*{mean, unbiased var}.  scratch: the reduction scratch described at ptpp_layernorm_bwd. */
int ptpp_bn_stats(const void* x, int64_t rows, int C, float momentum, float eps,
                  float* running_mean, float* running_var, float* mean, float* rstd,
                  int dtype, void* scratch, size_t scratch_bytes, void* stream);
/* y = act(gamma*(x-mean)*rstd + beta), act in {NONE, RELU, SWISH} */
int ptpp_bn_act_fwd(const void* x, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, void* y, int64_t rows,
                    int C, int act, int dtype, void* stream);
/* sums (2C f32, overwritten): [sum g | sum g*xhat], g = dy*act'(.) -> dbeta, dgamma;
 * dx = gamma*rstd*(g - (sum_g + xhat*sum_gx)/rows) (train) or gamma*rstd*g (eval) */
int ptpp_bn_act_bwd(const void* x, const void* dy, const float* mean,
                    const float* rstd, const float* gamma, const float* beta,
                    float* sums, void* dx, int64_t rows, int C, int act, int train,
                    int dtype, void* scratch, size_t scratch_bytes, void* stream);
/* ... with the two sums also ADDED to the parameter gradients (dbeta_acc / dgamma_acc, (C) f32 each or NULL) by the same
 * finishing launch (round 6: the Conformer block's backward added them with a tensor op per block before) */
int ptpp_bn_act_bwd_acc(const void* x, const void* dy, const float* mean,
                        const float* rstd, const float* gamma, const float* beta,
                        float* sums, float* dbeta_acc, float* dgamma_acc, void* dx, int64_t rows, int C, int act,
                        int train, int dtype, void* scratch, size_t scratch_bytes, void* stream);
/* u = h[:, :C] * sigmoid(h[:, C:]) and its backward */
int ptpp_glu_fwd(const void* h, void* u, int64_t rows, int C, int dtype, void* stream);
int ptpp_glu_bwd(const void* h, const void* du, void* dh, int64_t rows, int C,
                 int dtype, void* stream);
/* ... on (B, T, .) rows with the gradient of rows t >= lengths[b] written as zero (the mask of the pointwise conv that produced h,
 * modules/esp/conformer/convolution.py:58-85 differentiated): one launch instead of glu_bwd + a masking pass; lengths NULL = no mask */
int ptpp_glu_bwd_masked(const void* h, const void* du, void* dh, const int32_t* lengths, int B, int T, int C, int dtype,
                        void* stream);
/* depthwise conv over time, w: (C, ks) f32, "same" padding, output rows t >= len
 * zeroed; flip=1: data gradient (input = dy, masked rows ignored). ks in {7,15,31} */
int ptpp_dwconv1d(const void* u, const float* w, const float* bias, void* y,
                  const int32_t* lengths, int B, int T, int C, int ks, int flip,
                  int dtype, void* stream);
int ptpp_dwconv1d_wgrad(const void* u, const void* dy, float* dw, float* dbias,
                        const int32_t* lengths, int B, int T, int C, int ks,
                        int dtype, void* stream);
/* x: (B,H,W,C) channels-last -> col: (B*Ho*Wo, 9*C), Ho=(H-1)/2+1; and its adjoint */
int ptpp_im2col3x3s2(const void* x, void* col, int B, int H, int W, int C,
                     int dtype, void* stream);
int ptpp_col2im3x3s2(const void* dcol, void* dx, int B, int H, int W, int C,
                     int dtype, void* stream);
/* x with ONE channel (B, H, W): col rows get the 8-channel granule [x, 0 x 7] per tap (K = 72), i.e. ptpp_im2col3x3s2 of x
 * zero-padded to 8 channels without materialising the padded input. */
int ptpp_im2col3x3s2_c1(const void* x, void* col, int B, int H, int W, int dtype,
                        void* stream);

/* ------------------------------------------------------------------ *
 * GRU cell gate algebra of the GST reference encoder
 * (modules/reference_encoder.py:108-123, torch.nn.GRU semantics), f32:
 *   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n),
 *   h' = (1 - z) n + z h;  rows with step >= lens[b] keep h (packed sequence).
 *   gi: (B, 3H) rows with stride ldgi (a time slice of W_ih x + b_ih);
 *   gh: (B, 3H) = W_hh h + b_hh; h, hout: (B, H).  lens nullable.
 * Backward recomputes the gates: dgi (rows with stride lddgi), dgh, dh
 * (the direct dh'/dh term; the W_hh path is the caller's GEMM).
 * ------------------------------------------------------------------ */
int ptpp_gru_gate_fwd(const float* gi, int64_t ldgi, const float* gh,
                      const float* h, const int32_t* lens, int step,
                      float* hout, int B, int H, void* stream);
int ptpp_gru_gate_bwd(const float* gi, int64_t ldgi, const float* gh,
                      const float* h, const int32_t* lens, int step,
                      const float* dhout, float* dgi, int64_t lddgi,
                      float* dgh, float* dh, int B, int H, void* stream);

/* The whole recurrence in ONE launch (H = 128 or 256: ptpp_gru_seq_supported(H) != 0; the reference's gru_units is 256 in
 * conf/model/prompttts_mdn_v2_wo_erg_final.yaml): a block owns one sequence for all L steps with (the first 128 entries per
 * thread of) its slice of W_hh in registers.
 *   gi_all (B, L, 3H) = W_ih x + b_ih;  w_hh (3H, H) and b_hh (3H) f32 as stored in the state dict (gru.weight_hh_l0);
 *   w_hh_t: the transposed copy, (H, ptpp_conv_cin_padded(3H, PTPP_F32)) f32 = ptpp_pack_conv_weight mode 1 of w_hh (read
 *   only for H = 256, may be NULL for H = 128);
 *   saved for the backward: hs_all (L + 1, B, H) with hs_all[0] = 0, gh_all (L, B, 3H) = W_hh h_s + b_hh (rows of steps
 *   >= lens[b] are not written and not read);  hout (B, H) = the last valid hidden state.
 * Backward: dgi_all (B, L, 3H) and dgh_all (L, B, 3H) (zero at steps >= lens[b]); the caller finishes with ONE weight
 * gradient over all steps, dW_hh = dgh_all^T hs_all[:L], db_hh = column sums of dgh_all (ptpp_conv1d_wgrad, rows = L B). */
int ptpp_gru_seq_supported(int H);
int ptpp_gru_seq_fwd(const float* gi_all, const float* w_hh, const float* w_hh_t,
                     const float* b_hh, const int32_t* lens, float* hs_all, float* gh_all,
                     float* hout, int B, int L, int H, void* stream);
int ptpp_gru_seq_bwd(const float* gi_all, const float* w_hh, const int32_t* lens,
                     const float* hs_all, const float* gh_all, const float* dhout,
                     float* dgi_all, float* dgh_all, int B, int L, int H, void* stream);

/* ------------------------------------------------------------------ *
 * Anti-aliased Snake activation, one fused pass (layers/activations.py:22-44,
 * 74-138): replicate-pad -> x2 polyphase Kaiser-sinc up-FIR (12 taps, gain 2)
 * -> x + sin^2(x e^alpha)/(e^alpha + 1e-9) -> 12-tap low-pass, stride 2.
 *   x, y: (B, T, C) channels-last; log_alpha: (C) f32 (log domain, as stored
 *   in the state dict); filt_up/filt_down: HOST pointers to the 12 f32 taps of
 *   `up.filter` / `down.lowpass.filter` (passed as kernel arguments).
 * ------------------------------------------------------------------ */
int ptpp_aa_snake_fwd(const void* x, void* y, const float* log_alpha,
                      const float* filt_up, const float* filt_down, int B,
                      int T, int C, int dtype, void* stream);

/* One whole AMP layer of BigVGAN (vocoders/bigvgan.py:42-47 with layers/activations.py:22-44) in ONE kernel:
 *   y = res_scale * x + out_scale * (conv2(snake2(conv1(snake1(x)))) + b2) [+ res2]
 * x, y, res2: (B, T, C) channels-last `dtype`, contiguous, all utterances of length T; conv1: ks taps, dilation
 * dil, padding dil*(ks-1)/2; conv2: ks taps, padding (ks-1)/2; w1p / w2p: ptpp_pack_conv_weight(mode 0)
 * operands of the (weight-norm folded) (C, C, ks) weights; b1 / b2: (C) f32; log_alpha*: (C) f32 (the Snake
 * parameter in the log domain); up* / dn*: the 12-tap anti-alias filters of act1 / act2 (HOST values, by value).
 * res2 (nullable): the running mean of the AMP blocks (bigvgan.py:124-128).  The x tile and its halo stay in
 * LDS: x is read once and y written once (9 tensor passes when the four stages are separate launches).
 * Built for C in {32, 64}: ptpp_amp_layer_supported(C, dtype) != 0; otherwise PTPP_ENOTSUP.
 * 16-bit dtypes take the conv weights as FRAGMENT STREAMS w1s / w2s (ptpp_amp_pack_wstream of w1p / w2p: the MFMA weight
 * fragments in consumption order, a wave reads 2 KiB contiguous per K step); w1p / w2p are then unused and may be NULL. */
typedef struct {
  const void* x;
  void* y;
  const void* res2;
  const void* w1p;
  const void* w2p;
  const float* b1;
  const float* b2;
  const float* log_alpha1;
  const float* log_alpha2;
  float up1[12], dn1[12], up2[12], dn2[12];
  int32_t B, T, C, ks, dil;
  float out_scale, res_scale;
  int32_t dtype;
  const void* w1s;
  const void* w2s;
} ptpp_amp_layer_args;
int ptpp_amp_layer_supported(int C, int dtype);
int ptpp_amp_layer_fwd(const ptpp_amp_layer_args* a, void* stream);
/* (C, ks, C) packed operand `wp` (ptpp_pack_conv_weight mode 0, no channel padding: C in {32, 64, 128, 256}) -> fragment
 * stream `out` (same number of bytes) for the 16-bit fused AMP layer / ptpp_snake_conv1d_fwd (vocoders/bigvgan.py:24-47: the
 * layer's two weight-normed Conv1d). */
int ptpp_amp_pack_wstream(const void* wp, void* out, int C, int ks, int dtype, void* stream);

/* One half of an AMP layer in the WIDE stages of BigVGAN (C = 128, 256; vocoders/bigvgan.py:42-47 with
 * layers/activations.py:22-44, 74-138): the anti-aliased Snake applied while the conv's input tile is staged,
 *   y = res_scale * res + out_scale * (conv(snake(x)) + bias) [+ res2]
 * x, y, res, res2: (B, T, C) channels-last 16-bit, contiguous (res, res2 nullable); conv: C -> C, ks taps, dilation dil,
 * padding dil*(ks-1)/2; ws: ptpp_amp_pack_wstream of the packed weight; bias (C) f32; log_alpha (C) f32; up / dn: the 12-tap
 * anti-alias filters (HOST values).  The activated tensor never exists in HBM (ptpp_aa_snake_fwd + ptpp_conv1d_fwd: + 2 passes). */
typedef struct {
  const void* x;
  void* y;
  const void* res;
  const void* res2;
  const void* ws;
  const float* bias;
  const float* log_alpha;
  float up[12], dn[12];
  int32_t B, T, C, ks, dil;
  float out_scale, res_scale;
  int32_t dtype;
} ptpp_snake_conv_args;
int ptpp_snake_conv1d_supported(int C, int dtype);
int ptpp_snake_conv1d_fwd(const ptpp_snake_conv_args* a, void* stream);

/* y = (a + b + c) * scale  (b, c nullable) */
int ptpp_add3_scale(const void* a, const void* b, const void* c, void* y,
                    float scale, int64_t n, int dtype, void* stream);

/* Final BigVGAN stage (vocoders/bigvgan.py:129-131): conv_post with ONE
 * output channel (k taps over C channels) followed by tanh.
 *   x: (B, T, C) ; w: (ks, C) f32 ; y: (B, T) f32 */
int ptpp_conv_post_tanh(const void* x, const float* w, float bias, float* y,
                        int B, int T, int C, int ks, int dtype, void* stream);

/* act_post + conv_post + tanh in ONE launch (vocoders/bigvgan.py:129-131; layers/activations.py:22-44, 74-138):
 *   y[b, t] = tanh(bias + sum_{j, c} w[j, c] * aa_snake(x)[b, t + j - ks/2, c])
 * x: (B, T, C) 16-bit; log_alpha (C) f32; filt_up / filt_down: HOST pointers to the 12 taps; w: (ks, C) f32; y: (B, T) f32.
 * Built for C = 32 (the generator's last stage): ptpp_snake_conv_post_supported(C, ks, dtype) != 0. */
int ptpp_snake_conv_post_supported(int C, int ks, int dtype);
int ptpp_snake_conv_post_tanh(const void* x, const float* log_alpha, const float* filt_up, const float* filt_down,
                              const float* w, float bias, float* y, int B, int T, int C, int ks, int dtype, void* stream);

/* Masked L1 mean: out[0] = sum_i |pred_i - target_i| * mask[i / cols] / denom[0] / scale -- the L1 losses of the training step
 * (models/prompttts_mdn_v2_final/model.py:126, 138-170: F.l1_loss on masked tensors; mel / noise, log-F0, V/UV, energy).  pred
 * (rows * cols, contiguous) f32 or bf16, target f32, mask (rows) f32 or NULL, denom / out / gout device scalars (no host sync).
 * One launch each way; the forward adds its partial sums in a fixed order (bit-reproducible).  scratch: ptpp_l1_scratch_bytes()
 * of device memory, zero before the FIRST call (the kernel leaves it zero).  _bwd: dpred (pred's dtype) =
 * sgn(pred - target) * mask * ((gout / scale) / denom), the value autograd's nodes produce for the tensor-op form. */
int64_t ptpp_l1_scratch_bytes(void);
int ptpp_l1_masked_mean_fwd(const void* pred, const float* target, const float* mask, const float* denom, float scale,
                            int64_t rows, int cols, int dtype, float* out, void* scratch, void* stream);
int ptpp_l1_masked_mean_bwd(const void* pred, const float* target, const float* mask, const float* denom, const float* gout,
                            float scale, int64_t rows, int cols, int dtype, void* dpred, void* stream);

/* Dimension-wise mixture-density NLL (modules/mdn.py:81-175, `dim_wise`): log_pi / log_sigma / mu (rows, G, D) f32,
 * target (rows, D), mask (rows) bytes or NULL (0 = masked: loss +inf, zero gradients) -> loss (rows, D) = -logsumexp_g.
 * _bwd recomputes the component log-likelihoods from the inputs and the saved loss. */
int ptpp_mdn_nll_fwd(const float* log_pi, const float* log_sigma, const float* mu,
                     const float* target, const unsigned char* mask, float* loss,
                     int64_t rows, int G, int D, float log_pi_min, float log_sigma_min,
                     void* stream);
int ptpp_mdn_nll_bwd(const float* log_pi, const float* log_sigma, const float* mu,
                     const float* target, const unsigned char* mask, const float* loss,
                     const float* gout, float* d_log_pi, float* d_log_sigma, float* d_mu,
                     int64_t rows, int G, int D, float log_pi_min, float log_sigma_min,
                     void* stream);
/* Zero-phase IIR low-pass of the predicted log-F0 tracks (utils/model.py:164-196, called at app.py:77 /
 * synthesize.py:131): y = flip(lfilter(flip(lfilter(x)))) with zero initial state -- the arithmetic of
 * torchaudio.functional.filtfilt(x, a, b, clamp=False), which the reference uses for tensors.
 *   x, y: rows x T f32 (row stride ldx); lengths (nullable): per-row valid length (the rest is copied);
 *   b, a: HOST coefficient arrays of order + 1 doubles; tmp: rows * T doubles of device scratch. */
int ptpp_filtfilt(const float* x, float* y, double* tmp, const int32_t* lengths,
                  const double* b, const double* a, int order, int rows, int T,
                  int ldx, void* stream);

/* Layout bridges between the reference's (B, C, T) f32 tensors and the
 * library's channels-last (B, T, C) `dtype` tensors. */
int ptpp_bct_to_btc(const float* x, void* y, int B, int C, int T, int dtype,
                    void* stream);
int ptpp_btc_to_bct(const void* x, float* y, int B, int T, int C, int dtype,
                    void* stream);

/* ------------------------------------------------------------------ *
 * Fused multi-tensor optimiser step (trainers/tts.py:206-211):
 * clip_grad_norm_(max_norm) then AdamW, for ALL parameters in two launches
 * and without a host sync.  `refs`: device array of nt records
 * {float* p; const float* g; float* m; float* v; int64 n; int64 block0},
 * block0 = index of the tensor's first 4096-element block in the flat grid;
 * block_map (nullable: binary search over block0 instead) = the record index
 * of every block.  sumsq: PTPP_SUMSQ_SLOTS floats -- the sum of squares is
 * spread over the slots (atomics on one address serialise), ptpp_grad_sumsq
 * overwrites them and ptpp_adamw_step clips by sqrt(sum of the slots).
 * ------------------------------------------------------------------ */
#define PTPP_SUMSQ_SLOTS 64
int ptpp_grad_sumsq(const void* refs, int nt, const int32_t* block_map,
                    long long total_blocks, float* sumsq, void* stream);
/* The same sums without atomics: every block stores its partial into `partials`
 * (total_blocks floats, caller-owned), a second launch adds them per slot in a
 * fixed order -- bit-reproducible, so that the ranks of a data-parallel job clip
 * identical gradients by the identical factor (DDP's invariant of identical
 * parameters on every rank, trainers/tts.py:117,206-211). */
int ptpp_grad_sumsq_det(const void* refs, int nt, const int32_t* block_map,
                        long long total_blocks, float* sumsq, float* partials,
                        void* stream);
int ptpp_adamw_step(const void* refs, int nt, const int32_t* block_map,
                    long long total_blocks, const float* sumsq, const float* lr,
                    float beta1, float beta2, float eps, float weight_decay,
                    int step, float max_norm, void* stream);

/* ------------------------------------------------------------------ *
 * Whole-unit drivers: ONE call issues every launch of a unit of the model (host launch work per kernel drops from a
 * Python / FFI round trip, 9-13 us through ctypes, to the raw launch).  A driver is composed of the entry points above, in
 * the order the per-launch path issues them, so its results are bit-identical to that path (tests compare the two).
 * Pointer tables are HOST arrays of L device pointers.  Slabs: layer l uses slab (l % n_slabs) of yin_all / a_all /
 * g_all -- n_slabs = L keeps every layer's activations for the backward, n_slabs = 2 ping-pongs (inference).
 * ------------------------------------------------------------------ */
/* DiffNet residual stack, forward (reference modules/denoiser.py:69-83 per layer, :136-140 the loop):
 *   yin_0 = h0 + dsteps[0];  per layer l: a = dilconv_l(yin) + cond_all[.., l*2C:(l+1)*2C]; g = sigmoid(a[:C])*tanh(a[C:]);
 *   o = outproj_l(g) (masked);  x = (x + o[:C]) / sqrt2;  skip += o[C:];  yin = x + dsteps[l+1].
 * fused_gate = 1 (bf16 inference): the dilated conv's weights / bias and cond_all are in the interleaved gate order
 * (ptpp_pack_conv_weight mode 2) and the gate runs in the conv epilogue (PTPP_ACT_GATE); a_all is then unused.
 * fused_gate = 2 (bf16 training): the same, and the pre-activation is kept in a_all in the standard channel order
 * (ptpp_conv1d_gate_fwd_save): one launch per layer less, bit-identical to fused_gate = 0. */
typedef struct {
  const void* h0;           /* (B, T, C) dtype                                        */
  const void* cond_all;     /* (B, T, L*2C) dtype, row stride L*2C                    */
  const float* dsteps;      /* (L, B, C) f32 diffusion-step projections               */
  const int32_t* lengths;   /* (B) or NULL (inference: no masks)                      */
  float* skip;              /* (B, T, C) f32 out: sum of the skip halves              */
  const void* const* dil_wp;   /* [L] packed (2C, 3, C) operands (mode 0)             */
  const float* const* dil_b;   /* [L] (2C) f32                                        */
  const void* const* out_wp;   /* [L] packed (2C, 1, C) operands                      */
  const float* const* out_b;   /* [L] (2C) f32                                        */
  void* yin_all;            /* (n_slabs, B, T, C) dtype                               */
  void* a_all;              /* (n_slabs, B, T, 2C) dtype, NULL with fused_gate = 1    */
  void* g_all;              /* (n_slabs, B, T, C) dtype                               */
  void* x_buf[2];           /* two (B, T, C) dtype scratch tensors for x              */
  void* o_buf;              /* (B, T, 2C) dtype scratch, only used where the fused tail is unsupported (f32) */
  int32_t B, T, C, L, cycle, n_slabs, fused_gate, dtype;
  const void* wstream;      /* L * ptpp_diffnet_wstream_bytes(C) bytes from ptpp_diffnet_pack_wstream, or NULL.  With it (and
                             * fused_gate = 1 or 2, bf16, C = 256) every layer is ONE launch (ptpp_diffnet_layer_fwd),
                             * bit-identical to the two launches it replaces */
  void* skip_scaled;        /* (B, T, C) dtype out or NULL: dtype(skip * skip_scale) from the last layer's launch (needs the */
  float skip_scale;         /* one-launch layer, i.e. `wstream`): the input of the skip projection (denoiser.py:150)          */
  const void* condx;        /* NULL, or the conditioner input (B, T, 256) dtype, row stride ldcx: every layer projects it     */
  int32_t ldcx;             /* inside its launch (ptpp_diffnet_layer_args.condx) -- cond_all may then be NULL, `wstream` is   */
                            /* the _cond form and dil_b[l] the summed biases; needs the one-launch layer                      */
  const void* yin0;         /* NULL, or layer 0's input h0 + dsteps[0] (B, T, C) already formed (ptpp_sampler_head): the      */
                            /* driver then skips that launch                                                                  */
} ptpp_diffnet_stack_fwd_args;
int ptpp_diffnet_stack_fwd(const ptpp_diffnet_stack_fwd_args* a, void* stream);

/* Backward of the stack (hand-derived; the per-launch form is functional.DiffNetStackFn.backward).  Weight gradients
 * ACCUMULATE into the f32 targets (e.g. views of the flat gradient buffer) and run on `side_stream` when it is not NULL
 * (forked after the producing launch with a pooled event; the CALLER joins the streams and keeps the slabs alive until
 * then).  Outputs: gx_all[0] = gradient w.r.t. h0;  dcond_all = gradient w.r.t. cond_all;  S[l][b][c] = sum_t
 * gx_all[l][b][t][c] for l < L (the caller forms the step-projection gradients S[l] - S[l+1] / sqrt2 with S[L] = 0). */
typedef struct {
  const void* gS;           /* (B, T, C) dtype: gradient of the skip sum               */
  const void* yin_all;      /* saved by the forward (n_slabs = L)                      */
  const void* a_all;
  const void* g_all;
  const int32_t* lengths;
  const void* const* dil_wpt;  /* [L] mode-1 packed operands of the dilated convs     */
  const void* const* out_wpt;  /* [L] mode-1 packed operands of the output projections */
  float* const* dw_dil;     /* [L] (2C, C, 3) f32 accumulation targets                */
  float* const* db_dil;     /* [L] (2C)                                               */
  float* const* dw_out;     /* [L] (2C, C, 1)                                         */
  float* const* db_out;     /* [L] (2C)                                               */
  void* gx_all;             /* (L+1, B, T, C) dtype scratch / out; slab L is zero-filled here */
  void* do_all;             /* (L, B, T, 2C) dtype scratch                            */
  void* dg_buf;             /* (B, T, C) dtype scratch, only where the fused gate backward is unsupported (f32) */
  void* dcond_all;          /* (B, T, L*2C) dtype out                                 */
  float* S;                 /* (L, B, C) f32 out                                      */
  void* ws_main; size_t ws_main_bytes;   /* split-K scratch of the main stream        */
  void* ws_side; size_t ws_side_bytes;   /* ... of the stream the weight gradients run on */
  void* side_stream;        /* NULL: weight gradients on `stream`                     */
  int32_t B, T, C, L, cycle, dtype;
  int32_t batched_wgrad;    /* 1: the 2 L weight gradients as TWO ptpp_conv1d_wgrad_batched calls after the loop (all
                             * operands are slabs that outlive it): no split-K partials, bit-reproducible; 0: one
                             * ptpp_conv1d_wgrad per layer inside the loop, exactly as the per-launch path */
  const void* const* dil_wst;  /* [L] pack mode 4 operands of the dilated convs or NULL: their data gradients on the row-tile kernel */
  const void* const* out_wst;  /* [L] pack mode 4 operands of the output projections or NULL: the fused gate backward on the row-tile
                                * engine (ptpp_conv1d_rt_gate_bwd) where it is supported, bit-identically */
  float* colpart;           /* round 6: (L, B, ceil(T / 32), C) f32 scratch or NULL.  With it (and dil_wst, C = 256, where
                             * ptpp_conv1d_rt_colpart_supported) S comes from the data-gradient convs' epilogues
                             * (ptpp_conv1d_rt_fwd_cs) instead of a pass over gx_all: same sums, another (fixed) order */
} ptpp_diffnet_stack_bwd_args;
int ptpp_diffnet_stack_bwd(const ptpp_diffnet_stack_bwd_args* a, void* stream);

/* A run of post-LN Transformer encoder layers, forward only (the frozen layers of the prompt encoder's BERT: the reference
 * trains only encoder.layer[-1].attention, modules/prompt_encoder.py:25-38; per layer, transformers BertLayer =
 * fused q|k|v projection -> plain attention (probability dropout) -> output projection (+ hidden dropout + residual) ->
 * LayerNorm -> intermediate (exact GELU) -> output (+ dropout + residual) -> LayerNorm).  h (B, T, C) is read from h_in;
 * the result of the last layer is in h_out.  Tables are HOST arrays of L device pointers; seeds: HOST array of 3 L dropout
 * seeds (attention, attention-output, output) -- ignored where the matching p is 0.  scratch: >= B*T*(7C + F) elements of
 * dtype (F = intermediate width); ws: the split-K scratch of ptpp_conv1d_fwd_ws (used exactly where the per-launch path
 * hands it over: T <= 512 and K >= 2048). */
typedef struct {
  const void* h_in;
  void* h_out;
  const int32_t* lengths;      /* (B) keys at positions >= length are masked */
  const void* const* qkv_wp;   /* [L] packed (3C, 1, C) */
  const float* const* qkv_b;   /* [L] (3C) */
  const void* const* ao_wp;    /* [L] packed (C, 1, C): attention output projection */
  const float* const* ao_b;
  const float* const* ln1_g;
  const float* const* ln1_b;
  const void* const* i_wp;     /* [L] packed (F, 1, C) */
  const float* const* i_b;
  const void* const* o_wp;     /* [L] packed (C, 1, F) */
  const float* const* o_b;
  const float* const* ln2_g;
  const float* const* ln2_b;
  const uint64_t* seeds;       /* [3 L] */
  void* scratch; size_t scratch_bytes;
  void* ws; size_t ws_bytes;
  float eps, p_att, p_hid;
  int32_t B, T, C, F, H, L, dtype;
} ptpp_encoder_layers_fwd_args;
int ptpp_encoder_layers_fwd(const ptpp_encoder_layers_fwd_args* a, void* stream);

/* A stack of n [Conv1d (C -> C, kernel ks, "same" padding) -> LayerNorm] layers: the variance predictors
 * (modules/variance_adaptor.py:23-62: dropout(LN(relu(conv(x)))) * mask) and the frame prior network
 * (modules/frame_prior.py:76-89: x = LN(x + dropout(gelu(conv(x * mask)))), masked after the last layer).
 *   z_i = act(conv_i(in_mask ? x_i masked : x_i) + b_i)                          conv_act: PTPP_ACT_NONE | PTPP_ACT_RELU
 *   x_{i+1} = drop_out(LN(drop_in(act_in(z_i)) + (ln_res ? x_i : 0)) * g_i + b_i) * [mask]
 * out_mask: 0 none, 1 every layer, 2 the last layer only.  conv_mask: the conv reads lengths and masks its input rows.
 * Everything the backward needs is kept in slabs: x_all[i] = x_{i+1}, z_all[i], sum_all[i] (the LayerNorm's input; NULL
 * when ln_res == act_in == drop_in == 0), mean_all / rstd_all (n, B*T).  seeds: HOST [2n] = (drop_in, drop_out) per layer. */
typedef struct {
  const void* x0;
  const int32_t* lengths;
  const void* const* wp;      /* [n] packed (C, ks, C) forward operands */
  const float* const* bias;   /* [n] (C) */
  const float* const* gamma;  /* [n] (C) */
  const float* const* beta;
  void* x_all; void* z_all; void* sum_all;
  float* mean_all; float* rstd_all;
  const uint64_t* seeds;
  void* ws; size_t ws_bytes;  /* split-K scratch of ptpp_conv1d_fwd_ws (handed over where the per-launch path does) */
  float eps, drop_in, drop_out;
  int32_t B, T, C, n, ks, conv_act, conv_mask, ln_res, act_in, out_mask, dtype;
  const void* const* wstream; /* [n] the same weights in pack mode 3, or NULL: with them the convs of a frame-level stack
                               * (B * T >= 24576, ptpp_conv1d_rt_supported) run on the row-tile kernel, bit-identically */
} ptpp_conv_ln_stack_fwd_args;
int ptpp_conv_ln_stack_fwd(const ptpp_conv_ln_stack_fwd_args* a, void* stream);

/* Backward of the stack.  gy: gradient w.r.t. x_n.  gx (nullable): gradient w.r.t. x0.  Weight / bias / gamma / beta
 * gradients ACCUMULATE into the f32 targets; the conv weight gradients run on side_stream when it is not NULL (per layer
 * inside the loop, or -- batched_wgrad -- as ONE ptpp_conv1d_wgrad_batched call after it: gz_all keeps every layer's
 * conv-output gradient).  Scratch: gz_all (n, B, T, C); tmp: 4 (B, T, C) tensors; red_scratch: the zero-filled reduction
 * scratch of ptpp_layernorm_bwd. */
typedef struct {
  const void* gy;
  const void* x0;
  const void* x_all; const void* z_all; const void* sum_all;
  const float* mean_all; const float* rstd_all;
  const int32_t* lengths;
  const void* const* wpt;     /* [n] mode-1 packed operands */
  const float* const* gamma;
  float* const* dw; float* const* db; float* const* dgamma; float* const* dbeta;
  void* gz_all;
  void* tmp;
  void* gx;
  const uint64_t* seeds;
  void* red_scratch; size_t red_bytes;
  void* ws_main; size_t ws_main_bytes;
  void* ws_side; size_t ws_side_bytes;
  void* side_stream;
  float drop_in, drop_out;
  int32_t B, T, C, n, ks, conv_act, conv_mask, ln_res, act_in, out_mask, dtype, batched_wgrad;
  const void* const* wstream_t; /* [n] pack mode 4 operands or NULL: the data gradients on the row-tile kernel */
} ptpp_conv_ln_stack_bwd_args;
int ptpp_conv_ln_stack_bwd(const ptpp_conv_ln_stack_bwd_args* a, void* stream);

/* y (dtype, n elements, n % 4 == 0) = x (f32), round to nearest even */
int ptpp_cast_from_f32(const float* x, void* y, int64_t n, int dtype, void* stream);

/* One Conformer encoder block (reference modules/esp/conformer/encoder_layer.py:74-162 with the configuration of
 * prompttts_mdn_v2_wo_erg_final.yaml: macaron conv1d feed-forward pair, relative-position self-attention, convolution module,
 * normalize_before, final LayerNorm), forward and hand-derived backward, as TWO calls per block.  The launches, their order,
 * arguments and dropout seeds are those of the per-launch path (promptttspp_amd/modules/esp/__init__.py::EncoderLayer.cl and
 * the autograd Functions behind it): results are bit-identical to it.
 *   x1 = x  + 0.5 drop(ffn_macaron(LN0(x)))        ffn(v) = mask w2(drop'(mask relu(w1(mask v))))   (k = 9 convs)
 *   x2 = x1 + drop(mask out(attn(qkv(LN1(x1)), pos_w(pos_emb), u, v)))
 *   x3 = x2 + drop(mask pw2(swish(BN(dwconv(glu(mask pw1(LN2(x2))))))))
 *   x4 = x3 + 0.5 drop(ffn(LN3(x3)));   y = mask LN4(x4)
 * Weights: packed operands of the compute dtype (ptpp_pack_conv_weight; qkv = the three projections concatenated), biases /
 * norm parameters / pos_bias_u, v / depthwise taps (C, ks) in f32.  slab: everything the backward re-reads, laid out by
 * the library (ptpp_conformer_block_slab_bytes); scratch (backward): ptpp_conformer_block_bwd_scratch_bytes. */
typedef struct {
  const float* ln_g[5]; const float* ln_b[5];   /* 0 ff_macaron, 1 mha, 2 conv, 3 ff, 4 final */
  const void* ffm_w1; const float* ffm_b1; const void* ffm_w2; const float* ffm_b2;
  const void* ff_w1; const float* ff_b1; const void* ff_w2; const float* ff_b2;
  const void* qkv_w; const float* qkv_b; const void* pos_w; const void* out_w; const float* out_b;
  const float* bias_u; const float* bias_v;
  const void* pw1_w; const float* pw1_b; const void* pw2_w; const float* pw2_b;
  const float* dw_w; const float* dw_b; const float* bn_g; const float* bn_b;
  float* bn_rmean; float* bn_rvar;              /* running estimates, updated when bn_train */
  const float* bn_mean_in; const float* bn_rstd_in;  /* statistics to use when !bn_train */
} ptpp_conformer_weights;

typedef struct {
  const void* x; void* y;                       /* (B, T, C) in / out */
  const void* pos_emb;                          /* (L, C) relative-position table: L = 2T-1 (new) or T (legacy) */
  const int32_t* lengths;
  ptpp_conformer_weights w;
  void* slab; size_t slab_bytes;
  void* ws; size_t ws_bytes;                    /* split-K scratch of ptpp_conv1d_fwd_ws */
  void* red_scratch; size_t red_bytes;          /* reduction scratch (BatchNorm statistics) */
  const uint64_t* seeds;                        /* HOST [6]: ffm w1, ffm w2, attention out, pw2, ff w1, ff w2 */
  float p_ffn, p_drop, bn_momentum, bn_eps;
  int32_t B, T, C, F, H, L, ks_ffn, ks_dw, variant, bn_train, save, dtype;
  const void* ffn_ws[4];                        /* round 6: ffm_w1, ffm_w2, ff_w1, ff_w2 as operand streams (pack mode 3) or NULL:
                                                   those convs on ptpp_conv1d_rt_fwd_ex (all four or none) */
} ptpp_conformer_block_fwd_args;
size_t ptpp_conformer_block_slab_bytes(int B, int T, int C, int F, int H, int L, int dtype);
int ptpp_conformer_block_fwd(const ptpp_conformer_block_fwd_args* a, void* stream);

typedef struct {            /* accumulation targets (f32), one per parameter tensor */
  float* ln_g[5]; float* ln_b[5];
  float* ffm_w1; float* ffm_b1; float* ffm_w2; float* ffm_b2;
  float* ff_w1; float* ff_b1; float* ff_w2; float* ff_b2;
  float* q_w; float* q_b; float* k_w; float* k_b; float* v_w; float* v_b;
  float* pos_w; float* out_w; float* out_b; float* bias_u; float* bias_v;
  float* pw1_w; float* pw1_b; float* pw2_w; float* pw2_b;
  float* dw_w; float* dw_b;  /* ZERO-FILLED (C, ks) / (C): the depthwise kernel accumulates with atomics */
  float* bn_sums;            /* (2C) overwritten: [dbeta | dgamma] */
} ptpp_conformer_grads;

typedef struct {
  const void* gy; void* gx;                     /* gradient w.r.t. y in, w.r.t. x out */
  const void* x; const void* pos_emb; const int32_t* lengths;
  ptpp_conformer_weights w;                     /* forward operands (norm parameters, u, v, depthwise taps) */
  const void* ffm_w1t; const void* ffm_w2t; const void* ff_w1t; const void* ff_w2t;   /* mode-1 packed operands */
  const void* qkv_wt; const void* out_wt; const void* pw1_wt; const void* pw2_wt;
  ptpp_conformer_grads g;
  const void* slab; void* scratch; size_t scratch_bytes;
  void* ws_main; size_t ws_main_bytes; void* ws_side; size_t ws_side_bytes;
  void* red_scratch; size_t red_bytes;
  void* side_stream;                            /* weight gradients fork onto it when not NULL */
  const uint64_t* seeds;
  float p_ffn, p_drop;
  int32_t B, T, C, F, H, L, ks_ffn, ks_dw, variant, bn_train, dtype;
  const void* ffn_wts[4];                       /* round 6: ffm_w1t, ffm_w2t, ff_w1t, ff_w2t as operand streams (pack mode 4) or NULL */
  void* side_stream2;                           /* round 6: with side_stream, a SECOND stream that takes the depthwise and the 1 x 1
                                                   weight gradients while side_stream runs the k = 9 ones; or NULL.  The caller
                                                   joins both before it reads the gradients. */
  float* bn_dgamma; float* bn_dbeta;            /* round 6: accumulation targets of the BatchNorm parameter gradients or NULL (then
                                                   the caller adds g.bn_sums itself) */
} ptpp_conformer_block_bwd_args;
size_t ptpp_conformer_block_bwd_scratch_bytes(int B, int T, int C, int F, int H, int L, int dtype);
int ptpp_conformer_block_bwd(const ptpp_conformer_block_bwd_args* a, void* stream);

/* The convolution stack of the GST reference encoder, training mode (modules/reference_encoder.py:61-82, 100-106):
 * nlayer x [Conv2d 3x3 stride 2 pad 1, no bias -> BatchNorm2d (batch statistics, running estimates updated) -> ReLU] on a
 * channels-last spectrogram.  x: (B, H, W) with ONE channel; layer i maps (B, H_i, W_i, cin_i) -> (B, ceil(H_i/2),
 * ceil(W_i/2), cout_i) as [weight pack, im2col gather, GEMM with K = 9 cin_i (8 for the first layer's channel), batch
 * statistics, normalise + ReLU]: 5 launches.  All tables are HOST arrays of nlayer entries; w: the f32 nn.Conv2d weights;
 * wp_fwd / wp_bwd: device buffers for the packed operands (cout_i x Kp_i and 9 cin_i x coutp_i elements of dtype; wp_bwd[0]
 * unused).  The slab keeps what the backward reads (col_i, pre-BN z_i, mean / rstd) -- ptpp_refenc_convs_slab_bytes. */
typedef struct {
  const void* x;
  void* y;                      /* (B, H_n, W_n, cout_last) */
  const int32_t* cout;          /* HOST [nlayer] */
  const float* const* w;
  void* const* wp_fwd;
  void* const* wp_bwd;
  const float* const* bn_g;
  const float* const* bn_b;
  float* const* bn_rmean;       /* entries nullable: no running estimates */
  float* const* bn_rvar;
  void* slab;
  size_t slab_bytes;
  void* ws;                     /* split-K scratch of ptpp_conv1d_fwd_ws */
  size_t ws_bytes;
  void* red_scratch;            /* PTPP_RED_SCRATCH_BYTES(max cout), zero */
  size_t red_bytes;
  float bn_momentum, bn_eps;
  int32_t B, H, W, nlayer, dtype;
} ptpp_refenc_convs_fwd_args;
size_t ptpp_refenc_convs_slab_bytes(int B, int H, int W, int nlayer, const int32_t* cout, int dtype);
int ptpp_refenc_convs_fwd(const ptpp_refenc_convs_fwd_args* a, void* stream);

/* Backward of the same stack from gy (B, H_n, W_n, cout_last): per layer [BatchNorm + ReLU backward, weight gradient,
 * data gradient GEMM, col2im] (the first layer has no data gradient).  dwg[i]: f32 (cout_i, 9 cinq_i) matrices in the
 * GEMM's K order, ACCUMULATED into (the caller permutes them into the nn.Conv2d layout); bn_sums[i]: 2 cout_i f32,
 * overwritten with [dbeta | dgamma].  scratch: ptpp_refenc_convs_bwd_scratch_bytes. */
typedef struct {
  const void* gy;
  const int32_t* cout;
  void* const* wp_bwd;
  const float* const* bn_g;
  const float* const* bn_b;
  float* const* dwg;
  float* const* bn_sums;
  const void* slab;
  void* scratch;
  size_t scratch_bytes;
  void* ws;                     /* workspace of ptpp_conv1d_wgrad / split-K scratch of the data gradient */
  size_t ws_bytes;
  void* red_scratch;
  size_t red_bytes;
  int32_t B, H, W, nlayer, dtype;
} ptpp_refenc_convs_bwd_args;
size_t ptpp_refenc_convs_bwd_scratch_bytes(int B, int H, int W, int nlayer, const int32_t* cout, int dtype);
int ptpp_refenc_convs_bwd(const ptpp_refenc_convs_bwd_args* a, void* stream);

/* ------------------------------------------------------------------ *
 * Deferred reduction (round 6).  The finishing launch of a PARAMETER-gradient column sum (ptpp_layernorm_bwd's dgamma / dbeta,
 * ptpp_attention_bwd's du / dvb, ptpp_scalar_embed_bwd) has no reader before the optimiser or the gradient exchange.  After
 * ptpp_red_defer(arena, bytes) -- `arena`: ZEROED device memory, >= 4 MiB, owned by the caller, one device per process -- such a
 * call ignores its `scratch` argument, takes a private slice of the arena (one sub-arena per stream) and queues its finishing
 * step; ptpp_red_flush(stream) finishes everything queued with one launch per producing stream (on that stream) and makes
 * `stream` wait for them.  The destinations are complete only after the flush: call it before the optimiser step and before a
 * gradient bucket is exchanged.  ptpp_red_defer_suspend(+1 / -1) brackets calls whose destination IS read right away (they
 * finish immediately, as without deferral); a full arena falls back the same way.  ptpp_red_defer(NULL, 0) turns deferral off
 * (nothing may be queued).  ptpp_red_pending(): queued sums (diagnostics).
 * ------------------------------------------------------------------ */
int ptpp_red_defer(void* arena, size_t bytes);
int ptpp_red_defer_suspend(int delta);
int ptpp_red_pending(void);
int ptpp_red_flush(void* stream);

/* ------------------------------------------------------------------ *
 * Training-step glue (round 6, csrc/glue.hip): chains of small tensor ops around the hot kernels as one launch each.
 * ------------------------------------------------------------------ */
/* Every loss of PromptTTSMDNDurCFG.forward (models/prompttts_mdn_v2_final/model.py:126-183) in ONE launch each way:
 *   dec   = sum_{valid frames} |noise - pred| / n_frames / dec_scale          (:138-157, F.l1_loss of the masked tensors)
 *   cf0   = sum |pv[..., 0] - cf0_tgt| / n_frames,  vuv = sum |pv[..., 1] - vuv_tgt| / n_frames        (:168-170; no mask)
 *   dur   = mean over valid phones of the mixture NLL of to_log_scale(dur)    (:151-154; modules/mdn.py:81-175, D = 1)
 *   style = mean over (B, D) of the dimension-wise mixture NLL of sty_tgt     (:159-163)
 *   total = dec + dur + cf0 + vuv + style
 * y_dur (B, Tp, 3 G) / y_sty (B, 3 G D) are the RAW outputs of the MDN heads, [pi logits | log_sigma | mu] with (G, D) order
 * inside each part; the log-softmax over the G components (mdn.py:56-60) happens here.  pred (B, Tf, M) and pv (B, Tf, 2) in
 * `dtype` (f32 / bf16), everything else f32; flen / plen int32 (B).  Outputs: total[0]; comps[0..4] = dec, dur, cf0, vuv, style,
 * comps[5..6] = n_frames, n_phones; nll_dur (B, Tp) and nll_sty (B, D) are kept for the backward.  scratch:
 * ptpp_tts_losses_scratch_bytes() bytes, zero before the first call (left zero).  Partial sums are added in a fixed order.
 * _bwd: g_total / g_comps (5) device scalars (either may be NULL = 0) -> dpred, dpv (`dtype`), dy_dur, dy_sty (f32), every
 * element written. */
typedef struct {
  const void* pred;
  const float* noise;
  const int32_t* flen;
  const void* pv;
  const float* cf0_tgt;
  const float* vuv_tgt;
  const float* y_dur;
  const float* dur;
  const int32_t* plen;
  const float* y_sty;
  const float* sty_tgt;
  float* total;
  float* comps;
  float* nll_dur;
  float* nll_sty;
  void* scratch;
  int32_t B, Tf, Tp, M, G_dur, G_sty, D_sty, dtype;
  float dec_scale, lp_min, ls_min;
} ptpp_tts_loss_args;
int64_t ptpp_tts_losses_scratch_bytes(void);
int ptpp_tts_losses_fwd(const ptpp_tts_loss_args* a, void* stream);
int ptpp_tts_losses_bwd(const ptpp_tts_loss_args* a, const float* g_total, const float* g_comps, void* dpred, void* dpv,
                        float* dy_dur, float* dy_sty, void* stream);

/* DDPM q_sample on the dataset's mel layout (modules/diffusion.py:97-101,110-115,304-313):
 *   out[b, t, m] = sqrt_ac[step_b] * norm(mel[b, m, t]) + sqrt_1mac[step_b] * noise[b, t, m]   (unfused f32 sequence, then `dtype`)
 * norm(x) = x / norm_scale (use_scale) or (x - a_min) / (a_max - a_min) * 2 - 1.  mel (B, M, T) f32, noise (B, T, M) f32, M <= 128. */
int ptpp_q_sample_bct(const float* mel, const float* noise, const int64_t* step, const float* sqrt_ac, const float* sqrt_1mac,
                      int K, float norm_scale, float a_min, float a_max, int use_scale, void* out, int B, int M, int T,
                      int dtype, void* stream);

/* SinusoidalPosEmb (modules/denoiser.py:29-41): out[b] = [sin(e) | cos(e)], e = float(scale * step_b) * exp(k * neg_log_rate),
 * k < half, neg_log_rate = -(ln 10000 / (half - 1)) as f32; out (B, 2 half) f32.  Mish (:23-26) forward / backward on f32. */
int ptpp_step_sinusoid(const int64_t* step, int64_t scale, float neg_log_rate, int B, int half, float* out, void* stream);
int ptpp_mish_fwd(const float* x, float* y, int64_t n, void* stream);
int ptpp_mish_bwd(const float* x, const float* gy, float* gx, int64_t n, void* stream);

/* PhonemeEmbedding (layers/embedding.py:21-36), channels-last: out[b, t] = table[ids[b, t]] (* scale) for t < lengths[b], else 0.
 * _bwd ADDS scale * (sum over the valid rows of an id of dout) into dtable, one block per vocabulary entry walking the rows in
 * order (bit-reproducible); row `padding_idx` gets nothing.  ids int64, table / dtable (V, C) f32, out / dout `dtype`. */
int ptpp_embed_cl_fwd(const int64_t* ids, const float* table, const int32_t* lengths, float scale, int do_scale, void* out,
                      int B, int T, int C, int V, int dtype, void* stream);
int ptpp_embed_cl_bwd(const int64_t* ids, const void* dout, const int32_t* lengths, float scale, int do_scale, float* dtable,
                      int B, int T, int C, int V, int padding_idx, int dtype, void* stream);

/* x + Conv1d(1 -> C, k = 1)(track) * mask (modules/variance_adaptor.py:139-146, pitch / energy embedding):
 *   out[b, t, :] = x[b, t, :] + T(track[b, t] * w + bias)  for t < lengths[b], else x[b, t, :].
 * _bwd: dw += sum_valid track * dout, db += sum_valid dout (C in {256 .. 1024}, through the reduction scratch; deferrable). */
int ptpp_scalar_embed_add(const void* x, const float* track, const float* w, const float* bias, const int32_t* lengths,
                          void* out, int B, int T, int C, int dtype, void* stream);
int ptpp_scalar_embed_bwd(const void* dout, const float* track, const int32_t* lengths, float* dw, float* db, int B, int T,
                          int C, int dtype, void* scratch, size_t scratch_bytes, void* stream);

/* Linear / 1x1 Conv1d with Cout in 1..4 (the pitch / V-UV head, modules/variance_adaptor.py:52-62): y[b, t, o] = x[b, t] . w[o] + bias[o]
 * for t < lengths[b] (lengths NULL: every row), else 0; x (B, T, Cin) `dtype`, Cin in {256, 512, 768, 1024}, w (Cout, Cin) / bias
 * f32, f32 accumulation.  _bwd: dx (nullable) = dy . w on valid rows (0 elsewhere); dw += sum_valid dy^T x, db += sum_valid dy
 * (through the reduction scratch: PTPP_RED_SCRATCH_BYTES(Cout * Cin + Cout) / 2 bytes; deferrable). */
int ptpp_linear_small_fwd(const void* x, const float* w, const float* bias, const int32_t* lengths, void* y, int B, int T,
                          int Cin, int Cout, int dtype, void* stream);
int ptpp_linear_small_bwd(const void* x, const void* dy, const float* w, const int32_t* lengths, void* dx, float* dw, float* db,
                          int B, int T, int Cin, int Cout, int dtype, void* scratch, size_t scratch_bytes, void* stream);

/* F.normalize over the channels of (rows, C) f32 (model.py:108: style embeddings): y = x / max(||x||, eps); nrm (rows) kept. */
int ptpp_l2norm_fwd(const float* x, float* y, float* nrm, int rows, int C, float eps, void* stream);
int ptpp_l2norm_bwd(const float* y, const float* nrm, const float* gy, float* gx, int rows, int C, float eps, void* stream);

/* Running frame count per phone for ptpp_length_regulate_* (utils/model.py:37-47): cum[b, p] = min(sum_{q <= p} dur[b, q],
 * 2^31 - 1); dur (B, Tp) f32 (integer valued; is_float) or int64. */
int ptpp_durations_cumsum(const void* dur, int is_float, int32_t* cum, int B, int Tp, void* stream);

/* y[b, t, :] = x[b, t, :] + T(e[b, :]) for EVERY row (model.py:111: the style embedding is added to padded phones too);
 * ptpp_rows_sum: de[b, :] = sum_t dy[b, t, :] (f32, fixed order). */
int ptpp_bcast_add_rows(const void* x, const float* e, void* y, int B, int T, int C, int dtype, void* stream);
int ptpp_rows_sum(const void* dy, float* de, int B, int T, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ *
 * Data-parallel gradient exchange over RCCL / xGMI (reference: DistributedDataParallel set up in
 * trainers/tts.py:52-55 (init_process_group("nccl")) and :117 (DDP(model, device_ids=[rank])), whose
 * bucketed all-reduce averages the gradients over the ranks during backward).
 * One process per GPU.  librccl is bound at run time (the instance the process already holds, else ROCm's);
 * without it these return PTPP_ENOTSUP.  Rank 0 obtains the 128-byte id and hands it to the other ranks
 * through any side channel (the trainer uses the torch.distributed store); every rank then calls
 * ptpp_comm_init with the same id.  `comm` is an opaque ncclComm_t.  The collectives are asynchronous on
 * `stream`, in place, and must be issued in the same order on every rank.
 * ------------------------------------------------------------------ */
#define PTPP_COMM_ID_BYTES 128
int ptpp_comm_unique_id(void* id_out /* HOST, PTPP_COMM_ID_BYTES */);
int ptpp_comm_init(int rank, int world, const void* unique_id /* HOST */, void** comm_out /* HOST */);
int ptpp_comm_destroy(void* comm);
/* buf[i] <- mean over ranks of buf[i]   (n elements of dtype, in place) */
int ptpp_allreduce_mean(void* buf, int64_t n, int dtype, void* comm, void* stream);
/* buf <- root's buf  (initial parameter broadcast, DDP's constructor) */
int ptpp_broadcast(void* buf, int64_t n, int dtype, int root, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTPP_H_ */
