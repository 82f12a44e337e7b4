// Whole-unit drivers of the C ABI (include/ptpp.h "Whole-unit drivers"): one call issues every launch of a unit.
// Host code only: each driver is a sequence of the library's own entry points, in the order the per-launch path
// (promptttspp_amd/functional.py) issues them, so the results are bit-identical to it.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include "../../include/ptpp.h"

void ptpp_set_error(const char* fmt, ...);

#define ST_CHECK_ARG(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      ptpp_set_error(__VA_ARGS__);   \
      return PTPP_EINVAL;            \
    }                                \
  } while (0)
#define ST_TRY(call)                 \
  do {                               \
    const int rc_ = (call);          \
    if (rc_ != PTPP_OK) return rc_;  \
  } while (0)

namespace {

inline size_t esize(int dtype) { return dtype == PTPP_F32 ? 4 : 2; }
inline char* at(void* base, size_t elems, int dtype) { return static_cast<char*>(base) + elems * esize(dtype); }
inline const char* at(const void* base, size_t elems, int dtype) { return static_cast<const char*>(base) + elems * esize(dtype); }

ptpp_conv1d_args conv_args(const void* x, int ldx, const void* wp, const float* bias, const void* res, int ldr, void* y, int ldy,
                           const int32_t* lengths, int B, int T, int cin, int cout, int ks, int dil, int pad, int act, int in_mask,
                           int out_mask, int dtype) {
  ptpp_conv1d_args c;
  memset(&c, 0, sizeof(c));
  c.x = x; c.wp = wp; c.bias = bias; c.res = res; c.y = y; c.lengths = lengths;
  c.B = B; c.T = T; c.Cin = cin; c.Cout = cout; c.ks = ks; c.dil = dil; c.pad = pad;
  c.ldx = ldx; c.ldy = ldy; c.ldr = ldr;
  c.act = act; c.in_mask = in_mask; c.out_mask = out_mask; c.out_scale = 1.0f; c.dtype = dtype;
  return c;
}

}  // namespace

// A conv launch of a driver: on the row-tile kernel (conv1d_rt.hip) when its operand stream was handed over and the launch is
// frame-level -- the SAME rule as promptttspp_amd/ops.py::conv1d_rt_ok, so that both paths take the same kernel -- else as before
// The threshold comes from the Python layer (ptpp_conv_rt_set_min_rows, pushed by ops.conv_rt_min_rows whenever its value
// changes) so that both layers decide alike; a C-only caller gets the environment variable / the default.
static long long g_rt_min_rows = -1;
extern "C" int ptpp_conv_rt_set_min_rows(long long v) {
  g_rt_min_rows = v;
  return PTPP_OK;
}
static int64_t rt_min_rows() {
  if (g_rt_min_rows >= 0) return g_rt_min_rows;
  int64_t min_rows = 24576;
  if (const char* e = getenv("PTPP_CONV_RT_MIN_ROWS")) min_rows = atoll(e);
  return min_rows;
}
static bool rt_takes(const ptpp_conv1d_args& c, const void* wstream) {
  return wstream && (int64_t)c.B * c.T >= rt_min_rows() && ptpp_conv1d_rt_supported(c.Cin, c.Cout, c.ks, c.dil, c.act, c.dtype);
}
// 1: the row-tile kernel takes the launch; 0: no operand stream was handed over (the tile kernel reads wp); -1: a stream WAS
// handed over but this side would not take the row-tile kernel.  The caller packs ONLY the stream form in that case (and may
// alias wp to it), so falling back to wp would read a stream image as a [Cout][ks][Cin] operand: refuse instead.
static int rt_decide(const ptpp_conv1d_args& c, const void* wstream, const char* who) {
  if (!wstream) return 0;
  if (rt_takes(c, wstream)) return 1;
  ptpp_set_error("%s: an operand stream (pack mode 3 / 4) was passed for a launch the row-tile kernel does not take here (B*T=%lld, "
                 "Cin=%d, Cout=%d, ks=%d, dil=%d; PTPP_CONV_RT_MIN_ROWS must read the same in both layers)",
                 who, (long long)c.B * c.T, c.Cin, c.Cout, c.ks, c.dil);
  return -1;
}
#define ST_RT(var, c, ws, who)                 \
  const int var = rt_decide(c, ws, who);       \
  if (var < 0) return PTPP_EINVAL

extern "C" int ptpp_diffnet_stack_fwd(const ptpp_diffnet_stack_fwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->h0 && (a->cond_all || a->condx) && a->dsteps && a->skip && a->dil_wp && a->dil_b && a->out_wp && a->out_b && a->yin_all &&
                   a->g_all && a->x_buf[0] && a->x_buf[1],
               "diffnet_stack_fwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->L > 0 && a->cycle > 0 && a->n_slabs >= 2, "diffnet_stack_fwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16 || a->dtype == PTPP_F16, "diffnet_stack_fwd: bad dtype %d", a->dtype);
  ST_CHECK_ARG(a->dtype != PTPP_F16 || (a->fused_gate == 1 && a->wstream && !a->condx && !a->lengths),
               "diffnet_stack_fwd: f16 is the sampler's form (fused gate without saving, operand stream, no masks)");
  // fused_gate: 0 = conv + gate_fwd, 1 = gate in the conv's epilogue, pre-activation not kept (inference), 2 = gate in the
  // epilogue AND the pre-activation kept in a_all (training: ptpp_conv1d_gate_fwd_save)
  ST_CHECK_ARG(a->fused_gate == 1 || a->a_all, "diffnet_stack_fwd: a_all is needed unless the gate is fused without saving");
  ST_CHECK_ARG(a->fused_gate != 1 || ((a->dtype == PTPP_BF16 || a->dtype == PTPP_F16) && !a->lengths),
               "diffnet_stack_fwd: the fused gate is the 16-bit inference path");
  ST_CHECK_ARG(a->fused_gate != 2 || ptpp_conv1d_gate_fwd_save_supported(a->C, a->C, a->dtype),
               "diffnet_stack_fwd: the fused gate with the kept pre-activation needs bf16 and C %% 64 == 0");
  const int B = a->B, T = a->T, C = a->C, L = a->L, dt = a->dtype;
  const size_t BTC = (size_t)B * T * C;
  const int ldc = L * 2 * C;
  const bool fuse_post = ptpp_conv1d_diffnet_post_supported(C, C, dt) != 0;
  ST_CHECK_ARG(fuse_post || a->o_buf, "diffnet_stack_fwd: o_buf is needed where the fused tail is unsupported");
  const int masked = a->lengths != nullptr;

  // yin_0 = h0 + dsteps[0]  (x = h0 itself)
  if (!a->yin0) ST_TRY(ptpp_diffnet_post_fwd(nullptr, a->h0, nullptr, a->dsteps, nullptr, at(a->yin_all, 0, dt), B, T, C, 1, dt, stream));
  const void* x = a->h0;
  // the whole layer as one launch (csrc/diffnet_layer.hip) where the fused gate is on and the operand stream was handed over
  const bool one_launch = a->wstream && a->fused_gate && ptpp_diffnet_layer_supported(C, dt) && (1 << ((L - 1) % a->cycle)) <= 8 && a->cycle <= 4;
  const int64_t wsb = one_launch ? (a->condx ? ptpp_diffnet_wstream_bytes_cond(C) : ptpp_diffnet_wstream_bytes(C)) : 0;
  ST_CHECK_ARG(!a->condx || one_launch, "diffnet_stack_fwd: condx needs the one-launch layer (wstream, fused gate, bf16, C = 256)");
  ST_CHECK_ARG(!a->skip_scaled || one_launch, "diffnet_stack_fwd: skip_scaled needs the one-launch layer (wstream, fused gate, bf16, C = 256)");
  for (int l = 0; l < L; ++l) {
    const int d = 1 << (l % a->cycle);
    const int slab = l % a->n_slabs;
    const void* yin = (l == 0 && a->yin0) ? a->yin0 : at(a->yin_all, slab * BTC, dt);
    void* g = at(a->g_all, slab * BTC, dt);
    const void* cond = a->cond_all ? at(a->cond_all, (size_t)l * 2 * C, dt) : nullptr;
    if (one_launch) {
      const float* dnext = l + 1 < L ? a->dsteps + (size_t)(l + 1) * B * C : nullptr;
      ptpp_diffnet_layer_args la;
      memset(&la, 0, sizeof(la));
      la.yin = yin; la.x = x; la.cond = cond; la.wstream = static_cast<const char*>(a->wstream) + (size_t)l * wsb;
      la.dil_b = a->dil_b[l]; la.out_b = a->out_b[l]; la.dnext = dnext; la.skip = a->skip;
      la.xn = a->x_buf[l & 1];
      la.yin_next = dnext ? at(a->yin_all, ((l + 1) % a->n_slabs) * BTC, dt) : nullptr;
      if (a->fused_gate == 2) { la.a_out = at(a->a_all, slab * 2 * BTC, dt); la.g_out = g; }
      la.lengths = a->lengths;
      if (l == L - 1 && a->skip_scaled) { la.skip_scaled = a->skip_scaled; la.skip_scale = a->skip_scale; }
      la.condx = a->condx; la.ldcx = a->ldcx;
      la.B = B; la.T = T; la.C = C; la.dil = d; la.ldc = ldc; la.init = l == 0; la.dtype = dt;
      ST_TRY(ptpp_diffnet_layer_fwd(&la, stream));
      x = la.xn;
      continue;
    }
    // (training, ragged batch: the dilated conv's output past an utterance's end only meets the masked output projection, so
    //  it is masked too -- a tile past the end then skips its K loop; a third of the row tiles of a token-bucket batch)
    if (a->fused_gate == 2) {
      ptpp_conv1d_args c = conv_args(yin, C, a->dil_wp[l], a->dil_b[l], cond, ldc, g, C, a->lengths, B, T, C, 2 * C, 3, d, d,
                                     PTPP_ACT_GATE, 0, masked, dt);
      ST_TRY(ptpp_conv1d_gate_fwd_save(&c, at(a->a_all, slab * 2 * BTC, dt), 2 * C, stream));
    } else if (a->fused_gate) {
      ptpp_conv1d_args c = conv_args(yin, C, a->dil_wp[l], a->dil_b[l], cond, ldc, g, C, nullptr, B, T, C, 2 * C, 3, d, d,
                                     PTPP_ACT_GATE, 0, 0, dt);
      ST_TRY(ptpp_conv1d_fwd(&c, stream));
    } else {
      void* act = at(a->a_all, slab * 2 * BTC, dt);
      ptpp_conv1d_args c = conv_args(yin, C, a->dil_wp[l], a->dil_b[l], cond, ldc, act, 2 * C, a->lengths, B, T, C, 2 * C, 3, d, d,
                                     PTPP_ACT_NONE, 0, masked, dt);
      ST_TRY(ptpp_conv1d_fwd(&c, stream));
      ST_TRY(ptpp_gate_fwd(act, g, (int64_t)B * T, C, dt, stream));
    }
    const float* dnext = l + 1 < L ? a->dsteps + (size_t)(l + 1) * B * C : nullptr;
    void* xn = a->x_buf[l & 1];
    void* yin_next = dnext ? at(a->yin_all, ((l + 1) % a->n_slabs) * BTC, dt) : nullptr;
    if (fuse_post) {
      ptpp_conv1d_args c = conv_args(g, C, a->out_wp[l], a->out_b[l], nullptr, 0, nullptr, C, a->lengths, B, T, C, 2 * C, 1, 1, 0,
                                     PTPP_ACT_NONE, 0, masked, dt);
      ST_TRY(ptpp_conv1d_diffnet_post(&c, x, a->skip, dnext, xn, yin_next, l == 0, stream));
    } else {
      ptpp_conv1d_args c = conv_args(g, C, a->out_wp[l], a->out_b[l], nullptr, 0, a->o_buf, 2 * C, a->lengths, B, T, C, 2 * C, 1, 1, 0,
                                     PTPP_ACT_NONE, 0, masked, dt);
      ST_TRY(ptpp_conv1d_fwd(&c, stream));
      ST_TRY(ptpp_diffnet_post_fwd(a->o_buf, x, a->skip, dnext, xn, yin_next, B, T, C, l == 0, dt, stream));
    }
    x = xn;
  }
  return PTPP_OK;
}

extern "C" int ptpp_diffnet_stack_bwd(const ptpp_diffnet_stack_bwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->gS && a->yin_all && a->a_all && a->g_all && a->dil_wpt && a->out_wpt && a->dw_dil && a->db_dil && a->dw_out &&
                   a->db_out && a->gx_all && a->do_all && a->dcond_all && a->S,
               "diffnet_stack_bwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->L > 0 && a->cycle > 0, "diffnet_stack_bwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "diffnet_stack_bwd: bad dtype %d", a->dtype);
  const int B = a->B, T = a->T, C = a->C, L = a->L, dt = a->dtype;
  const size_t BTC = (size_t)B * T * C;
  const int ldc = L * 2 * C;
  const bool fuse_gbwd = ptpp_conv1d_gate_bwd_supported(C, 2 * C, dt) != 0;
  ST_CHECK_ARG(fuse_gbwd || a->dg_buf, "diffnet_stack_bwd: dg_buf is needed where the fused gate backward is unsupported");
  void* wstream = a->side_stream ? a->side_stream : stream;
  void* ws_w = a->side_stream ? a->ws_side : a->ws_main;
  const size_t ws_w_bytes = a->side_stream ? a->ws_side_bytes : a->ws_main_bytes;
  const float r2 = (float)(1.0 / sqrt(2.0));
  const int bmask = a->lengths != nullptr;  // ragged batch: the gradients past an utterance's end are zero

  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(at(a->gx_all, (size_t)L * BTC, dt), 0, BTC * esize(dt), st) != hipSuccess) {
    ptpp_set_error("diffnet_stack_bwd: hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
    return PTPP_ELAUNCH;
  }
  // Where the row-tile kernel runs every dilated data gradient, its epilogue also writes the NEXT iteration's residual half of
  // dout (gx / sqrt2, masked) and one launch fills the skip halves of all layers: no per-layer pass over gx and gS.
  // the output projections' operand streams: the fused gate backward on the row-tile engine, at frame-level row counts
  const bool gbwd_rt = fuse_gbwd && a->out_wst != nullptr && ptpp_conv1d_rt_gate_bwd_supported(C, 2 * C, dt) != 0 &&
                       (int64_t)B * T >= rt_min_rows();
  bool fold = a->dil_wst != nullptr;
  for (int l = 0; l < L && fold; ++l) {
    const int d = 1 << (l % a->cycle);
    ptpp_conv1d_args c = conv_args(a->dcond_all, ldc, a->dil_wpt[l], nullptr, a->gx_all, C, a->gx_all, C, a->lengths, B, T, 2 * C, C, 3, d, d,
                                   PTPP_ACT_NONE, bmask, 0, dt);
    ST_RT(rt, c, a->dil_wst[l], "diffnet_stack_bwd");
    fold = rt == 1;
  }
  if (fold) ST_TRY(ptpp_diffnet_post_bwd_fill(a->gS, a->do_all, a->lengths, B, T, C, L, dt, stream));
  // Round 6: the per-layer column sums of gx (S) from the data-gradient convs' epilogues instead of a 307 MB pass over gx_all
  const int nslot = (T + 31) / 32;
  bool cs = fold && a->colpart != nullptr && C == 256;
  for (int l = 0; l < L && cs; ++l) cs = ptpp_conv1d_rt_colpart_supported(2 * C, 3, 1 << (l % a->cycle), B, T) != 0;
  for (int l = L - 1; l >= 0; --l) {
    const int d = 1 << (l % a->cycle);
    const void* gx = at(a->gx_all, (size_t)(l + 1) * BTC, dt);
    void* dout = at(a->do_all, (size_t)l * 2 * BTC, dt);
    const void* yin = at(a->yin_all, (size_t)l * BTC, dt);
    const void* act = at(a->a_all, (size_t)l * 2 * BTC, dt);
    const void* g = at(a->g_all, (size_t)l * BTC, dt);
    void* da = at(a->dcond_all, (size_t)l * 2 * C, dt);
    if (!fold) ST_TRY(ptpp_diffnet_post_bwd(gx, a->gS, dout, a->lengths, B, T, C, dt, stream));
    if (!a->batched_wgrad) {
      if (a->side_stream) ST_TRY(ptpp_stream_wait(a->side_stream, stream));
      ST_TRY(ptpp_conv1d_wgrad(g, dout, a->dw_out[l], a->db_out[l], nullptr, B, T, C, 2 * C, 1, 1, 0, C, 2 * C, 0, dt, ws_w, ws_w_bytes,
                               wstream));
    }
    if (fuse_gbwd) {
      // (do / da are zero past an utterance's end: the input mask changes nothing but lets those row tiles skip their K loops)
      ptpp_conv1d_args c = conv_args(dout, 2 * C, a->out_wpt[l], nullptr, nullptr, 0, nullptr, 0, a->lengths, B, T, 2 * C, C, 1, 1, 0,
                                     PTPP_ACT_NONE, bmask, 0, dt);
      if (gbwd_rt) ST_TRY(ptpp_conv1d_rt_gate_bwd(&c, a->out_wst[l], act, da, ldc, stream));
      else ST_TRY(ptpp_conv1d_gate_bwd(&c, act, da, ldc, stream));
    } else {
      ptpp_conv1d_args c = conv_args(dout, 2 * C, a->out_wpt[l], nullptr, nullptr, 0, a->dg_buf, C, a->lengths, B, T, 2 * C, C, 1, 1, 0,
                                     PTPP_ACT_NONE, bmask, 0, dt);
      ST_TRY(ptpp_conv1d_fwd(&c, stream));
      ST_TRY(ptpp_gate_bwd(act, a->dg_buf, da, (int64_t)B * T, C, ldc, dt, stream));
    }
    if (!a->batched_wgrad) {
      if (a->side_stream) ST_TRY(ptpp_stream_wait(a->side_stream, stream));
      ST_TRY(ptpp_conv1d_wgrad(yin, da, a->dw_dil[l], a->db_dil[l], nullptr, B, T, C, 2 * C, 3, d, d, C, ldc, 0, dt, ws_w, ws_w_bytes,
                               wstream));
    }
    ptpp_conv1d_args c = conv_args(da, ldc, a->dil_wpt[l], nullptr, gx, C, at(a->gx_all, (size_t)l * BTC, dt), C, a->lengths, B, T, 2 * C, C,
                                   3, d, d, PTPP_ACT_NONE, bmask, 0, dt);
    if (fold) ST_TRY(ptpp_conv1d_rt_fwd_cs(&c, a->dil_wst[l], r2, l > 0 ? at(a->do_all, (size_t)(l - 1) * 2 * BTC, dt) : nullptr, 2 * C, r2,
                                           cs ? a->colpart + (size_t)l * B * nslot * C : nullptr, stream));
    else ST_TRY(ptpp_conv1d_fwd_ex(&c, nullptr, 0, r2, 0.f, 0, stream));  // (fold is false only without operand streams: checked above)
  }
  if (a->batched_wgrad) {
    // every layer's (g, do) and (yin, da) pair is still in its slab: all output-projection gradients in one launch, all
    // dilated-conv gradients in another, each dw element summed over all rows by ONE block (no split-K partials)
    ST_CHECK_ARG(L <= 64, "diffnet_stack_bwd: at most 64 layers");
    ptpp_wgrad_problem po[64], pd[64];
    for (int l = 0; l < L; ++l) {
      po[l].x = at(a->g_all, (size_t)l * BTC, dt); po[l].dy = at(a->do_all, (size_t)l * 2 * BTC, dt);
      po[l].dw = a->dw_out[l]; po[l].dbias = a->db_out[l]; po[l].dil = 1; po[l].pad = 0;
      const int d = 1 << (l % a->cycle);
      pd[l].x = at(a->yin_all, (size_t)l * BTC, dt); pd[l].dy = at(a->dcond_all, (size_t)l * 2 * C, dt);
      pd[l].dw = a->dw_dil[l]; pd[l].dbias = a->db_dil[l]; pd[l].dil = d; pd[l].pad = d;
    }
    if (a->side_stream) ST_TRY(ptpp_stream_wait(a->side_stream, stream));
    ST_TRY(ptpp_conv1d_wgrad_batched(pd, L, nullptr, B, T, C, 2 * C, 3, C, ldc, 0, dt, ws_w, ws_w_bytes, wstream));
    ST_TRY(ptpp_conv1d_wgrad_batched(po, L, nullptr, B, T, C, 2 * C, 1, C, 2 * C, 0, dt, ws_w, ws_w_bytes, wstream));
  }
  if (cs) return ptpp_colsum_batch(a->colpart, a->S, L * B, nslot, C, PTPP_F32, stream);
  return ptpp_colsum_batch(a->gx_all, a->S, L * B, T, C, dt, stream);
}

// One conv / linear launch exactly as promptttspp_amd/ops.py::conv1d dispatches it: the split-K scratch is handed over
// only for few-row long-K shapes.
static int linear_like_ops(const ptpp_conv1d_args& c, float drop_p, uint64_t seed, void* ws, size_t ws_bytes, void* stream) {
  if (c.T <= 512 && c.ks * c.Cin >= 2048 && ws) return ptpp_conv1d_fwd_ws(&c, nullptr, 0, 1.0f, drop_p, seed, ws, ws_bytes, stream);
  return ptpp_conv1d_fwd_ex(&c, nullptr, 0, 1.0f, drop_p, seed, stream);
}

// A feed-forward conv of a Conformer block: on the row-tile engine (ptpp_conv1d_rt_fwd_ex) when its operand stream was handed
// over and the shape qualifies -- the rule of promptttspp_amd/ops.py::conv1d (conv1d_rt_ex_ok) -- else as every other launch
static int ffn_conv(const ptpp_conv1d_args& c, const void* wstream, float drop_p, uint64_t seed, void* ws, size_t ws_bytes, void* stream) {
  if (wstream && ptpp_conv1d_rt_ex_supported(c.Cin, c.Cout, c.ks, c.dil, c.act, c.dtype))
    return ptpp_conv1d_rt_fwd_ex(&c, wstream, 1.0f, drop_p, seed, ws, ws_bytes, stream);
  // (a caller that hands over streams packs ONLY those: without a usable stream the plain operand must be there)
  ST_CHECK_ARG(c.wp, "conformer_block: a feed-forward conv has neither a usable operand stream nor a packed operand");
  return linear_like_ops(c, drop_p, seed, ws, ws_bytes, stream);
}

extern "C" int ptpp_encoder_layers_fwd(const ptpp_encoder_layers_fwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->h_in && a->h_out && a->qkv_wp && a->qkv_b && a->ao_wp && a->ao_b && a->ln1_g && a->ln1_b && a->i_wp && a->i_b &&
                   a->o_wp && a->o_b && a->ln2_g && a->ln2_b && a->scratch && a->seeds,
               "encoder_layers_fwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->F > 0 && a->H > 0 && a->C % a->H == 0 && a->L > 0, "encoder_layers_fwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "encoder_layers_fwd: bad dtype %d", a->dtype);
  const int B = a->B, T = a->T, C = a->C, F = a->F, dt = a->dtype;
  const size_t R = (size_t)B * T;
  ST_CHECK_ARG(a->scratch_bytes >= R * (7 * (size_t)C + F) * esize(dt), "encoder_layers_fwd: scratch too small (%zu bytes, need %zu)",
               a->scratch_bytes, R * (7 * (size_t)C + F) * esize(dt));
  char* s0 = static_cast<char*>(a->scratch);
  void* qkv = s0;
  void* ctx = at(qkv, R * 3 * C, dt);
  void* att = at(ctx, R * C, dt);
  void* h1 = at(att, R * C, dt);
  void* hbuf = at(h1, R * C, dt);       // layer outputs alternate between hbuf and h_out so that the last lands in h_out
  void* inter = at(hbuf, R * C, dt);    // (R, F)
  const void* h = a->h_in;
  for (int l = 0; l < a->L; ++l) {
    void* hn = ((a->L - 1 - l) & 1) ? hbuf : a->h_out;
    const uint64_t* sd = a->seeds + 3 * l;
    ptpp_conv1d_args c = conv_args(h, C, a->qkv_wp[l], a->qkv_b[l], nullptr, 0, qkv, 3 * C, nullptr, B, T, C, 3 * C, 1, 1, 0, PTPP_ACT_NONE,
                                   0, 0, dt);
    ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
    ST_TRY(ptpp_attention_fwd(qkv, at(qkv, C, dt), at(qkv, 2 * C, dt), nullptr, nullptr, nullptr, ctx, nullptr, a->lengths, B, T, a->H,
                              C / a->H, 3 * C, 0, C, PTPP_ATTN_PLAIN, a->p_att, a->p_att > 0.f ? sd[0] : 0, dt, stream));
    c = conv_args(ctx, C, a->ao_wp[l], a->ao_b[l], h, C, att, C, nullptr, B, T, C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
    ST_TRY(linear_like_ops(c, a->p_hid, a->p_hid > 0.f ? sd[1] : 0, a->ws, a->ws_bytes, stream));
    ST_TRY(ptpp_layernorm_fwd(att, nullptr, a->ln1_g[l], a->ln1_b[l], h1, nullptr, nullptr, nullptr, nullptr, B, T, C, a->eps, 0,
                              PTPP_ACT_NONE, 0.f, 0, 0.f, 0, dt, stream));
    c = conv_args(h1, C, a->i_wp[l], a->i_b[l], nullptr, 0, inter, F, nullptr, B, T, C, F, 1, 1, 0, PTPP_ACT_GELU, 0, 0, dt);
    ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
    c = conv_args(inter, F, a->o_wp[l], a->o_b[l], h1, C, att, C, nullptr, B, T, F, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
    ST_TRY(linear_like_ops(c, a->p_hid, a->p_hid > 0.f ? sd[2] : 0, a->ws, a->ws_bytes, stream));
    ST_TRY(ptpp_layernorm_fwd(att, nullptr, a->ln2_g[l], a->ln2_b[l], hn, nullptr, nullptr, nullptr, nullptr, B, T, C, a->eps, 0,
                              PTPP_ACT_NONE, 0.f, 0, 0.f, 0, dt, stream));
    h = hn;
  }
  return PTPP_OK;
}

extern "C" int ptpp_conv_ln_stack_fwd(const ptpp_conv_ln_stack_fwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->x0 && a->wp && a->bias && a->gamma && a->beta && a->x_all && a->z_all && a->mean_all && a->rstd_all,
               "conv_ln_stack_fwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->n > 0 && a->ks > 0 && (a->ks & 1), "conv_ln_stack_fwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "conv_ln_stack_fwd: bad dtype %d", a->dtype);
  const bool fused_in = a->ln_res || a->act_in != PTPP_ACT_NONE || a->drop_in > 0.f;
  ST_CHECK_ARG(!fused_in || a->sum_all, "conv_ln_stack_fwd: sum_all is needed with a residual / activation / dropout before the norm");
  ST_CHECK_ARG((!a->conv_mask && a->out_mask == 0) || a->lengths, "conv_ln_stack_fwd: masks need lengths");
  ST_CHECK_ARG((a->drop_in <= 0.f && a->drop_out <= 0.f) || a->seeds, "conv_ln_stack_fwd: dropout needs seeds");
  const int B = a->B, T = a->T, C = a->C, n = a->n, dt = a->dtype;
  const size_t BTC = (size_t)B * T * C, R = (size_t)B * T;
  const void* x = a->x0;
  for (int i = 0; i < n; ++i) {
    void* z = at(a->z_all, i * BTC, dt);
    void* y = at(a->x_all, i * BTC, dt);
    ptpp_conv1d_args c = conv_args(x, C, a->wp[i], a->bias[i], nullptr, 0, z, C, a->conv_mask ? a->lengths : nullptr, B, T, C, C, a->ks, 1,
                                   a->ks / 2, a->conv_act, a->conv_mask ? 1 : 0, 0, dt);
    ST_RT(rt, c, a->wstream ? a->wstream[i] : nullptr, "conv_ln_stack_fwd");
    if (rt) ST_TRY(ptpp_conv1d_rt_fwd(&c, a->wstream[i], 1.0f, stream));
    else ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
    const int om = a->out_mask == 1 || (a->out_mask == 2 && i == n - 1);
    ST_TRY(ptpp_layernorm_fwd(z, a->ln_res ? x : nullptr, a->gamma[i], a->beta[i], y, fused_in ? at(a->sum_all, i * BTC, dt) : nullptr,
                              a->mean_all + i * R, a->rstd_all + i * R, a->lengths, B, T, C, a->eps, om, a->act_in, a->drop_in,
                              a->drop_in > 0.f ? a->seeds[2 * i] : 0, a->drop_out, a->drop_out > 0.f ? a->seeds[2 * i + 1] : 0, dt, stream));
    x = y;
  }
  return PTPP_OK;
}

extern "C" int ptpp_conv_ln_stack_bwd(const ptpp_conv_ln_stack_bwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->gy && a->x0 && a->x_all && a->z_all && a->mean_all && a->rstd_all && a->wpt && a->gamma && a->dw && a->db &&
                   a->dgamma && a->dbeta && a->gz_all && a->tmp && a->red_scratch,
               "conv_ln_stack_bwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->n > 0 && a->n <= 64 && a->ks > 0 && (a->ks & 1), "conv_ln_stack_bwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "conv_ln_stack_bwd: bad dtype %d", a->dtype);
  const bool fused_in = a->ln_res || a->act_in != PTPP_ACT_NONE || a->drop_in > 0.f;
  ST_CHECK_ARG(!fused_in || a->sum_all, "conv_ln_stack_bwd: sum_all missing");
  const bool want_dz = a->act_in != PTPP_ACT_NONE || a->drop_in > 0.f;
  const bool relu = a->conv_act == PTPP_ACT_RELU;
  ST_CHECK_ARG(a->conv_act == PTPP_ACT_NONE || relu, "conv_ln_stack_bwd: conv activations none | relu");
  const int B = a->B, T = a->T, C = a->C, n = a->n, dt = a->dtype, ks = a->ks, pad = ks / 2;
  const size_t BTC = (size_t)B * T * C, R = (size_t)B * T;
  void* wstream = a->side_stream ? a->side_stream : stream;
  void* ws_w = a->side_stream ? a->ws_side : a->ws_main;
  const size_t ws_w_bytes = a->side_stream ? a->ws_side_bytes : a->ws_main_bytes;
  const int32_t* clen = a->conv_mask ? a->lengths : nullptr;
  // scratch tensors: the gradient w.r.t. a layer's output alternates between g[0] and g[1]; dsum; the pre-relu gradient / dx
  void* g[2] = {at(a->tmp, 0, dt), at(a->tmp, BTC, dt)};
  void* dsum = at(a->tmp, 2 * BTC, dt);
  void* aux = at(a->tmp, 3 * BTC, dt);
  const void* gout = a->gy;
  for (int i = n - 1; i >= 0; --i) {
    const void* xi = i == 0 ? a->x0 : at(a->x_all, (size_t)(i - 1) * BTC, dt);
    const void* z = at(a->z_all, (size_t)i * BTC, dt);
    void* gz = at(a->gz_all, (size_t)i * BTC, dt);
    const int om = a->out_mask == 1 || (a->out_mask == 2 && i == n - 1);
    // LayerNorm backward: dsum = gradient w.r.t. the norm's input (and the residual branch), dz = w.r.t. the conv output
    // when an activation / dropout sits between them; whichever is the conv-output gradient goes to gz (or aux before relu')
    void* conv_grad_dst = relu ? aux : gz;
    void* dsum_dst = want_dz ? dsum : conv_grad_dst;
    if (relu && !want_dz && !fused_in) {
      // the norm's input is the relu output itself: relu' comes from its sign in the same pass (no dsum tensor, no second launch)
      ST_TRY(ptpp_layernorm_bwd_add(gout, z, nullptr, a->gamma[i], a->mean_all + i * R, a->rstd_all + i * R, nullptr, gz, nullptr, 1.0f, 0,
                                    a->dgamma[i], a->dbeta[i], a->lengths, B, T, C, om, PTPP_ACT_RELU, 0.f, 0, a->drop_out,
                                    a->drop_out > 0.f ? a->seeds[2 * i + 1] : 0, dt, a->red_scratch, a->red_bytes, stream));
    } else {
      ST_TRY(ptpp_layernorm_bwd(gout, fused_in ? at(a->sum_all, (size_t)i * BTC, dt) : z, a->act_in != PTPP_ACT_NONE ? z : nullptr,
                                a->gamma[i], a->mean_all + i * R, a->rstd_all + i * R, dsum_dst, want_dz ? conv_grad_dst : nullptr, a->dgamma[i],
                                a->dbeta[i], a->lengths, B, T, C, om, a->act_in, a->drop_in, a->drop_in > 0.f ? a->seeds[2 * i] : 0, a->drop_out,
                                a->drop_out > 0.f ? a->seeds[2 * i + 1] : 0, dt, a->red_scratch, a->red_bytes, stream));
      if (relu) ST_TRY(ptpp_epilogue_bwd(aux, z, gz, nullptr, B, T, C, 1.0f, 1, 0, 0.f, 0, dt, stream));
    }
    const bool need_dx = i > 0 || a->gx != nullptr;
    if (need_dx) {
      void* gnext = i == 0 ? a->gx : g[i & 1];
      // with a residual around the layer the data gradient lands in aux and is added to dsum exactly as autograd adds the
      // two branches (rounded separately); without one dsum_dst == gz and the conv writes the next gradient directly
      void* dx = a->ln_res ? aux : gnext;
      ptpp_conv1d_args c = conv_args(gz, C, a->wpt[i], nullptr, nullptr, 0, dx, C, clen, B, T, C, C, ks, 1, (ks - 1) - pad, PTPP_ACT_NONE, 0,
                                     a->conv_mask ? 1 : 0, dt);
      ST_RT(rt, c, a->wstream_t ? a->wstream_t[i] : nullptr, "conv_ln_stack_bwd");
      if (rt) ST_TRY(ptpp_conv1d_rt_fwd(&c, a->wstream_t[i], 1.0f, stream));
      else ST_TRY(linear_like_ops(c, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
      if (a->ln_res) ST_TRY(ptpp_add3_scale(dx, want_dz ? dsum : gz, nullptr, gnext, 1.0f, (int64_t)BTC, dt, stream));
      gout = gnext;
    }
    if (!a->batched_wgrad) {
      if (a->side_stream) ST_TRY(ptpp_stream_wait(a->side_stream, stream));
      ST_TRY(ptpp_conv1d_wgrad(xi, gz, a->dw[i], a->db[i], clen, B, T, C, C, ks, 1, pad, C, C, a->conv_mask ? 1 : 0, dt, ws_w, ws_w_bytes,
                               wstream));
    }
  }
  if (a->batched_wgrad) {
    ptpp_wgrad_problem pr[64];
    for (int i = 0; i < n; ++i) {
      pr[i].x = i == 0 ? a->x0 : at(a->x_all, (size_t)(i - 1) * BTC, dt);
      pr[i].dy = at(a->gz_all, (size_t)i * BTC, dt);
      pr[i].dw = a->dw[i]; pr[i].dbias = a->db[i]; pr[i].dil = 1; pr[i].pad = pad;
    }
    if (a->side_stream) ST_TRY(ptpp_stream_wait(a->side_stream, stream));
    ST_TRY(ptpp_conv1d_wgrad_batched(pr, n, clen, B, T, C, C, ks, C, C, a->conv_mask ? 1 : 0, dt, ws_w, ws_w_bytes, wstream));
  }
  return PTPP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Conformer encoder block (include/ptpp.h).  Slab / scratch layouts are private to this file.
namespace {

struct CfLayout {  // byte offsets into the forward slab
  size_t n1, h1, x1, n2, qkv, pp, ctx, x2, n3, g, u, d, bno, x3, n4, h2, x4, probs, stats, bnstat, total;
};
CfLayout cf_layout(int B, int T, int C, int F, int H, int L, int dt) {
  const size_t R = (size_t)B * T, es = esize(dt);
  CfLayout o;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at_ = off; off += (bytes + 255) & ~(size_t)255; return at_; };
  o.n1 = take(R * C * es); o.h1 = take(R * F * es); o.x1 = take(R * C * es); o.n2 = take(R * C * es);
  o.qkv = take(R * 3 * C * es); o.pp = take((size_t)L * C * es); o.ctx = take(R * C * es); o.x2 = take(R * C * es);
  o.n3 = take(R * C * es); o.g = take(R * 2 * C * es); o.u = take(R * C * es); o.d = take(R * C * es);
  o.bno = take(R * C * es); o.x3 = take(R * C * es); o.n4 = take(R * C * es); o.h2 = take(R * F * es); o.x4 = take(R * C * es);
  o.probs = take((size_t)B * H * T * T * 4); o.stats = take(10 * R * 4); o.bnstat = take(2 * (size_t)C * 4);
  o.total = off;
  return o;
}
struct CfScratch {  // byte offsets into the backward scratch.  Every tensor a weight-gradient launch reads (dz*) has a region of
                    // its own: those launches run on the side stream while the main stream moves on, so nothing may overwrite
                    // their operands before the caller joins the streams
  size_t gA, gB, gC, t1, t2, gF, dqkv, g2a, dposc, dS, dpos, dz_c[5], dz_f[2], dz_2c, total;
};
CfScratch cf_scratch(int B, int T, int C, int F, int H, int L, int dt) {
  const size_t R = (size_t)B * T, es = esize(dt);
  CfScratch o;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at_ = off; off += (bytes + 255) & ~(size_t)255; return at_; };
  o.gA = take(R * C * es); o.gB = take(R * C * es); o.gC = take(R * C * es); o.t1 = take(R * C * es); o.t2 = take(R * C * es);
  o.gF = take(R * F * es); o.dqkv = take(R * 3 * C * es); o.g2a = take(R * 2 * C * es);
  o.dposc = take((size_t)L * C * es); o.dS = take((size_t)B * H * T * T * 4);
  o.dpos = take((size_t)L * C * 4);
  for (int i = 0; i < 5; ++i) o.dz_c[i] = take(R * C * es);   // ff w2, pw2, attention out, ffm w2, depthwise conv
  for (int i = 0; i < 2; ++i) o.dz_f[i] = take(R * F * es);   // ff w1, ffm w1
  o.dz_2c = take(R * 2 * C * es);                             // pw1
  o.total = off;
  return o;
}
inline void* sl(void* base, size_t off) { return static_cast<char*>(base) + off; }
inline const void* sl(const void* base, size_t off) { return static_cast<const char*>(base) + off; }

int ln_fwd_plain(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, const int32_t* lengths, int B, int T,
                 int C, int out_mask, int dt, void* stream) {
  return ptpp_layernorm_fwd(x, nullptr, g, b, y, nullptr, mean, rstd, lengths, B, T, C, 1e-12f, out_mask, PTPP_ACT_NONE, 0.f, 0, 0.f, 0,
                            dt, stream);
}
int ln_bwd_plain(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, void* dsum, float* dg, float* db,
                 const int32_t* lengths, int B, int T, int C, int out_mask, int dt, void* red, size_t red_bytes, void* stream,
                 const void* add = nullptr, void* dz = nullptr, float dz_scale = 1.f, float dz_drop = 0.f, uint64_t dz_seed = 0) {
  // dz (optional): the backward of the dropout / scale / length mask that fed this block (ptpp_epilogue_bwd on dsum), same pass
  return ptpp_layernorm_bwd_add(dy, x, nullptr, g, mean, rstd, dsum, dz, add, dz_scale, dz != nullptr, dg, db, lengths, B, T, C, out_mask,
                                PTPP_ACT_NONE, dz_drop, dz_seed, 0.f, 0, dt, red, red_bytes, stream);
}

}  // namespace

extern "C" size_t ptpp_conformer_block_slab_bytes(int B, int T, int C, int F, int H, int L, int dtype) {
  return cf_layout(B, T, C, F, H, L, dtype).total;
}
extern "C" size_t ptpp_conformer_block_bwd_scratch_bytes(int B, int T, int C, int F, int H, int L, int dtype) {
  return cf_scratch(B, T, C, F, H, L, dtype).total;
}

extern "C" int ptpp_conformer_block_fwd(const ptpp_conformer_block_fwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->x && a->y && a->pos_emb && a->lengths && a->slab && a->red_scratch, "conformer_block_fwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->F > 0 && a->H > 0 && a->C % a->H == 0 && a->L > 0, "conformer_block_fwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "conformer_block_fwd: bad dtype %d", a->dtype);
  ST_CHECK_ARG((a->p_ffn <= 0.f && a->p_drop <= 0.f) || a->seeds, "conformer_block_fwd: dropout needs seeds");
  const int B = a->B, T = a->T, C = a->C, F = a->F, H = a->H, L = a->L, dt = a->dtype, kf = a->ks_ffn, pf = (a->ks_ffn - 1) / 2;
  const CfLayout lo = cf_layout(B, T, C, F, H, L, dt);
  ST_CHECK_ARG(a->slab_bytes >= lo.total, "conformer_block_fwd: slab too small (%zu bytes, need %zu)", a->slab_bytes, lo.total);
  const ptpp_conformer_weights& w = a->w;
  const size_t R = (size_t)B * T;
  void* S = a->slab;
  float* stats = static_cast<float*>(sl(S, lo.stats));
  float* bnstat = static_cast<float*>(sl(S, lo.bnstat));
  const int32_t* len = a->lengths;
  const uint64_t zero6[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t* sd = a->seeds ? a->seeds : zero6;
  auto seed = [&](int i, float p) { return p > 0.f ? sd[i] : (uint64_t)0; };

  // ---- macaron feed-forward: x1 = x + 0.5 drop(mask w2(drop'(mask relu(w1(mask LN0(x)))))) ----
  ST_TRY(ln_fwd_plain(a->x, w.ln_g[0], w.ln_b[0], sl(S, lo.n1), stats, stats + R, nullptr, B, T, C, 0, dt, stream));
  ptpp_conv1d_args c = conv_args(sl(S, lo.n1), C, w.ffm_w1, w.ffm_b1, nullptr, 0, sl(S, lo.h1), F, len, B, T, C, F, kf, 1, pf, PTPP_ACT_RELU,
                                 1, 1, dt);
  ST_TRY(ffn_conv(c, a->ffn_ws[0], a->p_ffn, seed(0, a->p_ffn), a->ws, a->ws_bytes, stream));
  c = conv_args(sl(S, lo.h1), F, w.ffm_w2, w.ffm_b2, a->x, C, sl(S, lo.x1), C, len, B, T, F, C, kf, 1, pf, PTPP_ACT_NONE, 0, 1, dt);
  c.out_scale = 0.5f;
  ST_TRY(ffn_conv(c, a->ffn_ws[1], a->p_drop, seed(1, a->p_drop), a->ws, a->ws_bytes, stream));
  // ---- self-attention: x2 = x1 + drop(mask out(attn(...))) ----
  ST_TRY(ln_fwd_plain(sl(S, lo.x1), w.ln_g[1], w.ln_b[1], sl(S, lo.n2), stats + 2 * R, stats + 3 * R, nullptr, B, T, C, 0, dt, stream));
  c = conv_args(sl(S, lo.n2), C, w.qkv_w, w.qkv_b, nullptr, 0, sl(S, lo.qkv), 3 * C, nullptr, B, T, C, 3 * C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
  c = conv_args(a->pos_emb, C, w.pos_w, nullptr, nullptr, 0, sl(S, lo.pp), C, nullptr, 1, L, C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
  ST_TRY(ptpp_attention_fwd(sl(S, lo.qkv), at(sl(S, lo.qkv), C, dt), at(sl(S, lo.qkv), 2 * C, dt), sl(S, lo.pp), w.bias_u, w.bias_v,
                            sl(S, lo.ctx), a->save ? static_cast<float*>(sl(S, lo.probs)) : nullptr, len, B, T, H, C / H, 3 * C, C, C,
                            a->variant, 0.f, 0, dt, stream));
  c = conv_args(sl(S, lo.ctx), C, w.out_w, w.out_b, sl(S, lo.x1), C, sl(S, lo.x2), C, len, B, T, C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 1, dt);
  ST_TRY(linear_like_ops(c, a->p_drop, seed(2, a->p_drop), a->ws, a->ws_bytes, stream));
  // ---- convolution module: x3 = x2 + drop(mask pw2(swish(BN(dwconv(glu(mask pw1(LN2(x2)))))))) ----
  ST_TRY(ln_fwd_plain(sl(S, lo.x2), w.ln_g[2], w.ln_b[2], sl(S, lo.n3), stats + 4 * R, stats + 5 * R, nullptr, B, T, C, 0, dt, stream));
  c = conv_args(sl(S, lo.n3), C, w.pw1_w, w.pw1_b, nullptr, 0, sl(S, lo.g), 2 * C, len, B, T, C, 2 * C, 1, 1, 0, PTPP_ACT_NONE, 0, 1, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
  ST_TRY(ptpp_glu_fwd(sl(S, lo.g), sl(S, lo.u), (int64_t)R, C, dt, stream));
  ST_TRY(ptpp_dwconv1d(sl(S, lo.u), w.dw_w, w.dw_b, sl(S, lo.d), len, B, T, C, a->ks_dw, 0, dt, stream));
  const float* bmean = bnstat;
  const float* brstd = bnstat + C;
  if (a->bn_train) {
    ST_TRY(ptpp_bn_stats(sl(S, lo.d), (int64_t)R, C, a->bn_momentum, a->bn_eps, w.bn_rmean, w.bn_rvar, bnstat, bnstat + C, dt,
                         a->red_scratch, a->red_bytes, stream));
  } else {
    ST_CHECK_ARG(w.bn_mean_in && w.bn_rstd_in, "conformer_block_fwd: eval-mode BatchNorm needs its statistics");
    bmean = w.bn_mean_in;
    brstd = w.bn_rstd_in;
  }
  ST_TRY(ptpp_bn_act_fwd(sl(S, lo.d), bmean, brstd, w.bn_g, w.bn_b, sl(S, lo.bno), (int64_t)R, C, PTPP_ACT_SWISH, dt, stream));
  c = conv_args(sl(S, lo.bno), C, w.pw2_w, w.pw2_b, sl(S, lo.x2), C, sl(S, lo.x3), C, len, B, T, C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 1, dt);
  ST_TRY(linear_like_ops(c, a->p_drop, seed(3, a->p_drop), a->ws, a->ws_bytes, stream));
  // ---- feed-forward ----
  ST_TRY(ln_fwd_plain(sl(S, lo.x3), w.ln_g[3], w.ln_b[3], sl(S, lo.n4), stats + 6 * R, stats + 7 * R, nullptr, B, T, C, 0, dt, stream));
  c = conv_args(sl(S, lo.n4), C, w.ff_w1, w.ff_b1, nullptr, 0, sl(S, lo.h2), F, len, B, T, C, F, kf, 1, pf, PTPP_ACT_RELU, 1, 1, dt);
  ST_TRY(ffn_conv(c, a->ffn_ws[2], a->p_ffn, seed(4, a->p_ffn), a->ws, a->ws_bytes, stream));
  c = conv_args(sl(S, lo.h2), F, w.ff_w2, w.ff_b2, sl(S, lo.x3), C, sl(S, lo.x4), C, len, B, T, F, C, kf, 1, pf, PTPP_ACT_NONE, 0, 1, dt);
  c.out_scale = 0.5f;
  ST_TRY(ffn_conv(c, a->ffn_ws[3], a->p_drop, seed(5, a->p_drop), a->ws, a->ws_bytes, stream));
  return ln_fwd_plain(sl(S, lo.x4), w.ln_g[4], w.ln_b[4], a->y, stats + 8 * R, stats + 9 * R, len, B, T, C, 1, dt, stream);
}

extern "C" int ptpp_conformer_block_bwd(const ptpp_conformer_block_bwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->gy && a->gx && a->x && a->pos_emb && a->lengths && a->slab && a->scratch && a->red_scratch,
               "conformer_block_bwd: null pointer");
  ST_CHECK_ARG(a->B > 0 && a->T > 0 && a->C > 0 && a->F > 0 && a->H > 0 && a->C % a->H == 0 && a->L > 0, "conformer_block_bwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "conformer_block_bwd: bad dtype %d", a->dtype);
  const int B = a->B, T = a->T, C = a->C, F = a->F, H = a->H, L = a->L, dt = a->dtype, kf = a->ks_ffn, pf = (a->ks_ffn - 1) / 2;
  const CfLayout lo = cf_layout(B, T, C, F, H, L, dt);
  const CfScratch sc = cf_scratch(B, T, C, F, H, L, dt);
  ST_CHECK_ARG(a->scratch_bytes >= sc.total, "conformer_block_bwd: scratch too small (%zu bytes, need %zu)", a->scratch_bytes, sc.total);
  const ptpp_conformer_weights& w = a->w;
  const ptpp_conformer_grads& g = a->g;
  const size_t R = (size_t)B * T;
  const int64_t RC = (int64_t)R * C;
  const void* S = a->slab;
  void* X = a->scratch;
  const float* stats = static_cast<const float*>(sl(S, lo.stats));
  const float* bnstat = static_cast<const float*>(sl(S, lo.bnstat));
  const int32_t* len = a->lengths;
  const uint64_t zero6[6] = {0, 0, 0, 0, 0, 0};
  const uint64_t* sd = a->seeds ? a->seeds : zero6;
  auto seed = [&](int i, float p) { return p > 0.f ? sd[i] : (uint64_t)0; };
  void* wstream = a->side_stream ? a->side_stream : stream;
  void* ws_w = a->side_stream ? a->ws_side : a->ws_main;
  const size_t ws_w_bytes = a->side_stream ? a->ws_side_bytes : a->ws_main_bytes;
  // The block's weight gradients are collected and issued as ONE grouped call at the end (ptpp_conv1d_wgrad_grouped: at phone
  // level a launch + reduction per layer is fixed cost; 26 launches -> 1), on the side stream after everything enqueued on the
  // main one (functional.wgrad_stream).  Their operands (slab tensors, dz regions of the scratch) stay untouched until then.
  ptpp_wgrad_gproblem wg[16];
  int nwg = 0;
  auto wgrad = [&](const void* x, int ldx, const void* dy, int lddy, float* dw, float* db, const int32_t* lengths, int Bn, int Tn, int cin,
                   int cout, int ks, int pad, int in_mask) -> int {
    ST_CHECK_ARG(nwg < 16, "conformer_block_bwd: too many weight gradients");
    ptpp_wgrad_gproblem& q = wg[nwg++];
    q.x = x; q.dy = dy; q.dw = dw; q.dbias = db; q.lengths = in_mask ? lengths : nullptr;
    q.B = Bn; q.T = Tn; q.Cin = cin; q.Cout = cout; q.ks = ks; q.dil = 1; q.pad = pad; q.ldx = ldx; q.lddy = lddy;
    return PTPP_OK;
  };
  void *gA = sl(X, sc.gA), *gB = sl(X, sc.gB), *gC = sl(X, sc.gC), *t1 = sl(X, sc.t1), *t2 = sl(X, sc.t2);
  void* gF = sl(X, sc.gF);

  // backward of one feed-forward half: gres = gradient w.r.t. res + 0.5 drop(...) ; returns the gradient w.r.t. LN input in t1
  int t1_nsplit = 1;
  auto ffn_bwd = [&](const void* gres, const void* h, const void* n, const void* w1t, const void* w2t, float* dw1, float* db1, float* dw2,
                     float* db2, int s1, int s2, void* dz2, void* dzF, const void* w1ts, const void* w2ts) -> int {
    // (dz2 = gres * 0.5 * dropout mask s2, masked rows zero: written by the LayerNorm backward that produced gres)
    ptpp_conv1d_args c = conv_args(dz2, C, w2t, nullptr, nullptr, 0, gF, F, len, B, T, C, F, kf, 1, (kf - 1) - pf, PTPP_ACT_NONE, 0, 0, dt);
    ST_TRY(wgrad(h, F, dz2, C, dw2, db2, len, B, T, F, C, kf, pf, 0));
    if (w2ts && ptpp_conv1d_rt_ex_supported(c.Cin, c.Cout, c.ks, c.dil, c.act, c.dtype)) {
      // (round 6: the ReLU / dropout backward from the conv's epilogue -- no pass over gF, which is scratch here)
      ST_TRY(ptpp_conv1d_rt_fwd_ex_relu_bwd(&c, w2ts, h, dzF, a->p_ffn, a->ws_main, a->ws_main_bytes, stream));
    } else {
      ST_TRY(ffn_conv(c, w2ts, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
      ST_TRY(ptpp_epilogue_bwd(gF, h, dzF, len, B, T, F, 1.0f, 1, 1, a->p_ffn, seed(s1, a->p_ffn), dt, stream));
    }
    c = conv_args(dzF, F, w1t, nullptr, nullptr, 0, t1, C, len, B, T, F, C, kf, 1, (kf - 1) - pf, PTPP_ACT_NONE, 0, 1, dt);
    t1_nsplit = 1;
    const char* fe = getenv("PTPP_FFN_LN_SPLITK");  // (0: the finishing launch of rounds 1-5; A/B and the bit-identity test)
    if (!(fe && fe[0] == '0') && w1ts && ptpp_conv1d_rt_ex_supported(c.Cin, c.Cout, c.ks, c.dil, c.act, c.dtype) && a->ws_main &&
        ((uintptr_t)a->ws_main & 15) == 0) {
      // (round 6: where this launch is split over Cin its finishing pass is left to the LayerNorm backward that reads t1)
      ST_TRY(ptpp_conv1d_rt_fwd_ex_partial(&c, w1ts, a->ws_main, a->ws_main_bytes, &t1_nsplit, stream));
    } else {
      ST_TRY(ffn_conv(c, w1ts, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
    }
    return wgrad(n, C, dzF, F, dw1, db1, len, B, T, C, F, kf, pf, 1);
  };
  // the LayerNorm backward that follows a feed-forward half: dy = t1, or the partial sums the half's last conv left in ws_main
  auto ln_bwd_t1 = [&](const void* x, const float* gam, const float* mean, const float* rstd, void* dsum, float* dg, float* db,
                       const int32_t* lengths, const void* add, void* dz, float dz_scale, float dz_drop, uint64_t dz_seed) -> int {
    if (t1_nsplit > 1)
      return ptpp_layernorm_bwd_add_splitk(static_cast<const float*>(a->ws_main), t1_nsplit, len, x, gam, mean, rstd, dsum, dz, add, dz_scale,
                                           dz != nullptr, dg, db, lengths, B, T, C, 0, dz_drop, dz_seed, dt, a->red_scratch, a->red_bytes,
                                           stream);
    return ln_bwd_plain(t1, x, gam, mean, rstd, dsum, dg, db, lengths, B, T, C, 0, dt, a->red_scratch, a->red_bytes, stream, add, dz, dz_scale,
                        dz_drop, dz_seed);
  };

  // ---- final LayerNorm ----
  ST_TRY(ln_bwd_plain(a->gy, sl(S, lo.x4), w.ln_g[4], stats + 8 * R, stats + 9 * R, gA, g.ln_g[4], g.ln_b[4], len, B, T, C, 1, dt,
                      a->red_scratch, a->red_bytes, stream, nullptr, sl(X, sc.dz_c[0]), 0.5f, a->p_drop, seed(5, a->p_drop)));
  // ---- feed-forward ----
  ST_TRY(ffn_bwd(gA, sl(S, lo.h2), sl(S, lo.n4), a->ff_w1t, a->ff_w2t, g.ff_w1, g.ff_b1, g.ff_w2, g.ff_b2, 4, 5, sl(X, sc.dz_c[0]),
                 sl(X, sc.dz_f[0]), a->ffn_wts[2], a->ffn_wts[3]));
  // gradient w.r.t. x3 = gA + LayerNorm input gradient, and the pointwise conv's dropout backward, one pass
  void* dz_pw2 = sl(X, sc.dz_c[1]);
  ST_TRY(ln_bwd_t1(sl(S, lo.x3), w.ln_g[3], stats + 6 * R, stats + 7 * R, gB, g.ln_g[3], g.ln_b[3], len, gA, dz_pw2, 1.0f, a->p_drop,
                   seed(3, a->p_drop)));
  // ---- convolution module ----
  ptpp_conv1d_args c = conv_args(dz_pw2, C, a->pw2_wt, nullptr, nullptr, 0, t2, C, len, B, T, C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
  ST_TRY(wgrad(sl(S, lo.bno), C, dz_pw2, C, g.pw2_w, g.pw2_b, len, B, T, C, C, 1, 0, 0));
  const float* bmean = a->bn_train ? bnstat : w.bn_mean_in;
  const float* brstd = a->bn_train ? bnstat + C : w.bn_rstd_in;
  // (the depthwise conv's output gradient gets a region of its own: its weight gradient joins the block's other weight
  //  gradients on the side stream at the end instead of sitting on the main one -- 60 us per block of a serial, atomics-bound kernel)
  void* dz_dw = sl(X, sc.dz_c[4]);
  ST_TRY(ptpp_bn_act_bwd_acc(sl(S, lo.d), t2, bmean, brstd, w.bn_g, w.bn_b, g.bn_sums, a->bn_dbeta, a->bn_dgamma, dz_dw, (int64_t)R, C,
                             PTPP_ACT_SWISH, a->bn_train, dt, a->red_scratch, a->red_bytes, stream));
  ST_TRY(ptpp_dwconv1d(dz_dw, w.dw_w, nullptr, t2, len, B, T, C, a->ks_dw, 1, dt, stream));
  // (the GLU backward writes the masked gradient of the pointwise conv's output at once: round 5 ran a masking pass after it)
  ST_TRY(ptpp_glu_bwd_masked(sl(S, lo.g), t2, sl(X, sc.dz_2c), len, B, T, C, dt, stream));
  c = conv_args(sl(X, sc.dz_2c), 2 * C, a->pw1_wt, nullptr, nullptr, 0, t1, C, len, B, T, 2 * C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
  ST_TRY(wgrad(sl(S, lo.n3), C, sl(X, sc.dz_2c), 2 * C, g.pw1_w, g.pw1_b, len, B, T, C, 2 * C, 1, 0, 0));
  void* dz_out = sl(X, sc.dz_c[2]);
  ST_TRY(ln_bwd_plain(t1, sl(S, lo.x2), w.ln_g[2], stats + 4 * R, stats + 5 * R, gC, g.ln_g[2], g.ln_b[2], len, B, T, C, 0, dt,
                      a->red_scratch, a->red_bytes, stream, gB, dz_out, 1.0f, a->p_drop, seed(2, a->p_drop)));  // gradient w.r.t. x2
  // ---- self-attention ----
  c = conv_args(dz_out, C, a->out_wt, nullptr, nullptr, 0, t2, C, len, B, T, C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
  ST_TRY(wgrad(sl(S, lo.ctx), C, dz_out, C, g.out_w, g.out_b, len, B, T, C, C, 1, 0, 0));
  void* dqkv = sl(X, sc.dqkv);
  float* dpos = static_cast<float*>(sl(X, sc.dpos));
  const void* qkv = sl(S, lo.qkv);
  ST_TRY(ptpp_attention_bwd(qkv, at(qkv, C, dt), at(qkv, 2 * C, dt), sl(S, lo.pp), w.bias_u, w.bias_v,
                            static_cast<const float*>(sl(S, lo.probs)), t2, static_cast<float*>(sl(X, sc.dS)), dqkv, at(dqkv, C, dt),
                            at(dqkv, 2 * C, dt), dpos, g.bias_u, g.bias_v, len, B, T, H, C / H, 3 * C, C, C, 3 * C, a->variant, 0.f, 0, dt,
                            a->red_scratch, a->red_bytes, stream));
  // positional projection (no bias, its input is a constant table): weight gradient only
  const void* dposc = dpos;
  if (dt != PTPP_F32) {
    ST_TRY(ptpp_cast_from_f32(dpos, sl(X, sc.dposc), (int64_t)L * C, dt, stream));
    dposc = sl(X, sc.dposc);
  }
  ST_TRY(wgrad(a->pos_emb, C, dposc, C, g.pos_w, nullptr, nullptr, 1, L, C, C, 1, 0, 0));
  // fused q | k | v projection
  c = conv_args(dqkv, 3 * C, a->qkv_wt, nullptr, nullptr, 0, t1, C, nullptr, B, T, 3 * C, C, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
  ST_TRY(linear_like_ops(c, 0.f, 0, a->ws_main, a->ws_main_bytes, stream));
  float* const dws[3] = {g.q_w, g.k_w, g.v_w};
  float* const dbs[3] = {g.q_b, g.k_b, g.v_b};
  for (int i = 0; i < 3; ++i) ST_TRY(wgrad(sl(S, lo.n2), C, at(dqkv, (size_t)i * C, dt), 3 * C, dws[i], dbs[i], nullptr, B, T, C, C, 1, 0, 0));
  ST_TRY(ln_bwd_plain(t1, sl(S, lo.x1), w.ln_g[1], stats + 2 * R, stats + 3 * R, gA, g.ln_g[1], g.ln_b[1], len, B, T, C, 0, dt,
                      a->red_scratch, a->red_bytes, stream, gC, sl(X, sc.dz_c[3]), 0.5f, a->p_drop, seed(1, a->p_drop)));  // gradient w.r.t. x1
  // ---- macaron feed-forward ----
  ST_TRY(ffn_bwd(gA, sl(S, lo.h1), sl(S, lo.n1), a->ffm_w1t, a->ffm_w2t, g.ffm_w1, g.ffm_b1, g.ffm_w2, g.ffm_b2, 0, 1, sl(X, sc.dz_c[3]),
                 sl(X, sc.dz_f[1]), a->ffn_wts[0], a->ffn_wts[1]));
  ST_TRY(ln_bwd_t1(a->x, w.ln_g[0], stats, stats + R, a->gx, g.ln_g[0], g.ln_b[0], nullptr, gA, nullptr, 1.f, 0.f, 0));
  if (a->side_stream) ST_TRY(ptpp_stream_wait(a->side_stream, stream));
  // Round 6: the three launches (depthwise, k = 9 group, 1 x 1 group) are independent and none fills the chip (76 / 192 / 32
  // workgroups): with a second side stream the k = 9 group runs beside the other two.  It matters at the END of the backward,
  // where the main stream has nothing left and the step waits for the first encoder block's weight gradients.
  void* wstream2 = a->side_stream && a->side_stream2 ? a->side_stream2 : nullptr;
  if (wstream2) ST_TRY(ptpp_stream_wait(wstream2, stream));
  ST_TRY(ptpp_dwconv1d_wgrad(sl(S, lo.u), dz_dw, g.dw_w, g.dw_b, len, B, T, C, a->ks_dw, dt, wstream2 ? wstream2 : wstream));
  return ptpp_conv1d_wgrad_grouped2(wg, nwg, dt, ws_w, ws_w_bytes, wstream, wstream2);
}


// ---- reference-encoder convolution stack ----------------------------------------------------------------------------
namespace {

constexpr int RE_MAX_LAYERS = 16;
struct ReLayout {  // byte offsets; per layer: geometry, im2col rows, pre-BN output, statistics
  int H[RE_MAX_LAYERS + 1], W[RE_MAX_LAYERS + 1], cin[RE_MAX_LAYERS], cinq[RE_MAX_LAYERS];
  int64_t rows[RE_MAX_LAYERS];
  size_t col[RE_MAX_LAYERS], z[RE_MAX_LAYERS], stat[RE_MAX_LAYERS], ypp[2], total;
  // backward scratch
  size_t dz[RE_MAX_LAYERS], g[2], dcol, btotal;
};
ReLayout re_layout(int B, int H, int W, int n, const int32_t* cout, int dt) {
  ReLayout o;
  memset(&o, 0, sizeof(o));
  const size_t es = esize(dt);
  size_t off = 0, boff = 0;
  auto take = [&](size_t& cur, size_t bytes) { const size_t at_ = cur; cur += (bytes + 255) & ~(size_t)255; return at_; };
  o.H[0] = H; o.W[0] = W;
  size_t ymax = 0, dcolmax = 0;
  for (int i = 0; i < n; ++i) {
    o.cin[i] = i == 0 ? 1 : cout[i - 1];
    o.cinq[i] = i == 0 ? 8 : cout[i - 1];
    o.H[i + 1] = (o.H[i] - 1) / 2 + 1; o.W[i + 1] = (o.W[i] - 1) / 2 + 1;
    o.rows[i] = (int64_t)B * o.H[i + 1] * o.W[i + 1];
    o.col[i] = take(off, (size_t)o.rows[i] * 9 * o.cinq[i] * es);
    o.z[i] = take(off, (size_t)o.rows[i] * cout[i] * es);
    o.stat[i] = take(off, 2 * (size_t)cout[i] * 4);
    const size_t yb = (size_t)o.rows[i] * cout[i] * es;
    if (yb > ymax) ymax = yb;
    o.dz[i] = take(boff, yb);
    if (i > 0 && (size_t)o.rows[i] * 9 * o.cinq[i] * es > dcolmax) dcolmax = (size_t)o.rows[i] * 9 * o.cinq[i] * es;
  }
  o.ypp[0] = take(off, ymax); o.ypp[1] = take(off, ymax);
  o.total = off;
  o.g[0] = take(boff, ymax); o.g[1] = take(boff, ymax);
  o.dcol = take(boff, dcolmax);
  o.btotal = boff;
  return o;
}
bool re_shape_ok(int B, int H, int W, int n, const int32_t* cout) {
  if (!(B > 0 && H > 0 && W > 0 && n > 0 && n <= RE_MAX_LAYERS && cout)) return false;
  for (int i = 0; i < n; ++i)
    if (cout[i] <= 0 || cout[i] % 8) return false;
  return true;
}

}  // namespace

extern "C" size_t ptpp_refenc_convs_slab_bytes(int B, int H, int W, int nlayer, const int32_t* cout, int dtype) {
  return re_shape_ok(B, H, W, nlayer, cout) ? re_layout(B, H, W, nlayer, cout, dtype).total : 0;
}
extern "C" size_t ptpp_refenc_convs_bwd_scratch_bytes(int B, int H, int W, int nlayer, const int32_t* cout, int dtype) {
  return re_shape_ok(B, H, W, nlayer, cout) ? re_layout(B, H, W, nlayer, cout, dtype).btotal : 0;
}

extern "C" int ptpp_refenc_convs_fwd(const ptpp_refenc_convs_fwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->x && a->y && a->cout && a->w && a->wp_fwd && a->wp_bwd && a->bn_g && a->bn_b && a->bn_rmean && a->bn_rvar &&
                   a->slab && a->red_scratch,
               "refenc_convs_fwd: null pointer");
  ST_CHECK_ARG(re_shape_ok(a->B, a->H, a->W, a->nlayer, a->cout), "refenc_convs_fwd: bad shape (channels must be multiples of 8)");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "refenc_convs_fwd: bad dtype %d", a->dtype);
  const int n = a->nlayer, dt = a->dtype;
  const ReLayout lo = re_layout(a->B, a->H, a->W, n, a->cout, dt);
  ST_CHECK_ARG(a->slab_bytes >= lo.total, "refenc_convs_fwd: slab too small (%zu bytes, need %zu)", a->slab_bytes, lo.total);
  void* S = a->slab;
  const void* xin = a->x;
  for (int i = 0; i < n; ++i) {
    const int co = a->cout[i], K = 9 * lo.cinq[i];
    ST_TRY(ptpp_pack_conv2d_3x3(a->w[i], a->wp_fwd[i], i > 0 ? a->wp_bwd[i] : nullptr, co, lo.cin[i], lo.cinq[i], dt, stream));
    if (i == 0) ST_TRY(ptpp_im2col3x3s2_c1(xin, sl(S, lo.col[0]), a->B, lo.H[0], lo.W[0], dt, stream));
    else ST_TRY(ptpp_im2col3x3s2(xin, sl(S, lo.col[i]), a->B, lo.H[i], lo.W[i], lo.cinq[i], dt, stream));
    ST_CHECK_ARG(lo.rows[i] <= INT32_MAX, "refenc_convs_fwd: too many rows");
    ptpp_conv1d_args c = conv_args(sl(S, lo.col[i]), K, a->wp_fwd[i], nullptr, nullptr, 0, sl(S, lo.z[i]), co, nullptr, 1, (int)lo.rows[i], K,
                                   co, 1, 1, 0, PTPP_ACT_NONE, 0, 0, dt);
    ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
    float* st = static_cast<float*>(sl(S, lo.stat[i]));
    ST_TRY(ptpp_bn_stats(sl(S, lo.z[i]), lo.rows[i], co, a->bn_momentum, a->bn_eps, a->bn_rmean[i], a->bn_rvar[i], st, st + co, dt,
                         a->red_scratch, a->red_bytes, stream));
    void* yout = i == n - 1 ? a->y : sl(S, lo.ypp[i & 1]);
    ST_TRY(ptpp_bn_act_fwd(sl(S, lo.z[i]), st, st + co, a->bn_g[i], a->bn_b[i], yout, lo.rows[i], co, PTPP_ACT_RELU, dt, stream));
    xin = yout;
  }
  return PTPP_OK;
}

extern "C" int ptpp_refenc_convs_bwd(const ptpp_refenc_convs_bwd_args* a, void* stream) {
  ST_CHECK_ARG(a && a->gy && a->cout && a->wp_bwd && a->bn_g && a->bn_b && a->dwg && a->bn_sums && a->slab && a->scratch && a->red_scratch,
               "refenc_convs_bwd: null pointer");
  ST_CHECK_ARG(re_shape_ok(a->B, a->H, a->W, a->nlayer, a->cout), "refenc_convs_bwd: bad shape");
  ST_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "refenc_convs_bwd: bad dtype %d", a->dtype);
  const int n = a->nlayer, dt = a->dtype;
  const ReLayout lo = re_layout(a->B, a->H, a->W, n, a->cout, dt);
  ST_CHECK_ARG(a->scratch_bytes >= lo.btotal, "refenc_convs_bwd: scratch too small (%zu bytes, need %zu)", a->scratch_bytes, lo.btotal);
  const void* S = a->slab;
  void* X = a->scratch;
  const void* g = a->gy;
  for (int i = n - 1; i >= 0; --i) {
    const int co = a->cout[i], K = 9 * lo.cinq[i];
    const float* st = static_cast<const float*>(sl(S, lo.stat[i]));
    void* dz = sl(X, lo.dz[i]);
    ST_TRY(ptpp_bn_act_bwd(sl(S, lo.z[i]), g, st, st + co, a->bn_g[i], a->bn_b[i], a->bn_sums[i], dz, lo.rows[i], co, PTPP_ACT_RELU, 1, dt,
                           a->red_scratch, a->red_bytes, stream));
    if (i > 0) {
      ptpp_conv1d_args c = conv_args(dz, co, a->wp_bwd[i], nullptr, nullptr, 0, sl(X, lo.dcol), K, nullptr, 1, (int)lo.rows[i], co, K, 1, 1,
                                     0, PTPP_ACT_NONE, 0, 0, dt);
      ST_TRY(linear_like_ops(c, 0.f, 0, a->ws, a->ws_bytes, stream));
    }
    ST_TRY(ptpp_conv1d_wgrad(sl(S, lo.col[i]), dz, a->dwg[i], nullptr, nullptr, 1, (int)lo.rows[i], K, co, 1, 1, 0, K, co, 0, dt, a->ws,
                             a->ws_bytes, stream));
    if (i > 0) {
      void* gx = sl(X, lo.g[i & 1]);
      ST_TRY(ptpp_col2im3x3s2(sl(X, lo.dcol), gx, a->B, lo.H[i], lo.W[i], lo.cinq[i], dt, stream));
      g = gx;
    }
  }
  return PTPP_OK;
}
