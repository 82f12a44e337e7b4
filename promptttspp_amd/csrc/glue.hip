// Round 6: the "glue" of the training step as a handful of launches.
//
// profiles/r06_main_order_before.txt: of the 598 main-stream launches of a step ~250 were torch-native tensor ops -- masks, casts,
// the loss algebra and its ~120 autograd nodes, the DDPM q_sample chain, the step-embedding sinusoid, slice / select backward
// fills -- each 2-8 us of device time and 10-20 us of host time, with the host needing 13.6 ms to enqueue a 14.6 ms step.  The
// kernels here replace those chains one for one (same arithmetic, same rounding points where the reference fixes them):
//
//   tts_losses_fwd / _bwd     every loss of PromptTTSMDNDurCFG.forward (reference models/prompttts_mdn_v2_final/model.py:126-183)
//                             in ONE launch each way, log-softmax of both MDN heads included (modules/mdn.py:37-66, 81-175)
//   q_sample_bct              mel (B, M, T) -> normalised, noised, channels-last denoiser input (modules/diffusion.py:110-115,304-313)
//   step_sinusoid, mish       DiffNet step embedding (modules/denoiser.py:23-41)
//   embed_cl_fwd / _bwd       PhonemeEmbedding (layers/embedding.py:21-36) with the phone mask, deterministic table gradient
//   scalar_embed_add          x + Conv1d(1 -> C, k = 1)(track) * mask: pitch / energy embedding (modules/variance_adaptor.py:139-146)
//   l2norm_fwd / _bwd         F.normalize(style_emb, dim = 1) (model.py:108,148-150)
//   durations_cumsum          int32 running frame count per phone for the length regulator (utils/model.py:37-47)
//   bcast_add_rows            x + style_emb over every phone (model.py:111) and its gradient (sum over phones)
#include "ptpp_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int LS_NB_DEC = 128, LS_NB_PV = 16, LS_NB_DUR = 8, LS_NB_STY = 8;
constexpr int LS_NB = LS_NB_DEC + LS_NB_PV + LS_NB_DUR + LS_NB_STY;

// one element (row, d) of a dimension-wise mixture NLL on the RAW head output y = [pi logits G*D | log_sigma G*D | mu G*D] of a row:
// log_pi = log_softmax over the G components (mdn.py:56-60), clamps / 5-sigma clip / logsumexp as mdn.py:81-175 (the arithmetic
// of mdn_nll_fwd_kernel, misc.hip).  G <= 32.
__device__ __forceinline__ float mdn_lse(const float* __restrict__ y, int G, int D, int d) {
  float m = -__builtin_inff();
  for (int g = 0; g < G; ++g) m = fmaxf(m, y[g * D + d]);
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += expf(y[g * D + d] - m);
  return m + logf(s);
}
__device__ __forceinline__ float mdn_nll_elem(const float* __restrict__ y, int G, int D, int d, float tg, float lp_min, float ls_min) {
  const float lse = mdn_lse(y, G, D, d);
  const int GD = G * D;
  float m = -__builtin_inff(), s = 0.f;
  for (int g = 0; g < G; ++g) {
    const int k = g * D + d;
    const float lp = fmaxf(y[k] - lse, lp_min), ls = fmaxf(y[GD + k], ls_min);
    const float sg = expf(ls);
    const float dd = fmaxf(fminf(tg - y[2 * GD + k], 5.f * sg), -5.f * sg);
    const float z = dd / sg;
    const float ll = -0.5f * z * z - ls - 0.91893853320467274f + lp;
    if (ll > m) { s = s * expf(m - ll) + 1.f; m = ll; } else { s += expf(ll - m); }
  }
  return -(m + logf(s));
}
// gradient of go * nll wrt the raw row; dy has y's layout
__device__ __forceinline__ void mdn_nll_elem_bwd(const float* __restrict__ y, float* __restrict__ dy, int G, int D, int d, float tg,
                                                 float nl, float go, float lp_min, float ls_min) {
  const float lse = mdn_lse(y, G, D, d);
  const int GD = G * D;
  float sum_dlp = 0.f;
  for (int g = 0; g < G; ++g) {
    const int k = g * D + d;
    const float lpr = y[k] - lse, lsr = y[GD + k];
    const float lp = fmaxf(lpr, lp_min), ls = fmaxf(lsr, ls_min);
    const float sg = expf(ls);
    const float raw = tg - y[2 * GD + k];
    const bool clip = fabsf(raw) > 5.f * sg;
    const float dd = fmaxf(fminf(raw, 5.f * sg), -5.f * sg);
    const float z = dd / sg;
    const float ll = -0.5f * z * z - ls - 0.91893853320467274f + lp;
    const float dll = -go * expf(ll + nl);  // d(-logsumexp) / d ll_g = -softmax_g
    const float dlp = lpr >= lp_min ? dll : 0.f;
    dy[k] = dlp;  // finished below (log-softmax backward needs the sum)
    dy[GD + k] = lsr >= ls_min ? (clip ? -dll : dll * (z * z - 1.f)) : 0.f;
    dy[2 * GD + k] = clip ? 0.f : dll * z / sg;
    sum_dlp += dlp;
  }
  for (int g = 0; g < G; ++g) {
    const int k = g * D + d;
    dy[k] -= expf(y[k] - lse) * sum_dlp;
  }
}
// to_log_scale (utils/model.py: log of the non-zero durations, zeros stay)
__device__ __forceinline__ float log_scale(float d) { return d != 0.f ? logf(fmaxf(d, 1e-30f)) : d; }

__device__ __forceinline__ float block_sum256(float v, float* red) {  // all 256 threads; result valid in thread 0
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void tts_losses_fwd_kernel(const ptpp_tts_loss_args a) {
  __shared__ float red[4];
  __shared__ int last;
  const int bid = blockIdx.x, tid = threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
  if (bid < LS_NB_DEC) {
    // decoder: sum over valid rows of |noise - pred|
    const T* pred = reinterpret_cast<const T*>(a.pred);
    const int M = a.M;
    const int64_t n = (int64_t)a.B * a.Tf * M;
    if (M % 4 == 0) {
      for (int64_t v = (int64_t)bid * 256 + tid; v < n / 4; v += (int64_t)LS_NB_DEC * 256) {
        const int64_t e = v * 4, row = e / M;
        const int b = (int)(row / a.Tf), t = (int)(row - (int64_t)b * a.Tf);
        if (t >= a.flen[b]) continue;
        const f32x4 p = Elem<T>::ld4(pred + e), q = *reinterpret_cast<const f32x4*>(a.noise + e);
        s0 += (fabsf(q[0] - p[0]) + fabsf(q[1] - p[1])) + (fabsf(q[2] - p[2]) + fabsf(q[3] - p[3]));
      }
    } else {
      for (int64_t e = (int64_t)bid * 256 + tid; e < n; e += (int64_t)LS_NB_DEC * 256) {
        const int64_t row = e / M;
        const int b = (int)(row / a.Tf), t = (int)(row - (int64_t)b * a.Tf);
        if (t < a.flen[b]) s0 += fabsf(a.noise[e] - Elem<T>::ld(pred + e));
      }
    }
  } else if (bid < LS_NB_DEC + LS_NB_PV) {
    // log-F0 / V-UV tracks: NO mask (the prediction is masked, the targets are zero-padded: model.py:168-170)
    const T* pv = reinterpret_cast<const T*>(a.pv);
    const int64_t n = (int64_t)a.B * a.Tf;
    for (int64_t r = (int64_t)(bid - LS_NB_DEC) * 256 + tid; r < n; r += (int64_t)LS_NB_PV * 256) {
      s0 += fabsf(Elem<T>::ld(pv + 2 * r) - a.cf0_tgt[r]);
      s1 += fabsf(Elem<T>::ld(pv + 2 * r + 1) - a.vuv_tgt[r]);
    }
  } else if (bid < LS_NB_DEC + LS_NB_PV + LS_NB_DUR) {
    // duration MDN (D = 1): mean over valid phones of the NLL of the log duration
    const int G = a.G_dur;
    const int64_t n = (int64_t)a.B * a.Tp;
    for (int64_t r = (int64_t)(bid - LS_NB_DEC - LS_NB_PV) * 256 + tid; r < n; r += (int64_t)LS_NB_DUR * 256) {
      const int b = (int)(r / a.Tp), t = (int)(r - (int64_t)b * a.Tp);
      float nl = 0.f;
      if (t < a.plen[b]) {
        nl = mdn_nll_elem(a.y_dur + r * 3 * G, G, 1, 0, log_scale(a.dur[r]), a.lp_min, a.ls_min);
        s0 += nl;
      }
      a.nll_dur[r] = nl;
    }
  } else {
    // style MDN: (B, 1) rows x D dimensions, mean over everything
    const int G = a.G_sty, D = a.D_sty;
    const int64_t n = (int64_t)a.B * D;
    for (int64_t i = (int64_t)(bid - LS_NB_DEC - LS_NB_PV - LS_NB_DUR) * 256 + tid; i < n; i += (int64_t)LS_NB_STY * 256) {
      const int b = (int)(i / D), d = (int)(i - (int64_t)b * D);
      const float nl = mdn_nll_elem(a.y_sty + (int64_t)b * 3 * G * D, G, D, d, a.sty_tgt[i], a.lp_min, a.ls_min);
      a.nll_sty[i] = nl;
      s0 += nl;
    }
  }
  const float t0 = block_sum256(s0, red);
  const float t1 = block_sum256(s1, red);
  float* scratch = reinterpret_cast<float*>(a.scratch);
  if (tid == 0) {
    scratch[2 + 2 * bid] = t0;
    scratch[3 + 2 * bid] = t1;
    __threadfence();
    const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(scratch), 1u);
    last = t == (unsigned)LS_NB - 1u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last block to arrive adds the partial sums in a fixed order (bit-reproducible)
  int nf_i = 0, np_i = 0;
  if (tid < 64) {
    for (int b = tid; b < a.B; b += 64) {
      nf_i += a.flen[b] < a.Tf ? a.flen[b] : a.Tf;
      np_i += a.plen[b] < a.Tp ? a.plen[b] : a.Tp;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      nf_i += __shfl_xor(nf_i, o, 64);
      np_i += __shfl_xor(np_i, o, 64);
    }
  }
  if (tid == 0) {
    const volatile float* part = scratch + 2;
    float S[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // dec, cf0, vuv, dur, style
    int i = 0;
    for (; i < LS_NB_DEC; ++i) S[0] += part[2 * i];
    for (; i < LS_NB_DEC + LS_NB_PV; ++i) { S[1] += part[2 * i]; S[2] += part[2 * i + 1]; }
    for (; i < LS_NB_DEC + LS_NB_PV + LS_NB_DUR; ++i) S[3] += part[2 * i];
    for (; i < LS_NB; ++i) S[4] += part[2 * i];
    const float nf = (float)nf_i, np = (float)np_i;
    const float dec = (S[0] / nf) * (1.f / a.dec_scale);  // (torch divides by a host scalar as a multiplication by its reciprocal)
    const float cf0 = S[1] / nf, vuv = S[2] / nf;
    const float dur = S[3] / np;
    const float sty = S[4] / (float)((int64_t)a.B * a.D_sty);
    a.total[0] = (((dec + dur) + cf0) + vuv) + sty;
    a.comps[0] = dec; a.comps[1] = dur; a.comps[2] = cf0; a.comps[3] = vuv; a.comps[4] = sty;
    a.comps[5] = nf; a.comps[6] = np;
    *reinterpret_cast<unsigned*>(scratch) = 0u;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void tts_losses_bwd_kernel(const ptpp_tts_loss_args a, const float* __restrict__ g_total,
                                                            const float* __restrict__ g_comps, T* __restrict__ dpred,
                                                            T* __restrict__ dpv, float* __restrict__ dy_dur, float* __restrict__ dy_sty,
                                                            int nb_dec, int nb_pv, int nb_dur) {
  const int bid = blockIdx.x, tid = threadIdx.x;
  const float gt = g_total ? g_total[0] : 0.f;
  const float nf = a.comps[5], np = a.comps[6];
  if (bid < nb_dec) {
    const float coef = ((gt + (g_comps ? g_comps[0] : 0.f)) * (1.f / a.dec_scale)) / nf;
    const T* pred = reinterpret_cast<const T*>(a.pred);
    const int M = a.M;
    const int64_t n = (int64_t)a.B * a.Tf * M;
    if (M % 4 == 0) {
      const int64_t v = (int64_t)bid * 256 + tid;
      if (v >= n / 4) return;
      const int64_t e = v * 4, row = e / M;
      const int b = (int)(row / a.Tf), t = (int)(row - (int64_t)b * a.Tf);
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (t < a.flen[b]) {
        const f32x4 p = Elem<T>::ld4(pred + e), q = *reinterpret_cast<const f32x4*>(a.noise + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = p[i] - q[i];
          g[i] = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
        }
      }
      Elem<T>::st4(dpred + e, g);
    } else {
      for (int k = 0; k < 4; ++k) {
        const int64_t e = ((int64_t)bid * 256 + tid) * 4 + k;
        if (e >= n) return;
        const int64_t row = e / M;
        const int b = (int)(row / a.Tf), t = (int)(row - (int64_t)b * a.Tf);
        float g = 0.f;
        if (t < a.flen[b]) {
          const float d = Elem<T>::ld(pred + e) - a.noise[e];
          g = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
        }
        Elem<T>::st(dpred + e, g);
      }
    }
  } else if (bid < nb_dec + nb_pv) {
    const float c0 = (gt + (g_comps ? g_comps[2] : 0.f)) / nf, c1 = (gt + (g_comps ? g_comps[3] : 0.f)) / nf;
    const T* pv = reinterpret_cast<const T*>(a.pv);
    const int64_t r = (int64_t)(bid - nb_dec) * 256 + tid;
    if (r >= (int64_t)a.B * a.Tf) return;
    const float d0 = Elem<T>::ld(pv + 2 * r) - a.cf0_tgt[r], d1 = Elem<T>::ld(pv + 2 * r + 1) - a.vuv_tgt[r];
    Elem<T>::st(dpv + 2 * r, d0 > 0.f ? c0 : (d0 < 0.f ? -c0 : 0.f));
    Elem<T>::st(dpv + 2 * r + 1, d1 > 0.f ? c1 : (d1 < 0.f ? -c1 : 0.f));
  } else if (bid < nb_dec + nb_pv + nb_dur) {
    const float go = (gt + (g_comps ? g_comps[1] : 0.f)) / np;
    const int G = a.G_dur;
    const int64_t r = (int64_t)(bid - nb_dec - nb_pv) * 256 + tid;
    if (r >= (int64_t)a.B * a.Tp) return;
    const int b = (int)(r / a.Tp), t = (int)(r - (int64_t)b * a.Tp);
    float* dy = dy_dur + r * 3 * G;
    if (t < a.plen[b]) {
      mdn_nll_elem_bwd(a.y_dur + r * 3 * G, dy, G, 1, 0, log_scale(a.dur[r]), a.nll_dur[r], go, a.lp_min, a.ls_min);
    } else {
      for (int k = 0; k < 3 * G; ++k) dy[k] = 0.f;
    }
  } else {
    const int G = a.G_sty, D = a.D_sty;
    const float go = (gt + (g_comps ? g_comps[4] : 0.f)) / (float)((int64_t)a.B * D);
    const int64_t i = (int64_t)(bid - nb_dec - nb_pv - nb_dur) * 256 + tid;
    if (i >= (int64_t)a.B * D) return;
    const int b = (int)(i / D), d = (int)(i - (int64_t)b * D);
    mdn_nll_elem_bwd(a.y_sty + (int64_t)b * 3 * G * D, dy_sty + (int64_t)b * 3 * G * D, G, D, d, a.sty_tgt[i], a.nll_sty[i], go,
                     a.lp_min, a.ls_min);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// q_sample: out[b, t, m] = T( sa[t_b] * norm(mel[b, m, t]) + sb[t_b] * noise[b, t, m] ), the reference's unfused f32 sequence
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void q_sample_bct_kernel(const float* __restrict__ mel, const float* __restrict__ noise,
                                                          const int64_t* __restrict__ step, const float* __restrict__ sa,
                                                          const float* __restrict__ sb, float norm_scale, float a_min, float a_max,
                                                          int use_scale, T* __restrict__ out, int M, int Tn, int K) {
#pragma clang fp contract(off)
  __shared__ float tile[128][65];
  const int b = blockIdx.y, t0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int i = tid; i < M * 64; i += 256) {
    const int c = i >> 6, tt = i & 63;
    tile[c][tt] = t0 + tt < Tn ? mel[((int64_t)b * M + c) * Tn + t0 + tt] : 0.f;
  }
  __syncthreads();
  int64_t k = step[b];
  k = k < 0 ? 0 : (k >= K ? K - 1 : k);
  const float ca = sa[k], cb = sb[k];
  // (torch divides a tensor by a host scalar as a multiplication by the scalar's f32 reciprocal: same bits here)
  const float inv_scale = 1.f / norm_scale, inv_span = 1.f / (a_max - a_min);
  for (int i = tid; i < M * 64; i += 256) {
    const int tt = i / M, c = i - tt * M;
    if (t0 + tt >= Tn) break;
    const float x = tile[c][tt];
    float xn;
    if (use_scale) xn = x * inv_scale;
    else xn = (x - a_min) * inv_span * 2.f - 1.f;
    const int64_t o = ((int64_t)b * Tn + t0 + tt) * M + c;
    const float p0 = ca * xn, p1 = cb * noise[o];
    Elem<T>::st(out + o, p0 + p1);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// step embedding pieces
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void step_sinusoid_kernel(const int64_t* __restrict__ step, int64_t scale, float cf, int half, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = i / half, k = i - b * half;
  const float f = expf((float)k * cf);
  const float e = (float)(scale * step[b]) * f;
  out[(int64_t)b * 2 * half + k] = sinf(e);
  out[(int64_t)b * 2 * half + half + k] = cosf(e);
}
__global__ void mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float sp = v > 20.f ? v : log1pf(expf(v));  // F.softplus (threshold 20)
  y[i] = v * tanhf(sp);
}
__global__ void mish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], g = gy[i];
  const float sp = v > 20.f ? v : log1pf(expf(v));
  const float th = tanhf(sp);
  const float z = expf(v);
  const float dsp = v > 20.f ? 1.f : z / (z + 1.f);  // softplus backward
  gx[i] = g * th + (g * v) * (1.f - th * th) * dsp;
}

// ------------------------------------------------------------------------------------------------------------------------------
// phoneme embedding
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_cl_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                          const int* __restrict__ lengths, float scale, int do_scale,
                                                          T* __restrict__ out, int64_t rows, int Tn, int C, int V) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one 4-channel vector
  const int c4 = C / 4;
  if (v >= rows * c4) return;
  const int64_t row = v / c4;
  const int c = (int)(v - row * c4) * 4;
  const int b = (int)(row / Tn), t = (int)(row - (int64_t)b * Tn);
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (!lengths || t < lengths[b]) {
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    o = *reinterpret_cast<const f32x4*>(table + id * C + c);
    if (do_scale) o *= scale;
  }
  Elem<T>::st4(out + row * C + c, o);
}
// dtable[v, c] += scale * sum over valid rows with ids[row] == v of dout[row, c]; one block per vocabulary entry walks the rows in
// order (bit-reproducible); the padding row gets nothing (nn.Embedding(padding_idx))
template <typename T>
__global__ __launch_bounds__(256) void embed_cl_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout,
                                                          const int* __restrict__ lengths, float scale, int do_scale,
                                                          float* __restrict__ dtable, int64_t rows, int Tn, int C, int padding_idx) {
  const int v = blockIdx.x;
  if (v == padding_idx) return;
  __shared__ int hit[256];
  __shared__ int nhit;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + threadIdx.x;
    float acc = 0.f;
    for (int64_t r0 = 0; r0 < rows; r0 += 256) {
      // 256 rows at a time: every thread tests one row, the matching ones are compacted in row order
      __syncthreads();
      if (threadIdx.x == 0) nhit = 0;
      __syncthreads();
      const int64_t r = r0 + threadIdx.x;
      bool m = false;
      if (r < rows && ids[r] == v) {
        const int b = (int)(r / Tn), t = (int)(r - (int64_t)b * Tn);
        m = !lengths || t < lengths[b];
      }
      // ordered compaction: ballot per wave, waves in order
      const unsigned long long bal = __ballot(m);
      __shared__ int wcnt[4];
      if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(bal);
      __syncthreads();
      if (m) {
        int base = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wcnt[w];
        hit[base + __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int)threadIdx.x;
      }
      if (threadIdx.x == 0) nhit = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
      if (c < C)
        for (int k = 0; k < nhit; ++k) acc += Elem<T>::ld(dout + (r0 + hit[k]) * C + c);
    }
    if (c < C) dtable[(int64_t)v * C + c] += do_scale ? acc * scale : acc;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// x + (track * w + bias) * mask, the 1 -> C pointwise embedding of a scalar track, rounded like the tensor-op form: the embedding is
// rounded to x's dtype before the sum.  Backward: dw / db column sums through the replicated reduction scratch.
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void scalar_embed_add_kernel(const T* __restrict__ x, const float* __restrict__ track,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const int* __restrict__ lengths, T* __restrict__ out, int64_t rows,
                                                              int Tn, int C) {
#pragma clang fp contract(off)
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = C / 4;
  if (v >= rows * c4) return;
  const int64_t row = v / c4;
  const int c = (int)(v - row * c4) * 4;
  const int b = (int)(row / Tn), t = (int)(row - (int64_t)b * Tn);
  f32x4 o = Elem<T>::ld4(x + row * C + c);
  if (!lengths || t < lengths[b]) {
    const float tv = track[row];
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c), bv = *reinterpret_cast<const f32x4*>(bias + c);
    T e[4];
    f32x4 ev;
#pragma unroll
    for (int i = 0; i < 4; ++i) ev[i] = tv * wv[i] + bv[i];
    Elem<T>::st4(e, ev);
    ev = Elem<T>::ld4(e);  // (the embedding in x's dtype, then the sum)
    o += ev;
  }
  Elem<T>::st4(out + row * C + c, o);
}
// per block: rows [blockIdx * RPB, ...): thread (j = tid / (C/4) row lane, cv = tid % (C/4)) -- C == 256 only (64 x 4 layout)
template <typename T>
__global__ __launch_bounds__(256) void scalar_embed_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ track,
                                                              const int* __restrict__ lengths, void* scratch, int64_t rows, int Tn,
                                                              int C) {
  __shared__ float tot[2 * 1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = C / 256;  // vectors per lane (C in {256, 512, 768, 1024})
  f32x4 aw[4], ab[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) aw[i] = ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < rows; row += (int64_t)gridDim.x * 4) {
    const int b = (int)(row / Tn), t = (int)(row - (int64_t)b * Tn);
    if (lengths && t >= lengths[b]) continue;
    const float tv = track[row];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) {
        const f32x4 d = Elem<T>::ld4(dout + row * C + i * 256 + lane * 4);
        aw[i] += d * tv;
        ab[i] += d;
      }
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) tot[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {  // waves in order: a fixed summation order inside the block
    if (wv == w)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nv)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            tot[i * 256 + lane * 4 + e] += aw[i][e];
            tot[C + i * 256 + lane * 4 + e] += ab[i][e];
          }
    __syncthreads();
  }
  red_block_add(scratch, tot, 2 * C);
}

// ------------------------------------------------------------------------------------------------------------------------------
// F.normalize(x, dim = channels) of (rows, C) f32: one wave per row
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm,
                                                       int C, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[(int64_t)row * C + c]; s += v * v; }
  s = wave_sum(s);
  const float n = sqrtf(s), d = fmaxf(n, eps);
  for (int c = lane; c < C; c += 64) y[(int64_t)row * C + c] = x[(int64_t)row * C + c] / d;
  if (lane == 0) nrm[row] = n;
}
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ nrm,
                                                       const float* __restrict__ gy, float* __restrict__ gx, int C, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float n = nrm[row], d = fmaxf(n, eps);
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += gy[(int64_t)row * C + c] * y[(int64_t)row * C + c];
  dot = wave_sum(dot);
  const bool live = n >= eps;  // clamp_min passes the gradient where the norm is not clamped
  for (int c = lane; c < C; c += 64) {
    const float g = gy[(int64_t)row * C + c];
    gx[(int64_t)row * C + c] = live ? (g - y[(int64_t)row * C + c] * dot) / d : g / d;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// running frame count per phone, int32 (clamped like the tensor-op form); one wave per utterance
// ------------------------------------------------------------------------------------------------------------------------------
template <typename D>
__global__ __launch_bounds__(64) void durations_cumsum_kernel(const D* __restrict__ dur, int* __restrict__ cum, int Tp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int64_t carry = 0;
  for (int t0 = 0; t0 < Tp; t0 += 64) {
    const int t = t0 + lane;
    int64_t v = t < Tp ? (int64_t)dur[(int64_t)b * Tp + t] : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    v += carry;
    if (t < Tp) cum[(int64_t)b * Tp + t] = (int)(v > 2147483647ll ? 2147483647ll : v);
    carry = __shfl(v, 63, 64);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// y[b, t, :] = x[b, t, :] + T(e[b, :]) (every row, padded ones too: model.py:111); de[b, :] = sum_t dy[b, t, :]
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bcast_add_rows_kernel(const T* __restrict__ x, const float* __restrict__ e, T* __restrict__ y,
                                                            int64_t rows, int Tn, int C) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = C / 4;
  if (v >= rows * c4) return;
  const int64_t row = v / c4;
  const int c = (int)(v - row * c4) * 4;
  const int b = (int)(row / Tn);
  T er[4];
  Elem<T>::st4(er, *reinterpret_cast<const f32x4*>(e + (int64_t)b * C + c));
  Elem<T>::st4(y + row * C + c, Elem<T>::ld4(x + row * C + c) + Elem<T>::ld4(er));
}
template <typename T>
__global__ __launch_bounds__(256) void rows_sum_kernel(const T* __restrict__ dy, float* __restrict__ de, int Tn, int C) {
  // block (b, 64-channel group): 4 waves take every 4th row, combined in wave order
  __shared__ float part[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int t = wv; t < Tn; t += 4) s += Elem<T>::ld(dy + ((int64_t)b * Tn + t) * C + c);
  part[wv][threadIdx.x & 63] = s;
  __syncthreads();
  if (wv == 0 && c < C) de[(int64_t)b * C + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Linear / 1x1 Conv1d with 1-4 output channels (the pitch / V-UV head, modules/variance_adaptor.py:52-62: 256 -> 2): a row is one
// wave's dot products, HBM-bound; the GEMM path needed a padded operand and, in the backward, ~15 tensor ops for Cout % 4 != 0.
// f32 weights and accumulation.  Backward: dx, and (dw, db) as block column sums through the replicated reduction scratch.
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T, int CO>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const int* __restrict__ lengths,
                                                              T* __restrict__ y, int64_t rows, int Tn, int Cin) {
  const int lane = threadIdx.x & 63;
  const int nv = Cin / 256;
  f32x4 wr[CO][4];
#pragma unroll
  for (int o = 0; o < CO; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[o][i] = i < nv ? *reinterpret_cast<const f32x4*>(w + (int64_t)o * Cin + i * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const int b = (int)(row / Tn), t = (int)(row - (int64_t)b * Tn);
    const bool valid = !lengths || t < lengths[b];
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    if (valid) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nv) {
          const f32x4 xv = Elem<T>::ld4(x + row * Cin + i * 256 + lane * 4);
#pragma unroll
          for (int o = 0; o < CO; ++o) acc[o] += (xv[0] * wr[o][i][0] + xv[1] * wr[o][i][1]) + (xv[2] * wr[o][i][2] + xv[3] * wr[o][i][3]);
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = wave_sum(acc[o]);
    if (lane == 0)
#pragma unroll
      for (int o = 0; o < CO; ++o) Elem<T>::st(y + row * CO + o, valid ? acc[o] + (bias ? bias[o] : 0.f) : 0.f);
  }
}
template <typename T, int CO>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w,
                                                              const int* __restrict__ lengths, T* __restrict__ dx, void* scratch,
                                                              int64_t rows, int Tn, int Cin) {
  __shared__ float tot[CO * 1024 + 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = Cin / 256;
  f32x4 wr[CO][4], aw[CO][4];
  float ab[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    ab[o] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wr[o][i] = i < nv ? *reinterpret_cast<const f32x4*>(w + (int64_t)o * Cin + i * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      aw[o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < rows; row += (int64_t)gridDim.x * 4) {
    const int b = (int)(row / Tn), t = (int)(row - (int64_t)b * Tn);
    const bool valid = !lengths || t < lengths[b];
    float g[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      g[o] = valid ? Elem<T>::ld(dy + row * CO + o) : 0.f;
      ab[o] += g[o];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
          const f32x4 xv = Elem<T>::ld4(x + row * Cin + i * 256 + lane * 4);
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            d += wr[o][i] * g[o];
            aw[o][i] += xv * g[o];
          }
        }
        if (dx) Elem<T>::st4(dx + row * Cin + i * 256 + lane * 4, d);
      }
  }
  const int KC = CO * Cin + CO;
  for (int i = threadIdx.x; i < KC; i += 256) tot[i] = 0.f;
  __syncthreads();
  for (int k = 0; k < 4; ++k) {  // waves in order: a fixed summation order inside the block
    if (wv == k) {
#pragma unroll
      for (int o = 0; o < CO; ++o) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nv)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[o * Cin + i * 256 + lane * 4 + e] += aw[o][i][e];
        if (lane == 0) tot[CO * Cin + o] += ab[o];
      }
    }
    __syncthreads();
  }
  red_block_add(scratch, tot, KC);
}
}  // namespace

// ================================================================================================================================
extern "C" int64_t ptpp_tts_losses_scratch_bytes(void) { return (int64_t)(2 + 2 * LS_NB) * 4; }

static int loss_args_ok(const ptpp_tts_loss_args* a) {
  PTPP_CHECK_ARG(a && a->pred && a->noise && a->flen && a->pv && a->cf0_tgt && a->vuv_tgt && a->y_dur && a->dur && a->plen && a->y_sty &&
                     a->sty_tgt && a->total && a->comps && a->nll_dur && a->nll_sty && a->scratch,
                 "tts_losses: null pointer");
  PTPP_CHECK_ARG(a->B > 0 && a->Tf > 0 && a->Tp > 0 && a->M > 0 && a->G_dur > 0 && a->G_dur <= 32 && a->G_sty > 0 && a->G_sty <= 32 &&
                     a->D_sty > 0 && a->dec_scale != 0.f,
                 "tts_losses: bad shape B=%d Tf=%d Tp=%d M=%d G=%d/%d D=%d", a->B, a->Tf, a->Tp, a->M, a->G_dur, a->G_sty, a->D_sty);
  PTPP_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "tts_losses: dtype %d (f32 / bf16)", a->dtype);
  return PTPP_OK;
}

extern "C" int ptpp_tts_losses_fwd(const ptpp_tts_loss_args* a, void* stream) {
  if (int rc = loss_args_ok(a)) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == PTPP_F32)
    hipLaunchKernelGGL(tts_losses_fwd_kernel<float>, dim3(LS_NB), dim3(256), 0, st, *a);
  else
    hipLaunchKernelGGL(tts_losses_fwd_kernel<bf16_raw>, dim3(LS_NB), dim3(256), 0, st, *a);
  PTPP_CHECK_LAUNCH("tts_losses_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_tts_losses_bwd(const ptpp_tts_loss_args* a, const float* g_total, const float* g_comps, void* dpred, void* dpv,
                                   float* dy_dur, float* dy_sty, void* stream) {
  if (int rc = loss_args_ok(a)) return rc;
  PTPP_CHECK_ARG((g_total || g_comps) && dpred && dpv && dy_dur && dy_sty, "tts_losses_bwd: null pointer");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n_dec = (int64_t)a->B * a->Tf * a->M;
  const int nb_dec = (int)(((n_dec + 3) / 4 + 255) / 256), nb_pv = (int)(((int64_t)a->B * a->Tf + 255) / 256);
  const int nb_dur = (int)(((int64_t)a->B * a->Tp + 255) / 256), nb_sty = (int)(((int64_t)a->B * a->D_sty + 255) / 256);
  const dim3 grid((unsigned)(nb_dec + nb_pv + nb_dur + nb_sty));
  if (a->dtype == PTPP_F32)
    hipLaunchKernelGGL(tts_losses_bwd_kernel<float>, grid, dim3(256), 0, st, *a, g_total, g_comps, reinterpret_cast<float*>(dpred),
                       reinterpret_cast<float*>(dpv), dy_dur, dy_sty, nb_dec, nb_pv, nb_dur);
  else
    hipLaunchKernelGGL(tts_losses_bwd_kernel<bf16_raw>, grid, dim3(256), 0, st, *a, g_total, g_comps, reinterpret_cast<bf16_raw*>(dpred),
                       reinterpret_cast<bf16_raw*>(dpv), dy_dur, dy_sty, nb_dec, nb_pv, nb_dur);
  PTPP_CHECK_LAUNCH("tts_losses_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_q_sample_bct(const float* mel, const float* noise, const int64_t* step, const float* sqrt_ac, const float* sqrt_1mac,
                                 int K, float norm_scale, float a_min, float a_max, int use_scale, void* out, int B, int M, int T,
                                 int dtype, void* stream) {
  PTPP_CHECK_ARG(mel && noise && step && sqrt_ac && sqrt_1mac && out && B > 0 && M > 0 && M <= 128 && T > 0 && K > 0,
                 "q_sample_bct: bad args (M <= 128)");
  PTPP_CHECK_ARG(use_scale ? norm_scale != 0.f : a_max != a_min, "q_sample_bct: degenerate normalisation");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((T + 63) / 64), (unsigned)B);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(q_sample_bct_kernel<float>, grid, dim3(256), 0, st, mel, noise, step, sqrt_ac, sqrt_1mac, norm_scale, a_min, a_max,
                       use_scale, reinterpret_cast<float*>(out), M, T, K);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(q_sample_bct_kernel<bf16_raw>, grid, dim3(256), 0, st, mel, noise, step, sqrt_ac, sqrt_1mac, norm_scale, a_min,
                       a_max, use_scale, reinterpret_cast<bf16_raw*>(out), M, T, K);
  else
    PTPP_CHECK_ARG(false, "q_sample_bct: dtype %d (f32 / bf16)", dtype);
  PTPP_CHECK_LAUNCH("q_sample_bct");
  return PTPP_OK;
}

extern "C" int ptpp_step_sinusoid(const int64_t* step, int64_t scale, float neg_log_rate, int B, int half, float* out, void* stream) {
  PTPP_CHECK_ARG(step && out && B > 0 && half > 1, "step_sinusoid: bad args");
  const int n = B * half;
  hipLaunchKernelGGL(step_sinusoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), step,
                     scale, neg_log_rate, half, out, n);
  PTPP_CHECK_LAUNCH("step_sinusoid");
  return PTPP_OK;
}

extern "C" int ptpp_mish_fwd(const float* x, float* y, int64_t n, void* stream) {
  PTPP_CHECK_ARG(x && y && n > 0, "mish_fwd: bad args");
  hipLaunchKernelGGL(mish_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n);
  PTPP_CHECK_LAUNCH("mish_fwd");
  return PTPP_OK;
}
extern "C" int ptpp_mish_bwd(const float* x, const float* gy, float* gx, int64_t n, void* stream) {
  PTPP_CHECK_ARG(x && gy && gx && n > 0, "mish_bwd: bad args");
  hipLaunchKernelGGL(mish_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gy, gx, n);
  PTPP_CHECK_LAUNCH("mish_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_embed_cl_fwd(const int64_t* ids, const float* table, const int32_t* lengths, float scale, int do_scale, void* out,
                                 int B, int T, int C, int V, int dtype, void* stream) {
  PTPP_CHECK_ARG(ids && table && out && B > 0 && T > 0 && C > 0 && C % 4 == 0 && V > 0, "embed_cl_fwd: bad args (C % 4 == 0)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  const dim3 grid((unsigned)((rows * (C / 4) + 255) / 256));
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(embed_cl_fwd_kernel<float>, grid, dim3(256), 0, st, ids, table, lengths, scale, do_scale, reinterpret_cast<float*>(out),
                       rows, T, C, V);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(embed_cl_fwd_kernel<bf16_raw>, grid, dim3(256), 0, st, ids, table, lengths, scale, do_scale,
                       reinterpret_cast<bf16_raw*>(out), rows, T, C, V);
  else
    PTPP_CHECK_ARG(false, "embed_cl_fwd: dtype %d (f32 / bf16)", dtype);
  PTPP_CHECK_LAUNCH("embed_cl_fwd");
  return PTPP_OK;
}
extern "C" int ptpp_embed_cl_bwd(const int64_t* ids, const void* dout, const int32_t* lengths, float scale, int do_scale, float* dtable,
                                 int B, int T, int C, int V, int padding_idx, int dtype, void* stream) {
  PTPP_CHECK_ARG(ids && dout && dtable && B > 0 && T > 0 && C > 0 && V > 0, "embed_cl_bwd: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(embed_cl_bwd_kernel<float>, dim3((unsigned)V), dim3(256), 0, st, ids, reinterpret_cast<const float*>(dout), lengths,
                       scale, do_scale, dtable, rows, T, C, padding_idx);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(embed_cl_bwd_kernel<bf16_raw>, dim3((unsigned)V), dim3(256), 0, st, ids, reinterpret_cast<const bf16_raw*>(dout),
                       lengths, scale, do_scale, dtable, rows, T, C, padding_idx);
  else
    PTPP_CHECK_ARG(false, "embed_cl_bwd: dtype %d (f32 / bf16)", dtype);
  PTPP_CHECK_LAUNCH("embed_cl_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_scalar_embed_add(const void* x, const float* track, const float* w, const float* bias, const int32_t* lengths,
                                     void* out, int B, int T, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && track && w && bias && out && B > 0 && T > 0 && C > 0 && C % 4 == 0, "scalar_embed_add: bad args (C % 4 == 0)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  const dim3 grid((unsigned)((rows * (C / 4) + 255) / 256));
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(scalar_embed_add_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(x), track, w, bias, lengths,
                       reinterpret_cast<float*>(out), rows, T, C);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(scalar_embed_add_kernel<bf16_raw>, grid, dim3(256), 0, st, reinterpret_cast<const bf16_raw*>(x), track, w, bias,
                       lengths, reinterpret_cast<bf16_raw*>(out), rows, T, C);
  else
    PTPP_CHECK_ARG(false, "scalar_embed_add: dtype %d (f32 / bf16)", dtype);
  PTPP_CHECK_LAUNCH("scalar_embed_add");
  return PTPP_OK;
}
extern "C" int ptpp_scalar_embed_bwd(const void* dout, const float* track, const int32_t* lengths, float* dw, float* db, int B, int T,
                                     int C, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(dout && track && dw && db && B > 0 && T > 0 && C > 0 && C % 256 == 0 && C <= 1024,
                 "scalar_embed_bwd: bad args (C in {256, 512, 768, 1024})");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  RedSlot slot = red_take(scratch, scratch_bytes, 2 * C, st);
  PTPP_CHECK_ARG(slot.ptr, "scalar_embed_bwd: reduction scratch missing or too small");
  int64_t nb = (rows + 3) / 4;
  if (nb > 1024) nb = 1024;
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(scalar_embed_bwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const float*>(dout), track,
                       lengths, slot.ptr, rows, T, C);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(scalar_embed_bwd_kernel<bf16_raw>, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const bf16_raw*>(dout),
                       track, lengths, slot.ptr, rows, T, C);
  else
    PTPP_CHECK_ARG(false, "scalar_embed_bwd: dtype %d (f32 / bf16)", dtype);
  red_finish(slot, 2 * C, dw, C, db, 1, st);
  PTPP_CHECK_LAUNCH("scalar_embed_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_l2norm_fwd(const float* x, float* y, float* nrm, int rows, int C, float eps, void* stream) {
  PTPP_CHECK_ARG(x && y && nrm && rows > 0 && C > 0, "l2norm_fwd: bad args");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)rows), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), x, y, nrm, C, eps);
  PTPP_CHECK_LAUNCH("l2norm_fwd");
  return PTPP_OK;
}
extern "C" int ptpp_l2norm_bwd(const float* y, const float* nrm, const float* gy, float* gx, int rows, int C, float eps, void* stream) {
  PTPP_CHECK_ARG(y && nrm && gy && gx && rows > 0 && C > 0, "l2norm_bwd: bad args");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)rows), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), y, nrm, gy, gx, C, eps);
  PTPP_CHECK_LAUNCH("l2norm_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_durations_cumsum(const void* dur, int is_float, int32_t* cum, int B, int Tp, void* stream) {
  PTPP_CHECK_ARG(dur && cum && B > 0 && Tp > 0, "durations_cumsum: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (is_float)
    hipLaunchKernelGGL(durations_cumsum_kernel<float>, dim3((unsigned)B), dim3(64), 0, st, reinterpret_cast<const float*>(dur), cum, Tp);
  else
    hipLaunchKernelGGL(durations_cumsum_kernel<int64_t>, dim3((unsigned)B), dim3(64), 0, st, reinterpret_cast<const int64_t*>(dur), cum, Tp);
  PTPP_CHECK_LAUNCH("durations_cumsum");
  return PTPP_OK;
}

extern "C" int ptpp_bcast_add_rows(const void* x, const float* e, void* y, int B, int T, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && e && y && B > 0 && T > 0 && C > 0 && C % 4 == 0, "bcast_add_rows: bad args (C % 4 == 0)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  const dim3 grid((unsigned)((rows * (C / 4) + 255) / 256));
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(bcast_add_rows_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(x), e, reinterpret_cast<float*>(y),
                       rows, T, C);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(bcast_add_rows_kernel<bf16_raw>, grid, dim3(256), 0, st, reinterpret_cast<const bf16_raw*>(x), e,
                       reinterpret_cast<bf16_raw*>(y), rows, T, C);
  else
    PTPP_CHECK_ARG(false, "bcast_add_rows: dtype %d (f32 / bf16)", dtype);
  PTPP_CHECK_LAUNCH("bcast_add_rows");
  return PTPP_OK;
}
extern "C" int ptpp_rows_sum(const void* dy, float* de, int B, int T, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(dy && de && B > 0 && T > 0 && C > 0, "rows_sum: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)B);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(rows_sum_kernel<float>, grid, dim3(256), 0, st, reinterpret_cast<const float*>(dy), de, T, C);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(rows_sum_kernel<bf16_raw>, grid, dim3(256), 0, st, reinterpret_cast<const bf16_raw*>(dy), de, T, C);
  else
    PTPP_CHECK_ARG(false, "rows_sum: dtype %d (f32 / bf16)", dtype);
  PTPP_CHECK_LAUNCH("rows_sum");
  return PTPP_OK;
}

extern "C" int ptpp_linear_small_fwd(const void* x, const float* w, const float* bias, const int32_t* lengths, void* y, int B, int T,
                                     int Cin, int Cout, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && w && y && B > 0 && T > 0 && Cin > 0 && Cin % 256 == 0 && Cin <= 1024 && Cout >= 1 && Cout <= 4,
                 "linear_small_fwd: bad args (Cin in {256 .. 1024}, Cout 1-4)");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "linear_small_fwd: dtype %d (f32 / bf16)", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  int64_t nb = (rows + 3) / 4;
  if (nb > 4096) nb = 4096;
#define LS_FWD(TT, CO_)                                                                                                            \
  hipLaunchKernelGGL((linear_small_fwd_kernel<TT, CO_>), dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const TT*>(x), w, bias, \
                     lengths, reinterpret_cast<TT*>(y), rows, T, Cin)
#define LS_FWD_T(TT)                                                                                   \
  switch (Cout) { case 1: LS_FWD(TT, 1); break; case 2: LS_FWD(TT, 2); break; case 3: LS_FWD(TT, 3); break; default: LS_FWD(TT, 4); }
  if (dtype == PTPP_F32) { LS_FWD_T(float) } else { LS_FWD_T(bf16_raw) }
#undef LS_FWD_T
#undef LS_FWD
  PTPP_CHECK_LAUNCH("linear_small_fwd");
  return PTPP_OK;
}
extern "C" int ptpp_linear_small_bwd(const void* x, const void* dy, const float* w, const int32_t* lengths, void* dx, float* dw, float* db,
                                     int B, int T, int Cin, int Cout, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(x && dy && w && dw && db && B > 0 && T > 0 && Cin > 0 && Cin % 256 == 0 && Cin <= 1024 && Cout >= 1 && Cout <= 4,
                 "linear_small_bwd: bad args (Cin in {256 .. 1024}, Cout 1-4)");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "linear_small_bwd: dtype %d (f32 / bf16)", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  const int KC = Cout * Cin + Cout;
  RedSlot slot = red_take(scratch, scratch_bytes, KC, st);
  PTPP_CHECK_ARG(slot.ptr, "linear_small_bwd: reduction scratch missing or too small");
  int64_t nb = (rows + 3) / 4;
  if (nb > 1024) nb = 1024;
#define LS_BWD(TT, CO_)                                                                                                          \
  hipLaunchKernelGGL((linear_small_bwd_kernel<TT, CO_>), dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const TT*>(x),       \
                     reinterpret_cast<const TT*>(dy), w, lengths, reinterpret_cast<TT*>(dx), slot.ptr, rows, T, Cin)
#define LS_BWD_T(TT)                                                                                   \
  switch (Cout) { case 1: LS_BWD(TT, 1); break; case 2: LS_BWD(TT, 2); break; case 3: LS_BWD(TT, 3); break; default: LS_BWD(TT, 4); }
  if (dtype == PTPP_F32) { LS_BWD_T(float) } else { LS_BWD_T(bf16_raw) }
#undef LS_BWD_T
#undef LS_BWD
  red_finish(slot, KC, dw, Cout * Cin, db, 1, st);
  PTPP_CHECK_LAUNCH("linear_small_bwd");
  return PTPP_OK;
}
