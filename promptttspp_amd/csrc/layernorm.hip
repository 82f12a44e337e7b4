// LayerNorm over the channel dimension of channels-last rows (HBM-bound).
// One 64-lane wave per row: each lane owns 4 consecutive channels per 256-wide
// slab (8/16-byte vector loads), statistics by wavefront shuffles in f32.
//
//   s = drop_in(act_in(x)) + res            (act_in: none | GELU; res optional)
//   y = drop_out(LN(s) * gamma + beta) * [t < len]
//
// covers the plain LayerNorms (esp layer_norm.py, layers/norm.py), the frame
// prior's  x = LN(x + dropout(gelu(conv(x))))  (frame_prior.py:85-89) and the
// predictors'  dropout(LN(relu(conv)))*mask  (variance_adaptor.py:31-36).
#include "ptpp_common.h"

namespace {

struct LnDrop {
  unsigned in_thresh, out_thresh;
  float in_inv, out_inv;
  unsigned long long in_seed, out_seed;
  const void* add;  // backward only: dsum = add + (the LayerNorm input gradient rounded to the storage type), or NULL
  float dz_scale;   // backward only: dz = dsum * dz_scale * [t < len if dz_mask] * dropmask_in * act_in'(z)
  int dz_mask;
  int dz_stored;    // dz from dsum AS STORED (rounded to T) -- what a separate pass over dsum would read
  // backward only (round 6): dy is not a tensor but the raw split-K partial sums of the conv that produces it, [sp_n][rows][C] f32
  // (conv_splitk_finish_kernel's job: summed in split order, rows at or past sp_len[b] zero, rounded to T as that kernel stores it)
  const float* sp_ws;
  const int* sp_len;
  long long sp_slab;
  int sp_n;
};

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, T* __restrict__ sum_out,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     const int* __restrict__ lengths, int64_t rows, int Tlen, int C,
                                                     float eps, int out_mask, int act_in, const LnDrop dp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
      v[i] = Elem<T>::ld4(x + row * C + c);
      if (act_in == PTPP_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = gelu_f(v[i][e]);
      }
      if (dp.in_thresh) v[i] *= drop_mask4(dp.in_seed, (uint64_t)(row * C + c) >> 2, dp.in_thresh, dp.in_inv);
      if (res) v[i] += Elem<T>::ld4(res + row * C + c);
      if (sum_out) Elem<T>::st4(sum_out + row * C + c, v[i]);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  bool keep = true;
  if (out_mask) {
    const int b = (int)(row / Tlen), t = (int)(row % Tlen);
    keep = t < lengths[b];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = keep ? (v[i][e] - mean) * rstd * g[e] + bb[e] : 0.f;
      if (dp.out_thresh) o *= drop_mask4(dp.out_seed, (uint64_t)(row * C + c) >> 2, dp.out_thresh, dp.out_inv);
      Elem<T>::st4(y + row * C + c, o);
    }
  }
}

// Backward: a wave per row (grid-stride); dgamma / dbeta go wave -> block (LDS) -> replicated
// cross-block sums (ptpp_common.h), and red_sum_kernel adds the totals to the caller's buffers.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ xs,
                                                     const T* __restrict__ z, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     T* __restrict__ dsum, T* __restrict__ dz,
                                                     void* scratch, const int* __restrict__ lengths, int64_t rows,
                                                     int Tlen, int C, int out_mask, int act_in, const LnDrop dp) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  f32x4 g[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + lane * 4;
    g[i] = c < C ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t row = wid; row < rows; row += nw) {
    bool keep = true;
    if (out_mask) {
      const int b = (int)(row / Tlen), t = (int)(row % Tlen);
      keep = t < lengths[b];
    }
    const float mu = mean[row], rs = rstd[row];
    f32x4 d[NV], xh[NV];
    unsigned pos[NV];  // sign bits of the norm's input (act_in = relu: the input IS relu's output, positive <=> passed)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + lane * 4;
      d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      pos[i] = 0u;
      if (c < C && keep) {
        if (dp.sp_ws) {
          f32x4 v = *reinterpret_cast<const f32x4*>(dp.sp_ws + row * C + c);
          for (int k = 1; k < dp.sp_n; ++k) v += *reinterpret_cast<const f32x4*>(dp.sp_ws + k * dp.sp_slab + row * C + c);
          if (dp.sp_len) {
            const int b = (int)(row / Tlen), t = (int)(row % Tlen);
            if (t >= min(dp.sp_len[b], Tlen)) v = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32(f32_to_bf16(v[e]));
          }
          d[i] = v;
        } else {
          d[i] = Elem<T>::ld4(dy + row * C + c);
        }
        if (dp.out_thresh) d[i] *= drop_mask4(dp.out_seed, (uint64_t)(row * C + c) >> 2, dp.out_thresh, dp.out_inv);
        const f32x4 xv = Elem<T>::ld4(xs + row * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pos[i] |= (xv[e] > 0.f ? 1u : 0u) << e;
          xh[i][e] = (xv[e] - mu) * rs;
          const float dg = d[i][e] * g[i][e];
          s1 += dg;
          s2 += dg * xh[i][e];
          ag[i][e] += d[i][e] * xh[i][e];
          ab[i][e] += d[i][e];
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + lane * 4;
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs * (d[i][e] * g[i][e] - m1 - xh[i][e] * m2);
        if (dp.add) {  // the residual branch's gradient joins here: rounded first, like a separate add over the stored dsum
          f32x4 r = o;
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = bf16_to_f32(f32_to_bf16(o[e]));
          }
          o = r + Elem<T>::ld4(reinterpret_cast<const T*>(dp.add) + row * C + c);
        }
        if (dsum) Elem<T>::st4(dsum + row * C + c, o);
        if (dz) {
          if constexpr (sizeof(T) == 2) {
            if (dp.dz_stored) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = bf16_to_f32(f32_to_bf16(o[e]));
            }
          }
          o *= dp.dz_scale;
          if (dp.dz_mask) {
            const int b = (int)(row / Tlen), t = (int)(row % Tlen);
            if (t >= lengths[b]) o = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if (dp.in_thresh) o *= drop_mask4(dp.in_seed, (uint64_t)(row * C + c) >> 2, dp.in_thresh, dp.in_inv);
          if (act_in == PTPP_ACT_GELU) {
            const f32x4 zv = Elem<T>::ld4(z + row * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= gelu_grad(zv[e]);
          } else if (act_in == PTPP_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((pos[i] >> e) & 1u) ? o[e] : 0.f;
          }
          Elem<T>::st4(dz + row * C + c, o);
        }
      }
    }
  }
  if (!scratch) return;
  // column sums: combine the block's 4 waves through LDS, then one low-contention atomic per column
  __shared__ float red[2][4][256 * NV];
  __shared__ float tot[2 * 256 * NV];  // [0, C) dgamma, [C, 2C) dbeta
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][w][i * 256 + lane * 4 + e] = ag[i][e];
      red[1][w][i * 256 + lane * 4 + e] = ab[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int k = c >= C, cc = c - k * C;
    tot[c] = red[k][0][cc] + red[k][1][cc] + red[k][2][cc] + red[k][3][cc];
  }
  __syncthreads();
  red_block_add(scratch, tot, 2 * C);
}

LnDrop make_drop(float pin, uint64_t sin_, float pout, uint64_t sout) {
  LnDrop d;
  d.in_thresh = pin > 0.f ? (unsigned)(pin * 65536.f + 0.5f) : 0u;
  d.out_thresh = pout > 0.f ? (unsigned)(pout * 65536.f + 0.5f) : 0u;
  d.in_inv = pin > 0.f ? 1.f / (1.f - d.in_thresh / 65536.f) : 1.f;
  d.out_inv = pout > 0.f ? 1.f / (1.f - d.out_thresh / 65536.f) : 1.f;
  d.in_seed = sin_;
  d.out_seed = sout;
  d.add = nullptr;
  d.dz_scale = 1.f;
  d.dz_mask = 0;
  d.dz_stored = 0;
  d.sp_ws = nullptr;
  d.sp_len = nullptr;
  d.sp_slab = 0;
  d.sp_n = 0;
  return d;
}

template <typename T>
int ln_fwd_dispatch(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* sum_out,
                    float* mean, float* rstd, const int* lengths, int64_t rows, int Tlen, int C, float eps,
                    int out_mask, int act_in, const LnDrop& dp, hipStream_t st) {
  const int nv = (C + 255) / 256;
  const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
#define LN_FWD(NV)                                                                                          \
  hipLaunchKernelGGL((ln_fwd_kernel<T, NV>), grid, blk, 0, st, (const T*)x, (const T*)res, gamma, beta, (T*)y, \
                     (T*)sum_out, mean, rstd, lengths, rows, Tlen, C, eps, out_mask, act_in, dp)
  switch (nv) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    default: ptpp_set_error("layernorm: C=%d > 1024 unsupported", C); return PTPP_ENOTSUP;
  }
#undef LN_FWD
  PTPP_CHECK_LAUNCH("layernorm_fwd");
  return PTPP_OK;
}

template <typename T>
int ln_bwd_dispatch(const void* dy, const void* xs, const void* z, const float* gamma, const float* mean,
                    const float* rstd, void* dsum, void* dz, float* dgamma, float* dbeta, void* scratch,
                    size_t scratch_bytes, const int* lengths, int64_t rows, int Tlen, int C, int out_mask, int act_in,
                    const LnDrop& dp, hipStream_t st) {
  const int nv = (C + 255) / 256;
  int64_t nb = (rows + 3) / 4;
  if (nb > 2048) nb = 2048;
  const bool sums = dgamma || dbeta;
  RedSlot slot{nullptr, 0};
  if (sums) {
    slot = red_take(scratch, scratch_bytes, 2 * C, st);  // (a private arena slice when the finishing launch is deferred)
    if (!slot.ptr) {
      ptpp_set_error("layernorm_bwd: reduction scratch missing or too small");
      return PTPP_EINVAL;
    }
  }
  scratch = slot.ptr;
  const dim3 grid((unsigned)nb), blk(256);
#define LN_BWD(NV)                                                                                              \
  hipLaunchKernelGGL((ln_bwd_kernel<T, NV>), grid, blk, 0, st, (const T*)dy, (const T*)xs, (const T*)z, gamma, mean, \
                     rstd, (T*)dsum, (T*)dz, scratch, lengths, rows, Tlen, C, out_mask, act_in, dp)
  switch (nv) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    default: ptpp_set_error("layernorm: C=%d > 1024 unsupported", C); return PTPP_ENOTSUP;
  }
#undef LN_BWD
  if (sums) red_finish(slot, 2 * C, dgamma, C, dbeta, 1, st);
  PTPP_CHECK_LAUNCH("layernorm_bwd");
  return PTPP_OK;
}

}  // namespace

extern "C" int ptpp_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y,
                                  void* sum_out, float* mean, float* rstd, const int32_t* lengths, int B, int T, int C,
                                  float eps, int out_mask, int act_in, float drop_in_p, uint64_t drop_in_seed,
                                  float drop_out_p, uint64_t drop_out_seed, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null pointer");
  PTPP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0, "layernorm_fwd: bad shape B=%d T=%d C=%d", B, T, C);
  PTPP_CHECK_ARG(!out_mask || lengths, "layernorm_fwd: out_mask needs lengths");
  PTPP_CHECK_ARG(act_in == PTPP_ACT_NONE || act_in == PTPP_ACT_GELU, "layernorm_fwd: act_in must be none or gelu");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  const LnDrop dp = make_drop(drop_in_p, drop_in_seed, drop_out_p, drop_out_seed);
  if (dtype == PTPP_F32)
    return ln_fwd_dispatch<float>(x, res, gamma, beta, y, sum_out, mean, rstd, lengths, rows, T, C, eps, out_mask,
                                  act_in, dp, st);
  if (dtype == PTPP_BF16)
    return ln_fwd_dispatch<bf16_raw>(x, res, gamma, beta, y, sum_out, mean, rstd, lengths, rows, T, C, eps, out_mask,
                                     act_in, dp, st);
  PTPP_CHECK_ARG(false, "layernorm_fwd: bad dtype %d", dtype);
}

static int ln_bwd_entry(const void* dy, const void* xsum, const void* z, const float* gamma, const float* mean, const float* rstd, void* dsum,
                        void* dz, const void* add, float dz_scale, int dz_mask, int dz_stored, float* dgamma, float* dbeta,
                        const int32_t* lengths, int B, int T, int C, int out_mask, int act_in, float drop_in_p, uint64_t drop_in_seed,
                        float drop_out_p, uint64_t drop_out_seed, int dtype, void* scratch, size_t scratch_bytes, void* stream,
                        const float* sp_ws = nullptr, int sp_n = 0, const int32_t* sp_len = nullptr);

extern "C" int ptpp_layernorm_bwd(const void* dy, const void* xsum, const void* z, const float* gamma,
                                  const float* mean, const float* rstd, void* dsum, void* dz, float* dgamma,
                                  float* dbeta, const int32_t* lengths, int B, int T, int C, int out_mask, int act_in,
                                  float drop_in_p, uint64_t drop_in_seed, float drop_out_p, uint64_t drop_out_seed,
                                  int dtype, void* scratch, size_t scratch_bytes, void* stream) {
  return ln_bwd_entry(dy, xsum, z, gamma, mean, rstd, dsum, dz, nullptr, 1.f, 0, 0, dgamma, dbeta, lengths, B, T, C, out_mask, act_in,
                      drop_in_p, drop_in_seed, drop_out_p, drop_out_seed, dtype, scratch, scratch_bytes, stream);
}

extern "C" int ptpp_layernorm_bwd_add(const void* dy, const void* xsum, const void* z, const float* gamma,
                                      const float* mean, const float* rstd, void* dsum, void* dz, const void* add, float dz_scale,
                                      int dz_mask, float* dgamma, float* dbeta, const int32_t* lengths, int B, int T, int C, int out_mask, int act_in,
                                      float drop_in_p, uint64_t drop_in_seed, float drop_out_p, uint64_t drop_out_seed,
                                      int dtype, void* scratch, size_t scratch_bytes, void* stream) {
  return ln_bwd_entry(dy, xsum, z, gamma, mean, rstd, dsum, dz, add, dz_scale, dz_mask, 1, dgamma, dbeta, lengths, B, T, C, out_mask, act_in,
                      drop_in_p, drop_in_seed, drop_out_p, drop_out_seed, dtype, scratch, scratch_bytes, stream);
}

extern "C" int ptpp_layernorm_bwd_add_splitk(const float* partials, int nsplit, const int32_t* partial_lengths, const void* xsum,
                                             const float* gamma, const float* mean, const float* rstd, void* dsum, void* dz, const void* add,
                                             float dz_scale, int dz_mask, float* dgamma, float* dbeta, const int32_t* lengths, int B, int T,
                                             int C, int out_mask, float drop_in_p, uint64_t drop_in_seed, int dtype, void* scratch,
                                             size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(partials && nsplit >= 1, "layernorm_bwd_add_splitk: no partial sums");
  return ln_bwd_entry(nullptr, xsum, nullptr, gamma, mean, rstd, dsum, dz, add, dz_scale, dz_mask, 1, dgamma, dbeta, lengths, B, T, C, out_mask,
                      PTPP_ACT_NONE, drop_in_p, drop_in_seed, 0.f, 0, dtype, scratch, scratch_bytes, stream, partials, nsplit, partial_lengths);
}

static int ln_bwd_entry(const void* dy, const void* xsum, const void* z, const float* gamma, const float* mean, const float* rstd, void* dsum,
                        void* dz, const void* add, float dz_scale, int dz_mask, int dz_stored, float* dgamma, float* dbeta,
                        const int32_t* lengths, int B, int T, int C, int out_mask, int act_in, float drop_in_p, uint64_t drop_in_seed,
                        float drop_out_p, uint64_t drop_out_seed, int dtype, void* scratch, size_t scratch_bytes, void* stream,
                        const float* sp_ws, int sp_n, const int32_t* sp_len) {
  PTPP_CHECK_ARG((dy || sp_ws) && xsum && gamma && mean && rstd && (dsum || dz), "layernorm_bwd: null pointer");
  PTPP_CHECK_ARG(!sp_ws || (sp_n >= 1 && ((uintptr_t)sp_ws & 15) == 0 && dtype == PTPP_BF16), "layernorm_bwd: bad split-K source");
  PTPP_CHECK_ARG(!add || (dsum && add != dsum), "layernorm_bwd: `add` needs a dsum output that is another buffer");
  PTPP_CHECK_ARG(!dz_mask || lengths, "layernorm_bwd: dz_mask needs lengths");
  PTPP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0, "layernorm_bwd: bad shape");
  PTPP_CHECK_ARG(!out_mask || lengths, "layernorm_bwd: out_mask needs lengths");
  PTPP_CHECK_ARG(act_in == PTPP_ACT_NONE || (act_in == PTPP_ACT_GELU && z && dz) || (act_in == PTPP_ACT_RELU && dz),
                 "layernorm_bwd: gelu needs z and dz; relu (the norm's input is the relu output) needs dz");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t rows = (int64_t)B * T;
  LnDrop dp = make_drop(drop_in_p, drop_in_seed, drop_out_p, drop_out_seed);
  dp.add = add;
  dp.dz_scale = dz_scale;
  dp.dz_mask = dz_mask;
  dp.dz_stored = dz_stored;
  dp.sp_ws = sp_ws;
  dp.sp_n = sp_n;
  dp.sp_len = sp_len;
  dp.sp_slab = (long long)rows * C;
  if (dtype == PTPP_F32)
    return ln_bwd_dispatch<float>(dy, xsum, z, gamma, mean, rstd, dsum, dz, dgamma, dbeta, scratch, scratch_bytes, lengths, rows, T, C,
                                  out_mask, act_in, dp, st);
  if (dtype == PTPP_BF16)
    return ln_bwd_dispatch<bf16_raw>(dy, xsum, z, gamma, mean, rstd, dsum, dz, dgamma, dbeta, scratch, scratch_bytes, lengths, rows, T, C,
                                     out_mask, act_in, dp, st);
  PTPP_CHECK_ARG(false, "layernorm_bwd: bad dtype %d", dtype);
}
