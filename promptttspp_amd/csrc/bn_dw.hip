// BatchNorm (training-mode batch statistics) + activation on channels-last rows, GLU,
// depthwise Conv1d and the 3x3/stride-2 im2col of the GST reference encoder.
// These serve the Conformer convolution module (esp/conformer/convolution.py:58-85)
// and the reference encoder's Conv2d+BatchNorm2d+ReLU stack
// (modules/reference_encoder.py:65-81) -- all HBM-bound row-streaming kernels:
// a thread owns 4 consecutive channels (8/16-byte vectors), per-channel reductions
// go through LDS and the replicated cross-block sums of ptpp_common.h.
#include <stdlib.h>

#include "ptpp_common.h"

namespace {

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

// act codes: 0 none, 1 relu, 3 swish
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == PTPP_ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == PTPP_ACT_SWISH) return z * sigm(z);
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == PTPP_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == PTPP_ACT_SWISH) { const float s = sigm(z); return s * (1.f + z * (1.f - s)); }
  return 1.f;
}

// block geometry for column reductions: 256 threads = RG row-groups x CV vector-columns
struct ColGeom { int cv, rg; };
__device__ __forceinline__ ColGeom col_geom(int C) {
  ColGeom g; g.cv = C >> 2; g.rg = 256 / g.cv; return g;
}

// out[c] = sum_r f(x[r,c]) ; f = x (mean == NULL) or (x-mean)^2.  Block totals -> replicated sums (red_block_add) -> red_sum_kernel.
template <typename T>
__global__ __launch_bounds__(256) void col_reduce_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                         void* scratch, int64_t rows, int C, int rows_per_block) {
  const ColGeom g = col_geom(C);
  const int cvi = threadIdx.x % g.cv, rgi = threadIdx.x / g.cv;
  const int c = cvi * 4;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (rgi < g.rg) {
    const f32x4 mu = mean ? *reinterpret_cast<const f32x4*>(mean + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int64_t r = r0 + rgi; r < r1; r += g.rg) {
      f32x4 v = Elem<T>::ld4(x + r * C + c);
      if (mean) { v -= mu; v *= v; }
      acc += v;
    }
  }
  __shared__ f32x4 red[256];
  __shared__ float tot[1024];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rgi == 0) {
    for (int k = 1; k < g.rg; ++k) acc += red[k * g.cv + cvi];
#pragma unroll
    for (int e = 0; e < 4; ++e) tot[c + e] = acc[e];
  }
  __syncthreads();
  red_block_add(scratch, tot, C);
}

// finishing launches of the training-mode statistics: replicas -> mean (+ running mean), then
// replicas -> biased variance -> rstd (+ running variance, unbiased); the scratch is zeroed again
__global__ void bn_stats_finish_kernel(float* __restrict__ scratch, int C, int stage, float inv_rows, float unbias,
                                       float momentum, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                       float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float v[PTPP_RED_NREP];
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) v[r] = scratch[(size_t)r * C + c];
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) {
    s += v[r];
    scratch[(size_t)r * C + c] = 0.f;
  }
  s *= inv_rows;
  if (stage == 0) {
    mean[c] = s;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * s;
  } else {
    rstd[c] = rsqrtf(s + eps);
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (s * unbias);
  }
}

// y = act(gamma * (x - mean) * rstd + beta): a thread keeps its 4 channels' constants in registers
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y, int64_t rows,
                                                         int C, int act, int rows_per_block) {
  const ColGeom g = col_geom(C);
  const int cvi = threadIdx.x % g.cv, rgi = threadIdx.x / g.cv;
  if (rgi >= g.rg) return;
  const int c = cvi * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
#pragma unroll 4
  for (int64_t r = r0 + rgi; r < r1; r += g.rg) {
    const f32x4 v = Elem<T>::ld4(x + r * C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_fwd(ga[e] * (v[e] - mu[e]) * rs[e] + be[e], act);
    Elem<T>::st4(y + r * C + c, o);
  }
}

// the same for channel counts the row geometry does not cover (C/4 does not divide 256)
template <typename T>
__global__ void bn_act_fwd_generic_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, T* __restrict__ y, int C, int64_t nvec, int act) {
  const int cv = C >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    const f32x4 v = Elem<T>::ld4(x + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_fwd(gamma[c + e] * (v[e] - mean[c + e]) * rstd[c + e] + beta[c + e], act);
    Elem<T>::st4(y + i * 4, o);
  }
}

// pass 1 of the backward: sums[0][c] = sum g, sums[1][c] = sum g * xhat, g = dy * act'(z)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            void* scratch, int64_t rows, int C, int act,
                                                            int rows_per_block) {
  const ColGeom g = col_geom(C);
  const int cvi = threadIdx.x % g.cv, rgi = threadIdx.x / g.cv;
  const int c = cvi * 4;
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (rgi < g.rg) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll 4
    for (int64_t r = r0 + rgi; r < r1; r += g.rg) {
      const f32x4 v = Elem<T>::ld4(x + r * C + c), d = Elem<T>::ld4(dy + r * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[e] - mu[e]) * rs[e];
        const float gg = d[e] * act_grad(ga[e] * xh + be[e], act);
        a0[e] += gg;
        a1[e] += gg * xh;
      }
    }
  }
  __shared__ f32x4 red[2][256];
  __shared__ float tot[2048];
  red[0][threadIdx.x] = a0;
  red[1][threadIdx.x] = a1;
  __syncthreads();
  if (rgi == 0) {
    for (int k = 1; k < g.rg; ++k) { a0 += red[0][k * g.cv + cvi]; a1 += red[1][k * g.cv + cvi]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { tot[c + e] = a0[e]; tot[C + c + e] = a1[e]; }
  }
  __syncthreads();
  red_block_add(scratch, tot, 2 * C);
}

// pass 2: dx = gamma * rstd * (g - sum_g / N - xhat * sum_gx / N)   (train) ; = gamma * rstd * g (eval)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ sums, T* __restrict__ dx, int64_t rows,
                                                           int C, int act, float inv_n, int train, int rows_per_block) {
  const ColGeom g = col_geom(C);
  const int cvi = threadIdx.x % g.cv, rgi = threadIdx.x / g.cv;
  if (rgi >= g.rg) return;
  const int c = cvi * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  f32x4 sg = f32x4{0.f, 0.f, 0.f, 0.f}, sgx = sg;
  if (train) {
    sg = *reinterpret_cast<const f32x4*>(sums + c) * inv_n;
    sgx = *reinterpret_cast<const f32x4*>(sums + C + c) * inv_n;
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
#pragma unroll 4
  for (int64_t r = r0 + rgi; r < r1; r += g.rg) {
    const f32x4 v = Elem<T>::ld4(x + r * C + c), d = Elem<T>::ld4(dy + r * C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu[e]) * rs[e];
      const float gg = d[e] * act_grad(ga[e] * xh + be[e], act);
      o[e] = ga[e] * rs[e] * (gg - (sg[e] + xh * sgx[e]));
    }
    Elem<T>::st4(dx + r * C + c, o);
  }
}

// u = a * sigmoid(g), h = [a | g] (rows, 2C)
template <typename T>
__global__ void glu_fwd_kernel(const T* __restrict__ h, T* __restrict__ u, int C, int64_t nvec) {
  const int cv = C >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cv;
    const int c = (int)(i % cv) * 4;
    const f32x4 a = Elem<T>::ld4(h + row * 2 * C + c), g = Elem<T>::ld4(h + row * 2 * C + C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] * sigm(g[e]);
    Elem<T>::st4(u + i * 4, o);
  }
}
template <typename T>
__global__ void glu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ du, T* __restrict__ dh, int C, int64_t nvec,
                               const int* __restrict__ lengths, int Tlen) {
  const int cv = C >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cv;
    const int c = (int)(i % cv) * 4;
    if (lengths && (int)(row % Tlen) >= lengths[row / Tlen]) {  // (rows past an utterance's end: zero gradient)
      Elem<T>::st4(dh + row * 2 * C + c, f32x4{0.f, 0.f, 0.f, 0.f});
      Elem<T>::st4(dh + row * 2 * C + C + c, f32x4{0.f, 0.f, 0.f, 0.f});
      continue;
    }
    const f32x4 a = Elem<T>::ld4(h + row * 2 * C + c), g = Elem<T>::ld4(h + row * 2 * C + C + c), d = Elem<T>::ld4(du + i * 4);
    f32x4 da, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = sigm(g[e]);
      da[e] = d[e] * s;
      dg[e] = d[e] * a[e] * s * (1.f - s);
    }
    Elem<T>::st4(dh + row * 2 * C + c, da);
    Elem<T>::st4(dh + row * 2 * C + C + c, dg);
  }
}

// depthwise conv over time: y[b,t,c] = [t < len] * (bias[c] + sum_j w[c][j] * u[b, t + j - pad, c])
// flip = 1 computes the data gradient: du[b,t,c] = sum_j w[c][j] * dym[b, t - j + pad, c], dym = dy masked
template <typename T, int KS>
__global__ void dwconv_kernel(const T* __restrict__ u, const float* __restrict__ w, const float* __restrict__ bias,
                              T* __restrict__ y, const int* __restrict__ lengths, int Tlen, int C, int64_t nvec, int flip) {
  const int cv = C >> 2;
  constexpr int PAD = KS / 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cv;
    const int c = (int)(i % cv) * 4;
    const int b = (int)(row / Tlen), t = (int)(row % Tlen);
    const int len = lengths ? min(lengths[b], Tlen) : Tlen;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!flip && bias) acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int ts = flip ? t - j + PAD : t + j - PAD;
      const int lim = flip ? len : Tlen;  // the gradient only comes from unmasked output rows
      if (ts < 0 || ts >= lim) continue;
      const f32x4 v = Elem<T>::ld4(u + ((int64_t)b * Tlen + ts) * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += w[(c + e) * KS + j] * v[e];
    }
    if (!flip && t >= len) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    Elem<T>::st4(y + i * 4, acc);
  }
}

// dw[c][j] += sum_{b,t<len} dy[b,t,c] * u[b, t + j - pad, c] ; dbias[c] += sum dy
// Every block ends with (KS + 1) C atomics into the same ~60 cache lines (~50 ns per visit and line).
template <typename T, int KS>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ u, const T* __restrict__ dy,
                                                           float* __restrict__ dw, float* __restrict__ dbias,
                                                           const int* __restrict__ lengths, int Tlen, int C, int64_t rows,
                                                           int rows_per_block) {
  const ColGeom g = col_geom(C);
  const int cvi = threadIdx.x % g.cv, rgi = threadIdx.x / g.cv;
  const int c = cvi * 4;
  constexpr int PAD = KS / 2;
  f32x4 aw[KS], ab = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KS; ++j) aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (rgi < g.rg)
    for (int64_t r = r0 + rgi; r < r1; r += g.rg) {
      const int b = (int)(r / Tlen), t = (int)(r % Tlen);
      const int len = lengths ? min(lengths[b], Tlen) : Tlen;
      if (t >= len) continue;
      const f32x4 d = Elem<T>::ld4(dy + r * C + c);
      ab += d;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const int ts = t + j - PAD;
        if (ts < 0 || ts >= Tlen) continue;
        aw[j] += d * Elem<T>::ld4(u + ((int64_t)b * Tlen + ts) * C + c);
      }
    }
  __shared__ f32x4 red[256];
#pragma unroll
  for (int j = 0; j <= KS; ++j) {
    const f32x4 v = j < KS ? aw[j < KS ? j : 0] : ab;
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    if (rgi == 0) {
      f32x4 s = v;
      for (int k = 1; k < g.rg; ++k) s += red[k * g.cv + cvi];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j < KS) atomicAdd(dw + (c + e) * KS + j, s[e]);
        else if (dbias) atomicAdd(dbias + c + e, s[e]);
      }
    }
  }
}

// Short inputs (phone level: a few thousand rows).  The kernel above spent 68 us on 3 800 rows: every row of a thread's loop
// is a chain of two dependent memory latencies (lengths[row / T] -> the row's loads) behind a 64-bit division.  Here a block
// owns RPT * NT / (C/4) consecutive rows of ONE utterance (utterance index and length are block-uniform, no division), a
// thread issues all its RPT * (KS + 1) loads at once (512 threads x 4 rows: 160 registers, no scratch).
template <typename T, int KS, int NT, int RPT>
__global__ __launch_bounds__(NT) void dwconv_wgrad_short_kernel(const T* __restrict__ u, const T* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ dbias,
                                                                const int* __restrict__ lengths, int Tlen, int C, int nchunk) {
  const int cv = C >> 2, rg = NT / cv;
  const int cvi = threadIdx.x % cv, rgi = threadIdx.x / cv;
  const int c = cvi * 4;
  constexpr int PAD = KS / 2;
  constexpr int ITER = 4;  // row windows per block: 4 x fewer blocks queue at the accumulators' cache lines
  const int b = blockIdx.x / nchunk, tb = (blockIdx.x % nchunk) * (rg * RPT * ITER);
  const int len = lengths ? min(lengths[b], Tlen) : Tlen;
  if (tb >= len) return;  // block-uniform
  const T* ub = u + (int64_t)b * Tlen * C + c;
  const T* dyb = dy + (int64_t)b * Tlen * C + c;
  f32x4 aw[KS], ab = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < KS; ++j) aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < ITER; ++it) {
  const int t0 = tb + it * (rg * RPT);
  if (t0 >= len) break;
  f32x4 d[RPT], uu[RPT][KS];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int t = t0 + rgi + k * rg;
    const bool valid = t < len;
    d[k] = valid ? Elem<T>::ld4(dyb + (int64_t)t * C) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int ts = t + j - PAD;
      uu[k][j] = (valid && ts >= 0 && ts < Tlen) ? Elem<T>::ld4(ub + (int64_t)ts * C) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    ab += d[k];
#pragma unroll
    for (int j = 0; j < KS; ++j) aw[j] += d[k] * uu[k][j];
  }
  }
  __shared__ f32x4 red[NT];
#pragma unroll
  for (int j = 0; j <= KS; ++j) {
    const f32x4 v = j < KS ? aw[j < KS ? j : 0] : ab;
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    if (rgi == 0) {
      f32x4 s = v;
      for (int k = 1; k < rg; ++k) s += red[k * cv + cvi];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j < KS) atomicAdd(dw + (c + e) * KS + j, s[e]);
        else if (dbias) atomicAdd(dbias + c + e, s[e]);
      }
    }
  }
}

// im2col for Conv2d(k=3, stride=2, pad=1) on channels-last (B, H, W, C):
// col[(b, ho, wo)][(kh*3 + kw)*C + c] = x[b, 2ho + kh - 1, 2wo + kw - 1, c]  (0 outside)
template <typename T>
__global__ void im2col3x3s2_kernel(const T* __restrict__ x, T* __restrict__ col, int H, int W, int C, int Ho, int Wo,
                                   int64_t nvec) {
  const int cv = C >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    int64_t q = i / cv;
    const int k = (int)(q % 9); q /= 9;
    const int wo = (int)(q % Wo); q /= Wo;
    const int ho = (int)(q % Ho);
    const int b = (int)(q / Ho);
    const int hi = 2 * ho + k / 3 - 1, wi = 2 * wo + k % 3 - 1;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = Elem<T>::ld4(x + (((int64_t)b * H + hi) * W + wi) * C + c);
    Elem<T>::st4(col + i * 4, v);
  }
}

// the first layer of the reference encoder: x has ONE channel; col gets the 8-channel granule of the GEMM operand per tap,
// [x, 0 x 7] -- what im2col3x3s2 makes of x zero-padded to 8 channels, without materialising the padded input
template <typename T>
__global__ void im2col3x3s2_c1_kernel(const T* __restrict__ x, T* __restrict__ col, int H, int W, int Ho, int Wo, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t q = i;
    const int k = (int)(q % 9); q /= 9;
    const int wo = (int)(q % Wo); q /= Wo;
    const int ho = (int)(q % Ho);
    const int b = (int)(q / Ho);
    const int hi = 2 * ho + k / 3 - 1, wi = 2 * wo + k % 3 - 1;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v[0] = Elem<T>::ld(x + ((int64_t)b * H + hi) * W + wi);
    Elem<T>::st4(col + i * 8, v);
    Elem<T>::st4(col + i * 8 + 4, f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

// col2im (gather form): dx[b,hi,wi,c] = sum over (ho,kh),(wo,kw) with 2ho+kh-1 == hi, 2wo+kw-1 == wi of dcol[...]
template <typename T>
__global__ void col2im3x3s2_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int H, int W, int C, int Ho, int Wo,
                                   int64_t nvec) {
  const int cv = C >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    int64_t q = i / cv;
    const int wi = (int)(q % W); q /= W;
    const int hi = (int)(q % H);
    const int b = (int)(q / H);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h2 = hi + 1 - kh;
      if (h2 < 0 || (h2 & 1)) continue;
      const int ho = h2 >> 1;
      if (ho >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w2 = wi + 1 - kw;
        if (w2 < 0 || (w2 & 1)) continue;
        const int wo = w2 >> 1;
        if (wo >= Wo) continue;
        acc += Elem<T>::ld4(dcol + ((((int64_t)b * Ho + ho) * Wo + wo) * 9 + kh * 3 + kw) * C + c);
      }
    }
    Elem<T>::st4(dx + i * 4, acc);
  }
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
inline bool cgeom_ok(int C) { return C > 0 && C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0; }

}  // namespace

#define DISPATCH_T(dtype, name, ...)                         \
  if (dtype == PTPP_F32) { using T = float; __VA_ARGS__; }   \
  else if (dtype == PTPP_BF16) { using T = bf16_raw; __VA_ARGS__; } \
  else { ptpp_set_error("%s: bad dtype %d", name, dtype); return PTPP_EINVAL; }

// rows per block: a thread handles `per_thread` rows
inline int stream_rpb(int C, int per_thread) { return (256 / (C / 4)) * per_thread; }
// Column sums that feed the DATA path (BatchNorm batch statistics, the two sums of its backward).  Default: ~rows / 32 blocks add
// their totals into the 32 replicas of the reduction scratch by f32 atomics in arrival order (repeated runs of one step fall
// into classes ~3e-3 apart in the style embedding).  PTPP_BN_DET=1: at most one block per replica, so
// every replica has ONE writer and the finishing launch adds them in a fixed order -- bit-reproducible, but the reference
// encoder's 600 k-row reductions then run on 32 blocks: 15.76 -> 17.33 ms per training step (same box), hence opt-in.
inline void det_grid(int64_t rows, int C, unsigned* nb, int* rpb) {
  const int rpb0 = stream_rpb(C, 8);
  int64_t n = (rows + rpb0 - 1) / rpb0;
  const char* det = getenv("PTPP_BN_DET");
  if (n > PTPP_RED_NREP && det && det[0] == '1') n = PTPP_RED_NREP;
  *rpb = (int)((rows + n - 1) / n);
  *nb = (unsigned)((rows + *rpb - 1) / *rpb);
}

extern "C" int ptpp_col_reduce(const void* x, const float* mean, float* out, int64_t rows, int C, int dtype,
                               void* scratch, size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(x && out && rows > 0 && cgeom_ok(C), "col_reduce: bad args (C=%d)", C);
  PTPP_CHECK_ARG(red_scratch_ok(scratch, scratch_bytes, C), "col_reduce: reduction scratch missing or too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rpb;
  unsigned nb;
  det_grid(rows, C, &nb, &rpb);
  DISPATCH_T(dtype, "col_reduce",
             hipLaunchKernelGGL(col_reduce_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)x, mean, scratch, rows, C, rpb));
  red_sum_launch(scratch, C, out, C, nullptr, 0, st);
  PTPP_CHECK_LAUNCH("col_reduce");
  return PTPP_OK;
}

extern "C" int ptpp_bn_stats(const void* x, int64_t rows, int C, float momentum, float eps, float* running_mean,
                             float* running_var, float* mean, float* rstd, int dtype, void* scratch, size_t scratch_bytes,
                             void* stream) {
  PTPP_CHECK_ARG(x && mean && rstd && rows > 0 && cgeom_ok(C), "bn_stats: bad args (C=%d)", C);
  PTPP_CHECK_ARG(red_scratch_ok(scratch, scratch_bytes, C), "bn_stats: reduction scratch missing or too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rpb;
  unsigned nb;
  det_grid(rows, C, &nb, &rpb);
  const float inv = 1.0f / (float)rows, unbias = (float)rows / (float)(rows > 1 ? rows - 1 : 1);
  float* sc = reinterpret_cast<float*>(scratch);
  const dim3 fg((C + 63) / 64), fb(64);
  DISPATCH_T(dtype, "bn_stats",
             hipLaunchKernelGGL(col_reduce_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)x, (const float*)nullptr, scratch,
                                rows, C, rpb);
             hipLaunchKernelGGL(bn_stats_finish_kernel, fg, fb, 0, st, sc, C, 0, inv, unbias, momentum, eps, mean, rstd,
                                running_mean, running_var);
             hipLaunchKernelGGL(col_reduce_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)x, (const float*)mean, scratch,
                                rows, C, rpb);
             hipLaunchKernelGGL(bn_stats_finish_kernel, fg, fb, 0, st, sc, C, 1, inv, unbias, momentum, eps, mean, rstd,
                                running_mean, running_var));
  PTPP_CHECK_LAUNCH("bn_stats");
  return PTPP_OK;
}

extern "C" int ptpp_bn_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               void* y, int64_t rows, int C, int act, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && mean && rstd && gamma && beta && y && rows > 0 && C > 0 && C % 4 == 0, "bn_act_fwd: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (cgeom_ok(C)) {
    const int rpb = stream_rpb(C, 8);
    const unsigned nb = (unsigned)((rows + rpb - 1) / rpb);
    DISPATCH_T(dtype, "bn_act_fwd",
               hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)x, mean, rstd, gamma, beta,
                                  (T*)y, rows, C, act, rpb));
  } else {
    const int64_t nvec = rows * C / 4;
    DISPATCH_T(dtype, "bn_act_fwd",
               hipLaunchKernelGGL(bn_act_fwd_generic_kernel<T>, dim3(grid_for(nvec)), dim3(256), 0, st, (const T*)x, mean,
                                  rstd, gamma, beta, (T*)y, C, nvec, act));
  }
  PTPP_CHECK_LAUNCH("bn_act_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_bn_act_bwd(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, float* sums, void* dx, int64_t rows, int C, int act, int train, int dtype,
                               void* scratch, size_t scratch_bytes, void* stream) {
  return ptpp_bn_act_bwd_acc(x, dy, mean, rstd, gamma, beta, sums, nullptr, nullptr, dx, rows, C, act, train, dtype, scratch, scratch_bytes,
                             stream);
}

extern "C" int ptpp_bn_act_bwd_acc(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, float* sums, float* dbeta_acc, float* dgamma_acc, void* dx, int64_t rows, int C,
                                   int act, int train, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(x && dy && mean && rstd && gamma && beta && sums && dx && rows > 0 && cgeom_ok(C), "bn_act_bwd: bad args");
  PTPP_CHECK_ARG(red_scratch_ok(scratch, scratch_bytes, 2 * C), "bn_act_bwd: reduction scratch missing or too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rpb;
  unsigned nb;
  det_grid(rows, C, &nb, &rpb);
  const int rpb2 = stream_rpb(C, 8);
  const unsigned nb2 = (unsigned)((rows + rpb2 - 1) / rpb2);
  DISPATCH_T(dtype, "bn_act_bwd",
             hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(nb), dim3(256), 0, st, (const T*)x, (const T*)dy, mean, rstd,
                                gamma, beta, scratch, rows, C, act, rpb);
             if (dbeta_acc || dgamma_acc)
               hipLaunchKernelGGL(red_sum_both_kernel, dim3((2 * C + 63) / 64), dim3(64), 0, st, reinterpret_cast<float*>(scratch), 2 * C, sums,
                                  dbeta_acc, C, dgamma_acc);
             else red_sum_launch(scratch, 2 * C, sums, 2 * C, nullptr, 0, st);
             hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(nb2), dim3(256), 0, st, (const T*)x, (const T*)dy, mean, rstd,
                                gamma, beta, sums, (T*)dx, rows, C, act, 1.0f / (float)rows, train, rpb2));
  PTPP_CHECK_LAUNCH("bn_act_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_glu_fwd(const void* h, void* u, int64_t rows, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(h && u && rows > 0 && C > 0 && C % 4 == 0, "glu_fwd: bad args");
  const int64_t nvec = rows * C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, "glu_fwd", hipLaunchKernelGGL(glu_fwd_kernel<T>, dim3(grid_for(nvec)), dim3(256), 0, st, (const T*)h, (T*)u, C, nvec));
  PTPP_CHECK_LAUNCH("glu_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_glu_bwd(const void* h, const void* du, void* dh, int64_t rows, int C, int dtype, void* stream) {
  return ptpp_glu_bwd_masked(h, du, dh, nullptr, 1, (int)rows, C, dtype, stream);
}

extern "C" int ptpp_glu_bwd_masked(const void* h, const void* du, void* dh, const int32_t* lengths, int B, int T_, int C, int dtype,
                                   void* stream) {
  PTPP_CHECK_ARG(h && du && dh && B > 0 && T_ > 0 && C > 0 && C % 4 == 0, "glu_bwd: bad args");
  const int64_t nvec = (int64_t)B * T_ * C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, "glu_bwd",
             hipLaunchKernelGGL(glu_bwd_kernel<T>, dim3(grid_for(nvec)), dim3(256), 0, st, (const T*)h, (const T*)du, (T*)dh, C, nvec,
                                lengths, T_));
  PTPP_CHECK_LAUNCH("glu_bwd");
  return PTPP_OK;
}

#define DW_LAUNCH(KS_)                                                                                                   \
  hipLaunchKernelGGL((dwconv_kernel<T, KS_>), dim3(grid_for(nvec)), dim3(256), 0, st, (const T*)u, w, bias, (T*)y, lengths, \
                     T_, C, nvec, flip)

extern "C" int ptpp_dwconv1d(const void* u, const float* w, const float* bias, void* y, const int32_t* lengths, int B, int T_,
                             int C, int ks, int flip, int dtype, void* stream) {
  PTPP_CHECK_ARG(u && w && y && B > 0 && T_ > 0 && C > 0 && C % 4 == 0, "dwconv1d: bad args");
  PTPP_CHECK_ARG(ks == 7 || ks == 15 || ks == 31, "dwconv1d: kernel size %d not built (7, 15, 31)", ks);
  const int64_t nvec = (int64_t)B * T_ * C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, "dwconv1d", if (ks == 7) { DW_LAUNCH(7); } else if (ks == 15) { DW_LAUNCH(15); } else { DW_LAUNCH(31); });
  PTPP_CHECK_LAUNCH("dwconv1d");
  return PTPP_OK;
}

#define DWW_LAUNCH(KS_)                                                                                              \
  hipLaunchKernelGGL((dwconv_wgrad_kernel<T, KS_>), dim3(nb), dim3(256), 0, st, (const T*)u, (const T*)dy, dw, dbias, \
                     lengths, T_, C, rows, rpb)
#define DWW_SHORT(KS_)                                                                                                        \
  hipLaunchKernelGGL((dwconv_wgrad_short_kernel<T, KS_, 512, 4>), dim3((unsigned)(B * nchunk)), dim3(512), 0, st, (const T*)u, \
                     (const T*)dy, dw, dbias, lengths, T_, C, nchunk)

extern "C" int ptpp_dwconv1d_wgrad(const void* u, const void* dy, float* dw, float* dbias, const int32_t* lengths, int B,
                                   int T_, int C, int ks, int dtype, void* stream) {
  PTPP_CHECK_ARG(u && dy && dw && B > 0 && T_ > 0 && cgeom_ok(C), "dwconv1d_wgrad: bad args");
  PTPP_CHECK_ARG(ks == 7 || ks == 15 || ks == 31, "dwconv1d_wgrad: kernel size %d not built", ks);
  const int64_t rows = (int64_t)B * T_;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ks == 7 && rows <= 65536) {  // (the register window of the short kernel is built for the Conformer's k = 7)
    const int rpb4 = 4 * 4 * (512 / (C >> 2));  // rows per block (ITER x RPT x row groups of a 512-thread block)
    const int nchunk = (T_ + rpb4 - 1) / rpb4;
    DISPATCH_T(dtype, "dwconv1d_wgrad", DWW_SHORT(7));
    PTPP_CHECK_LAUNCH("dwconv1d_wgrad");
    return PTPP_OK;
  }
  // 32 rows per block keeps >100 blocks in flight for inputs of a few thousand rows
  const int rpb = rows > 65536 ? 128 : 32;
  const unsigned nb = (unsigned)((rows + rpb - 1) / rpb);
  DISPATCH_T(dtype, "dwconv1d_wgrad", if (ks == 7) { DWW_LAUNCH(7); } else if (ks == 15) { DWW_LAUNCH(15); } else { DWW_LAUNCH(31); });
  PTPP_CHECK_LAUNCH("dwconv1d_wgrad");
  return PTPP_OK;
}

extern "C" int ptpp_im2col3x3s2(const void* x, void* col, int B, int H, int W, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "im2col3x3s2: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t nvec = (int64_t)B * Ho * Wo * 9 * C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, "im2col3x3s2",
             hipLaunchKernelGGL(im2col3x3s2_kernel<T>, dim3(grid_for(nvec)), dim3(256), 0, st, (const T*)x, (T*)col, H, W, C, Ho, Wo, nvec));
  PTPP_CHECK_LAUNCH("im2col3x3s2");
  return PTPP_OK;
}

extern "C" int ptpp_im2col3x3s2_c1(const void* x, void* col, int B, int H, int W, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0, "im2col3x3s2_c1: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t n = (int64_t)B * Ho * Wo * 9;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, "im2col3x3s2_c1",
             hipLaunchKernelGGL(im2col3x3s2_c1_kernel<T>, dim3(grid_for(n)), dim3(256), 0, st, (const T*)x, (T*)col, H, W, Ho, Wo, n));
  PTPP_CHECK_LAUNCH("im2col3x3s2_c1");
  return PTPP_OK;
}

extern "C" int ptpp_col2im3x3s2(const void* dcol, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "col2im3x3s2: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t nvec = (int64_t)B * H * W * C / 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DISPATCH_T(dtype, "col2im3x3s2",
             hipLaunchKernelGGL(col2im3x3s2_kernel<T>, dim3(grid_for(nvec)), dim3(256), 0, st, (const T*)dcol, (T*)dx, H, W, C, Ho, Wo, nvec));
  PTPP_CHECK_LAUNCH("col2im3x3s2");
  return PTPP_OK;
}
