// Small fused elementwise / single-channel kernels (all HBM-bound).
#include "ptpp_common.h"
#include <mutex>

namespace {

template <typename T>
__global__ void add3_scale_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                  T* __restrict__ y, float scale, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = Elem<T>::ld4(a + i * 4);
    if (b) v += Elem<T>::ld4(b + i * 4);
    if (c) v += Elem<T>::ld4(c + i * 4);
    Elem<T>::st4(y + i * 4, v * scale);
  }
}

// y (dtype) = x (f32), 4 elements per thread
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    Elem<T>::st4(y + i * 4, *reinterpret_cast<const f32x4*>(x + i * 4));
}

// conv_post (Cout = 1) + tanh: one thread per output sample, the ks*C taps live
// in LDS; x rows are re-read by neighbouring threads out of L1/L2.
template <typename T>
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             float bias, float* __restrict__ y, int T_, int C, int ks) {
  extern __shared__ float wsm[];
  for (int i = threadIdx.x; i < ks * C; i += 256) wsm[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T_) return;
  const int pad = ks / 2;
  const T* xb = x + (int64_t)b * T_ * C;
  float acc = bias;
  for (int j = 0; j < ks; ++j) {
    const int ts = t + j - pad;
    if (ts < 0 || ts >= T_) continue;
    const T* xr = xb + (int64_t)ts * C;
    const float* wr = wsm + j * C;
    for (int c = 0; c < C; c += 4) {
      const f32x4 v = Elem<T>::ld4(xr + c);
      acc += v[0] * wr[c] + v[1] * wr[c + 1] + v[2] * wr[c + 2] + v[3] * wr[c + 3];
    }
  }
  y[(int64_t)b * T_ + t] = tanhf(acc);
}

}  // namespace

extern "C" int ptpp_add3_scale(const void* a, const void* b, const void* c, void* y, float scale, int64_t n, int dtype,
                               void* stream) {
  PTPP_CHECK_ARG(a && y && n > 0 && n % 4 == 0, "add3_scale: bad args");
  const int64_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(add3_scale_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)b,
                       (const float*)c, (float*)y, scale, n4);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(add3_scale_kernel<bf16_raw>, dim3(grid), dim3(256), 0, st, (const bf16_raw*)a,
                       (const bf16_raw*)b, (const bf16_raw*)c, (bf16_raw*)y, scale, n4);
  else if (dtype == PTPP_F16)
    hipLaunchKernelGGL(add3_scale_kernel<f16_raw>, dim3(grid), dim3(256), 0, st, (const f16_raw*)a, (const f16_raw*)b,
                       (const f16_raw*)c, (f16_raw*)y, scale, n4);
  else
    PTPP_CHECK_ARG(false, "add3_scale: bad dtype");
  PTPP_CHECK_LAUNCH("add3_scale");
  return PTPP_OK;
}

extern "C" int ptpp_cast_from_f32(const float* x, void* y, int64_t n, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "cast_from_f32: bad args");
  const int64_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32) hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(grid), dim3(256), 0, st, x, (float*)y, n4);
  else if (dtype == PTPP_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16_raw>, dim3(grid), dim3(256), 0, st, x, (bf16_raw*)y, n4);
  else if (dtype == PTPP_F16) hipLaunchKernelGGL(cast_from_f32_kernel<f16_raw>, dim3(grid), dim3(256), 0, st, x, (f16_raw*)y, n4);
  else PTPP_CHECK_ARG(false, "cast_from_f32: bad dtype");
  PTPP_CHECK_LAUNCH("cast_from_f32");
  return PTPP_OK;
}

extern "C" int ptpp_conv_post_tanh(const void* x, const float* w, float bias, float* y, int B, int T, int C, int ks,
                                   int dtype, void* stream) {
  PTPP_CHECK_ARG(x && w && y && B > 0 && T > 0 && C > 0 && C % 4 == 0 && ks > 0 && (ks & 1),
                 "conv_post_tanh: bad args");
  dim3 grid((T + 255) / 256, B), blk(256);
  const size_t smem = (size_t)ks * C * sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(conv_post_tanh_kernel<float>, grid, blk, smem, st, (const float*)x, w, bias, y, T, C, ks);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL(conv_post_tanh_kernel<bf16_raw>, grid, blk, smem, st, (const bf16_raw*)x, w, bias, y, T, C, ks);
  else if (dtype == PTPP_F16)
    hipLaunchKernelGGL(conv_post_tanh_kernel<f16_raw>, grid, blk, smem, st, (const f16_raw*)x, w, bias, y, T, C, ks);
  else
    PTPP_CHECK_ARG(false, "conv_post_tanh: bad dtype");
  PTPP_CHECK_LAUNCH("conv_post_tanh");
  return PTPP_OK;
}

// ---- stream fork: `waiter` waits for everything enqueued on `signaler` so far ----------------
// One call instead of torch's event-record + wait_event + stream-guard round trip (~25 us of host time
// per weight-gradient launch, ~110 per training step).  Events come from a per-device ring long enough that an
// event is not re-recorded while a wait on it can still be pending.
// ---- zero-phase IIR (forward, time-reverse, forward, time-reverse; zero initial state) --------------------
// torchaudio.functional.filtfilt(x, a, b, clamp=False), which the reference applies to the predicted log-F0 track
// (utils/model.py:187-192: 5th-order Butterworth, fs = 100 Hz, fc = 20 Hz).  The recurrence is sequential in time
// and the tracks are short ((B, 1, T) at 100 frames/s): one thread per row walks it twice in double precision, the
// intermediate in a caller-provided f64 row.  Rows past their length stay untouched (copied).
constexpr int IIR_MAXORD = 8;
struct IirP {
  double b[IIR_MAXORD + 1], a[IIR_MAXORD + 1];
  int order;
};
__global__ void filtfilt_kernel(const float* __restrict__ x, float* __restrict__ y, double* __restrict__ tmp,
                                const int* __restrict__ lengths, IirP f, int rows, int T, int ldx) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int n = lengths ? min(lengths[r], T) : T;
  const float* xr = x + (int64_t)r * ldx;
  float* yr = y + (int64_t)r * ldx;
  double* tr = tmp + (int64_t)r * T;
  double z[IIR_MAXORD];
  // direct form II transposed: y = b0 x + z0; z_i = b_{i+1} x + z_{i+1} - a_{i+1} y
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < IIR_MAXORD; ++i) z[i] = 0.0;
    for (int t = 0; t < n; ++t) {
      const int src = pass == 0 ? t : n - 1 - t;  // second pass runs over the time-reversed first result
      const double xv = pass == 0 ? (double)xr[src] : tr[src];
      const double yv = f.b[0] * xv + z[0];
#pragma unroll
      for (int i = 0; i < IIR_MAXORD; ++i) {
        if (i < f.order) z[i] = f.b[i + 1] * xv + (i + 1 < f.order ? z[i + 1] : 0.0) - f.a[i + 1] * yv;
      }
      if (pass == 0) tr[src] = yv;
      else yr[src] = (float)yv;
    }
  }
  for (int t = n; t < T; ++t) yr[t] = xr[t];
}

namespace {
constexpr int EV_RING = 1024, EV_DEVS = 16;  // (re-recording an event that still has a pending wait is slow on HIP: keep the ring long)
hipEvent_t g_ev[EV_DEVS][EV_RING];
bool g_ev_ready[EV_DEVS];
unsigned g_ev_next[EV_DEVS];
std::mutex g_ev_mu;
}  // namespace

extern "C" int ptpp_stream_wait(void* waiter, void* signaler) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= EV_DEVS) { ptpp_set_error("stream_wait: bad device"); return PTPP_ELAUNCH; }
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lk(g_ev_mu);
    if (!g_ev_ready[dev]) {
      for (int i = 0; i < EV_RING; ++i)
        if (hipEventCreateWithFlags(&g_ev[dev][i], hipEventDisableTiming) != hipSuccess) {
          ptpp_set_error("stream_wait: hipEventCreate failed");
          return PTPP_ELAUNCH;
        }
      g_ev_ready[dev] = true;
    }
    ev = g_ev[dev][g_ev_next[dev]++ % EV_RING];
  }
  if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(signaler)) != hipSuccess ||
      hipStreamWaitEvent(reinterpret_cast<hipStream_t>(waiter), ev, 0) != hipSuccess) {
    ptpp_set_error("stream_wait: %s", hipGetErrorString(hipGetLastError()));
    return PTPP_ELAUNCH;
  }
  return PTPP_OK;
}

// ---------------------------------------------------------------------------
// Dimension-wise mixture-density negative log-likelihood (reference modules/mdn.py:81-175, the `dim_wise` branch):
//   ls = max(log_sigma, ls_min), lp = max(log_pi, lp_min), sigma = exp(ls), d = clamp(target - mu, -5 sigma, 5 sigma)
//   ll_g = -0.5 (d / sigma)^2 - ls - 0.5 log(2 pi) + lp;   loss[b,t,dd] = -logsumexp_g ll_g   (+inf where masked out)
// One thread per (b, t, dd); log_pi / log_sigma / mu are (rows, G, D) f32.  The ~35 tensor ops of the forward and the
// ~70 of its autograd backward (two MDN heads per training step) become one launch each; the backward recomputes ll.
// ---------------------------------------------------------------------------
static __global__ void mdn_nll_fwd_kernel(const float* __restrict__ log_pi, const float* __restrict__ log_sigma,
                                   const float* __restrict__ mu, const float* __restrict__ target,
                                   const unsigned char* __restrict__ mask, float* __restrict__ loss, int64_t n, int G, int D,
                                   float lp_min, float ls_min) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / D;
  const int dd = (int)(i - row * D);
  if (mask && !mask[row]) {
    loss[i] = __builtin_inff();
    return;
  }
  const float tg = target[i];
  const int64_t base = row * G * D + dd;
  float m = -__builtin_inff(), s = 0.f;
  for (int g = 0; g < G; ++g) {
    const float ls = fmaxf(log_sigma[base + (int64_t)g * D], ls_min), lp = fmaxf(log_pi[base + (int64_t)g * D], lp_min);
    const float sg = expf(ls);
    const float d = fmaxf(fminf(tg - mu[base + (int64_t)g * D], 5.f * sg), -5.f * sg);
    const float z = d / sg;
    const float ll = -0.5f * z * z - ls - 0.91893853320467274f + lp;
    if (ll > m) { s = s * expf(m - ll) + 1.f; m = ll; } else { s += expf(ll - m); }
  }
  loss[i] = -(m + logf(s));
}

static __global__ void mdn_nll_bwd_kernel(const float* __restrict__ log_pi, const float* __restrict__ log_sigma,
                                   const float* __restrict__ mu, const float* __restrict__ target,
                                   const unsigned char* __restrict__ mask, const float* __restrict__ loss,
                                   const float* __restrict__ gout, float* __restrict__ d_log_pi,
                                   float* __restrict__ d_log_sigma, float* __restrict__ d_mu, int64_t n, int G, int D,
                                   float lp_min, float ls_min) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / D;
  const int dd = (int)(i - row * D);
  const int64_t base = row * G * D + dd;
  const bool off = mask && !mask[row];
  const float go = off ? 0.f : gout[i], nl = off ? 0.f : loss[i], tg = target[i];
  for (int g = 0; g < G; ++g) {
    const int64_t k = base + (int64_t)g * D;
    float dlp = 0.f, dls = 0.f, dmu = 0.f;
    if (!off) {
      const float lsr = log_sigma[k], lpr = log_pi[k];
      const float ls = fmaxf(lsr, ls_min), lp = fmaxf(lpr, lp_min);
      const float sg = expf(ls);
      const float raw = tg - mu[k];
      const bool clip = fabsf(raw) > 5.f * sg;
      const float d = fmaxf(fminf(raw, 5.f * sg), -5.f * sg);
      const float z = d / sg;
      const float ll = -0.5f * z * z - ls - 0.91893853320467274f + lp;
      const float dll = -go * expf(ll + nl);  // loss = -logsumexp: d loss / d ll_g = -softmax_g
      dlp = lpr >= lp_min ? dll : 0.f;
      dls = lsr >= ls_min ? (clip ? -dll : dll * (z * z - 1.f)) : 0.f;
      dmu = clip ? 0.f : dll * z / sg;
    }
    d_log_pi[k] = dlp;
    d_log_sigma[k] = dls;
    d_mu[k] = dmu;
  }
}

extern "C" int ptpp_mdn_nll_fwd(const float* log_pi, const float* log_sigma, const float* mu, const float* target,
                                const unsigned char* mask, float* loss, int64_t rows, int G, int D, float lp_min, float ls_min,
                                void* stream) {
  PTPP_CHECK_ARG(log_pi && log_sigma && mu && target && loss && rows > 0 && G > 0 && D > 0, "mdn_nll_fwd: bad args");
  const int64_t n = rows * D;
  hipLaunchKernelGGL(mdn_nll_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     log_pi, log_sigma, mu, target, mask, loss, n, G, D, lp_min, ls_min);
  PTPP_CHECK_LAUNCH("mdn_nll_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_mdn_nll_bwd(const float* log_pi, const float* log_sigma, const float* mu, const float* target,
                                const unsigned char* mask, const float* loss, const float* gout, float* d_log_pi,
                                float* d_log_sigma, float* d_mu, int64_t rows, int G, int D, float lp_min, float ls_min,
                                void* stream) {
  PTPP_CHECK_ARG(log_pi && log_sigma && mu && target && loss && gout && d_log_pi && d_log_sigma && d_mu && rows > 0 && G > 0 &&
                     D > 0, "mdn_nll_bwd: bad args");
  const int64_t n = rows * D;
  hipLaunchKernelGGL(mdn_nll_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     log_pi, log_sigma, mu, target, mask, loss, gout, d_log_pi, d_log_sigma, d_mu, n, G, D, lp_min, ls_min);
  PTPP_CHECK_LAUNCH("mdn_nll_bwd");
  return PTPP_OK;
}

// b, a: HOST arrays of order + 1 coefficients (a[0] is divided out); tmp: rows * T doubles of device scratch
extern "C" int ptpp_filtfilt(const float* x, float* y, double* tmp, const int32_t* lengths, const double* b,
                             const double* a, int order, int rows, int T, int ldx, void* stream) {
  PTPP_CHECK_ARG(x && y && tmp && b && a, "filtfilt: null pointer");
  PTPP_CHECK_ARG(order >= 1 && order <= IIR_MAXORD && a[0] != 0.0, "filtfilt: order %d not in 1..%d", order, IIR_MAXORD);
  PTPP_CHECK_ARG(rows > 0 && T > 0 && ldx >= T, "filtfilt: bad shape rows=%d T=%d ld=%d", rows, T, ldx);
  IirP f;
  f.order = order;
  for (int i = 0; i <= IIR_MAXORD; ++i) {
    f.b[i] = i <= order ? b[i] / a[0] : 0.0;
    f.a[i] = i <= order ? a[i] / a[0] : 0.0;
  }
  hipLaunchKernelGGL(filtfilt_kernel, dim3((rows + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), x, y, tmp,
                     lengths, f, rows, T, ldx);
  PTPP_CHECK_LAUNCH("filtfilt");
  return PTPP_OK;
}

// ---- masked L1 mean: the scalar losses of the training step as one launch each way ------------------------------------------------
// loss = sum_i |pred_i - target_i| * mask[i / cols] / denom[0] / scale   (reference models/prompttts_mdn_v2_final/model.py:126,138-170:
// F.l1_loss over masked tensors written as .abs().sum() / n_frames).  The tensor-op form is 6 launches forward and ~8 native
// autograd nodes backward per loss; on the main stream every launch costs ~8 us beyond its kernel time.
// Forward: L1_BLOCKS partial sums, the LAST block to arrive (ticket counter in `scratch`, reset for the next call) adds them in a
// fixed order -- one launch, bit-reproducible.  Backward: d pred = sgn(pred - target) * mask * ((gout * (1 / scale)) / denom), the
// arithmetic of autograd's two DivBackward nodes (a division by a host scalar is a multiplication by its reciprocal there),
// rounded once into pred's dtype.
constexpr int L1_BLOCKS = 128;
template <typename T>
static __global__ __launch_bounds__(256) void l1_masked_mean_fwd_kernel(const T* __restrict__ pred, const float* __restrict__ target,
                                                                      const float* __restrict__ mask, const float* __restrict__ denom,
                                                                      float scale, int64_t n, int cols, float* __restrict__ out,
                                                                      float* __restrict__ scratch) {
  __shared__ float red[4];
  __shared__ int last;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)L1_BLOCKS * 256) {
    const float d = fabsf(Elem<T>::ld(pred + i) - target[i]);
    s += mask ? d * mask[i / cols] : d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();
    const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(scratch), 1u);
    last = t == (unsigned)L1_BLOCKS - 1u;
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    __threadfence();
    const volatile float* part = scratch + 1;
    float v = part[threadIdx.x] + part[threadIdx.x + 64];
    v = wave_sum(v);
    if (threadIdx.x == 0) {
      out[0] = (v / denom[0]) * (1.f / scale);  // (torch divides by a host scalar as a multiplication by its f32 reciprocal)
      *reinterpret_cast<unsigned*>(scratch) = 0u;
    }
  }
}
template <typename T>
static __global__ __launch_bounds__(256) void l1_masked_mean_bwd_kernel(const T* __restrict__ pred, const float* __restrict__ target,
                                                                      const float* __restrict__ mask, const float* __restrict__ denom,
                                                                      const float* __restrict__ gout, float scale, int64_t n, int cols,
                                                                      T* __restrict__ dpred) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float coef = (gout[0] * (1.f / scale)) / denom[0];  // DivBackward by the host scalar (x reciprocal), then by the tensor
  const float d = Elem<T>::ld(pred + i) - target[i];
  float g = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
  if (mask) g *= mask[i / cols];
  Elem<T>::st(dpred + i, g);
}

extern "C" int64_t ptpp_l1_scratch_bytes(void) { return (int64_t)(1 + L1_BLOCKS) * 4; }

extern "C" int ptpp_l1_masked_mean_fwd(const void* pred, const float* target, const float* mask, const float* denom, float scale,
                                       int64_t rows, int cols, int dtype, float* out, void* scratch, void* stream) {
  PTPP_CHECK_ARG(pred && target && denom && out && scratch && rows > 0 && cols > 0 && scale != 0.f, "l1_masked_mean_fwd: bad args");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "l1_masked_mean_fwd: dtype %d (f32 / bf16)", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = rows * cols;
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(l1_masked_mean_fwd_kernel<float>, dim3(L1_BLOCKS), dim3(256), 0, st, reinterpret_cast<const float*>(pred), target,
                       mask, denom, scale, n, cols, out, reinterpret_cast<float*>(scratch));
  else
    hipLaunchKernelGGL(l1_masked_mean_fwd_kernel<bf16_raw>, dim3(L1_BLOCKS), dim3(256), 0, st, reinterpret_cast<const bf16_raw*>(pred),
                       target, mask, denom, scale, n, cols, out, reinterpret_cast<float*>(scratch));
  PTPP_CHECK_LAUNCH("l1_masked_mean_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_l1_masked_mean_bwd(const void* pred, const float* target, const float* mask, const float* denom, const float* gout,
                                       float scale, int64_t rows, int cols, int dtype, void* dpred, void* stream) {
  PTPP_CHECK_ARG(pred && target && denom && gout && dpred && rows > 0 && cols > 0 && scale != 0.f, "l1_masked_mean_bwd: bad args");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "l1_masked_mean_bwd: dtype %d (f32 / bf16)", dtype);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t n = rows * cols;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(l1_masked_mean_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, reinterpret_cast<const float*>(pred), target, mask,
                       denom, gout, scale, n, cols, reinterpret_cast<float*>(dpred));
  else
    hipLaunchKernelGGL(l1_masked_mean_bwd_kernel<bf16_raw>, dim3(nb), dim3(256), 0, st, reinterpret_cast<const bf16_raw*>(pred), target,
                       mask, denom, gout, scale, n, cols, reinterpret_cast<bf16_raw*>(dpred));
  PTPP_CHECK_LAUNCH("l1_masked_mean_bwd");
  return PTPP_OK;
}

