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
  else
    PTPP_CHECK_ARG(false, "add3_scale: bad dtype");
  PTPP_CHECK_LAUNCH("add3_scale");
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
