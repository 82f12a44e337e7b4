// Anti-aliased Snake activation, fused into one streaming pass (HBM-bound):
//   up   : u[m]  = 2 * sum_i xpad[i] * f[m + 15 - 2 i]       (replicate pad 5, polyphase x2)
//   snake: s[m]  = u[m] + sin^2(u[m] * e^alpha) / (e^alpha + 1e-9)
//   down : y[t]  = sum_{j<12} s[clamp(2 t + j - 5, 0, 2T-1)] * f[j]
// The reference (layers/activations.py:22-44, 74-138) materialises ~10 tensors of
// size 2T*C; here each thread walks R consecutive frames of 4 channels keeping a
// 6-deep window of x and a 12-deep window of s in registers, so every x element is
// read once (+ 6/R halo) and every s value (one sin) is evaluated once (+ 12/(2R)).
//
// Polyphase form used below (derived from the pad/stride/crop arithmetic):
//   u[2q]   = 2 * sum_{a<6} x[clamp(q - 3 + a)] * f[11 - 2a]
//   u[2q+1] = 2 * sum_{a<6} x[clamp(q - 2 + a)] * f[10 - 2a]
// so the two samples s[2tp+7], s[2tp+8] pushed at step tp both read x[tp+1 .. tp+6].
#include "ptpp_common.h"

namespace {

// sin of the snake argument.  f32 tensors (parity mode): libm-accurate sinf.  bf16 tensors:
// the hardware v_sin_f32 (|abs err| ~1e-6 for the |x*alpha| < 256 this activation sees,
// far inside the bf16 output rounding); the accurate routine's range reduction made this
// memory-bound kernel ALU-bound (profiles/r01_bigvgan_*).
template <typename T>
__device__ __forceinline__ float snake_sin(float v) {
  if constexpr (sizeof(T) == 2) return __sinf(v);
  else return sinf(v);
}


struct SnakeFilt {
  float up[12];
  float dn[12];
};

template <typename T, int G>
__global__ __launch_bounds__(256) void aa_snake_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const float* __restrict__ log_alpha, const SnakeFilt f, int T_,
                                                       int C, int nrun) {
  constexpr int R = 6 * G - 6;  // frames produced per thread
  const int cv = C >> 2;        // channel vectors per row
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (gid >= (int64_t)cv * nrun) return;
  const int c = (int)(gid % cv) * 4;
  const int run = (int)(gid / cv);
  const int t0 = run * R;
  const T* xb = x + (int64_t)b * T_ * C + c;
  T* yb = y + (int64_t)b * T_ * C + c;

  f32x4 ea, inv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ea[e] = __expf(log_alpha[c + e]);
    inv[e] = 1.0f / (ea[e] + 1e-9f);
  }
  const int last = T_ - 1;
  auto ldx = [&](int t) { return Elem<T>::ld4(xb + (int64_t)min(max(t, 0), last) * C); };

  // nx: the six x rows the coming group of steps will consume, requested one group (~600 VALU
  // instructions) ahead so their latency never stalls the walk
  f32x4 xw[6], sw[12], nx[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) xw[a] = ldx(t0 - 6 + a);
#pragma unroll
  for (int a = 0; a < 6; ++a) nx[a] = ldx(t0 + a);
#pragma unroll
  for (int i = 0; i < 12; ++i) sw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mlast = 2 * T_ - 1;

  // Ring positions are static inside the 6-step group: at step k the x ring slot
  // (k % 6) is the oldest entry and the s ring slots (2k % 12), (2k+1) % 12 are
  // the two oldest, so shifts cost no register moves.
#define SNAKE_STEP(K)                                                                     \
  {                                                                                       \
    const int tp = tg + (K);                                                              \
    xw[(K) % 6] = nx[(K)];                                                                \
    nx[(K)] = ldx(tp + 12);                                                               \
    f32x4 uo = f32x4{0.f, 0.f, 0.f, 0.f}, ue = f32x4{0.f, 0.f, 0.f, 0.f};                 \
    _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                       \
      const f32x4 xv = xw[((K) + 1 + a) % 6];                                             \
      uo += xv * f.up[10 - 2 * a];                                                        \
      ue += xv * f.up[11 - 2 * a];                                                        \
    }                                                                                     \
    uo *= 2.0f;                                                                           \
    ue *= 2.0f;                                                                           \
    f32x4 so, se;                                                                         \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                       \
      const float s1 = snake_sin<T>(uo[e] * ea[e]);                                       \
      const float s2 = snake_sin<T>(ue[e] * ea[e]);                                       \
      so[e] = uo[e] + inv[e] * (s1 * s1);                                                 \
      se[e] = ue[e] + inv[e] * (s2 * s2);                                                 \
    }                                                                                     \
    const int m1 = 2 * tp + 7;                                                            \
    const f32x4 prev = sw[(2 * (K) + 11) % 12]; /* newest entry = s(clamp(m1 - 1)) */     \
    if (m1 > mlast) so = prev;                                                            \
    if (m1 + 1 > mlast) se = so;                                                          \
    sw[(2 * (K)) % 12] = so;                                                              \
    sw[(2 * (K) + 1) % 12] = se;                                                          \
    if (m1 + 1 == 0) { /* left edge: every s index < 0 replicates s(0) */                 \
      _Pragma("unroll") for (int i = 0; i < 12; ++i) sw[i] = se;                          \
    }                                                                                     \
    const int t = tp + 1;                                                                 \
    if (t >= t0 && t < T_) {                                                              \
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};                                              \
      _Pragma("unroll") for (int j = 0; j < 12; ++j) acc += sw[(2 * (K) + 2 + j) % 12] * f.dn[j]; \
      Elem<T>::st4(yb + (int64_t)t * C, acc);                                             \
    }                                                                                     \
  }

  for (int g = 0; g < G; ++g) {
    const int tg = t0 - 6 + g * 6;
    if (tg + 1 >= T_) break;
    SNAKE_STEP(0)
    SNAKE_STEP(1)
    SNAKE_STEP(2)
    SNAKE_STEP(3)
    SNAKE_STEP(4)
    SNAKE_STEP(5)
  }
#undef SNAKE_STEP
}

}  // namespace

extern "C" int ptpp_aa_snake_fwd(const void* x, void* y, const float* log_alpha, const float* filt_up,
                                 const float* filt_down, int B, int T, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && y && log_alpha && filt_up && filt_down, "aa_snake: null pointer");
  PTPP_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 4 == 0, "aa_snake: bad shape B=%d T=%d C=%d", B, T, C);
  PTPP_CHECK_ARG(x != y, "aa_snake: in-place not supported");
  SnakeFilt f;
  // filt_* are HOST pointers to the 12 taps (they are tiny module buffers; the
  // Python side passes a host copy so they travel as kernel arguments / SGPRs).
  for (int i = 0; i < 12; ++i) {
    f.up[i] = filt_up[i];
    f.dn[i] = filt_down[i];
  }
  constexpr int G = 12;
  constexpr int R = 6 * G - 6;
  const int nrun = (T + R - 1) / R;
  const int64_t nthr = (int64_t)(C / 4) * nrun;
  dim3 grid((unsigned)((nthr + 255) / 256), B), blk(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL((aa_snake_kernel<float, G>), grid, blk, 0, st, (const float*)x, (float*)y, log_alpha, f, T, C,
                       nrun);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL((aa_snake_kernel<bf16_raw, G>), grid, blk, 0, st, (const bf16_raw*)x, (bf16_raw*)y, log_alpha,
                       f, T, C, nrun);
  else if (dtype == PTPP_F16)
    hipLaunchKernelGGL((aa_snake_kernel<f16_raw, G>), grid, blk, 0, st, (const f16_raw*)x, (f16_raw*)y, log_alpha, f, T, C, nrun);
  else
    PTPP_CHECK_ARG(false, "aa_snake: bad dtype %d", dtype);
  PTPP_CHECK_LAUNCH("aa_snake_fwd");
  return PTPP_OK;
}
