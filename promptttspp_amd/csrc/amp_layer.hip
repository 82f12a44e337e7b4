// One BigVGAN AMP layer in ONE kernel (vocoders/bigvgan.py:42-47, layers/activations.py:22-44):
//
//   y = res_scale * x + out_scale * (conv2(snake2(conv1(snake1(x)))) + b2) [+ res2]
//
// where snake_k is the anti-aliased Snake (replicate-pad, x2 polyphase FIR, x + sin^2(a x)/a, 12-tap
// low-pass + decimate) and conv1 / conv2 are the dilated / plain k-tap Conv1d of the layer.  Unfused, the
// layer moves 9 tensor passes through HBM (snake r+w, conv r+w, snake r+w, conv r+w + residual r); here a
// block owns BT output rows of one utterance with all C channels, keeps the x tile with its halo of
// H = pad2 + 6 + pad1 + 6 rows in LDS and walks the four stages on chip: x is read once (+ 2H/BT halo) and y
// written once.  Built for the two narrow, HBM-bound stages of the generator (C = 64 and C = 32: 74 % of its
// activation traffic); the wide stages stay on the tile GEMM.
//
// Phases of a block (4 waves, __syncthreads between phases):
//   P0  global -> LDS X: rows [t0 - H, t0 + BT + H) of x, time index clamped into [0, T) (the replicate
//       padding of the first Snake), 16-byte chunks, XOR-swizzled rows
//   P1  Snake 1, VALU: a thread owns one channel PAIR and a run of consecutive rows; a 6-deep x window and a
//       12-deep window of upsampled Snake values live in registers (every sin evaluated once); a1 -> LDS A in
//       the compute dtype (rows outside [0, T) are the conv's zero padding)
//   P2  conv1, MFMA: weights as the "A" operand straight from global (fragments of the packed
//       [Cout][ks][Cin] operand, L1/L2 resident: 22-90 KB shared by every block), a1 as "B" from LDS at row
//       offsets j * dil; c1 + bias -> LDS X (the x tile is dead by now) in the compute dtype
//   P3  Snake 2: c1 -> a2 in LDS A (time index clamped into [0, T))
//   P4  conv2, MFMA; epilogue adds bias, the residual x (re-read from global: an L2 hit) and the optional
//       running AMP-block mean res2, 16-byte coalesced stores.
// bf16: v_mfma_f32_16x16x32_bf16; f32 (parity mode): four exact v_mfma_f32_16x16x4_f32 per 16-byte chunk
// pair -- one code path, like conv1d_cl.hip.  Rounding points in bf16 mode are those of the unfused
// pipeline (a1, c1, a2, y are bf16 there too).
#include <stdlib.h>
#include <string.h>

#include "ptpp_common.h"

namespace {

template <int NCH>
__device__ __forceinline__ int aswz(int row);
template <>
__device__ __forceinline__ int aswz<4>(int row) { return (-(row >> 2)) & 3; }
template <>
__device__ __forceinline__ int aswz<8>(int row) { return (row >> 1) & 7; }
template <>
__device__ __forceinline__ int aswz<16>(int row) { return row & 15; }

template <typename T>
struct AMma;
template <>
struct AMma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    const f32x4 af = __builtin_bit_cast(f32x4, a);
    const f32x4 bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc, 0, 0, 0);
  }
};
template <>
struct AMma<bf16_raw> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                  0, 0, 0);
  }
};

// a channel PAIR in LDS / registers
template <typename T>
struct Pair;
template <>
struct Pair<bf16_raw> {
  typedef uint32_t raw;
  static __device__ __forceinline__ void unpack(raw r, float& a, float& b) {
    a = __uint_as_float(r << 16);
    b = __uint_as_float(r & 0xffff0000u);
  }
  static __device__ __forceinline__ raw pack(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  }
};
template <>
struct Pair<float> {
  typedef uint2 raw;
  static __device__ __forceinline__ void unpack(raw r, float& a, float& b) {
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
  }
  static __device__ __forceinline__ raw pack(float a, float b) { return make_uint2(__float_as_uint(a), __float_as_uint(b)); }
};

struct AmpP {
  const void* x;
  void* y;
  const void* res2;
  const void* w1p;
  const void* w2p;
  const float* b1;
  const float* b2;
  const float* la1;
  const float* la2;
  float up1[12], dn1[12], up2[12], dn2[12];
  int B, T, ks, dil;
  float out_scale, res_scale;
  int nMT;
  int skip;  // diagnostics (PTPP_AMP_SKIP): bit 0 P1, 1 P2, 2 P3, 3 P4 MFMA loop, 4 P0 loads
};

// Anti-aliased Snake of rows [o0, o0 + n) of `dst` (dst-local; dst row i is time tdst0 + i) from `src`
// (src row i is time tsrc0 + i; rows are read at clamp(t, 0, T-1) - tsrc0).  Rows whose time lies outside
// [0, T) are written as zeros.  One channel pair per call.
//   up   : u[2q]   = 2 sum_a x[q-3+a] f[11-2a],  u[2q+1] = 2 sum_a x[q-2+a] f[10-2a]   (replicate pad)
//   snake: s = u + sin^2(u e^alpha) / (e^alpha + 1e-9)
//   down : y[t] = sum_{j<12} s[clamp(2t + j - 5, 0, 2T-1)] fdn[j]
// fup2 = 2 * fup (exact: the reference's "* ratio" after the transposed conv commutes with the sum).
// bf16: w0/w1 = e^alpha / (2 pi) for v_sin_f32 (argument in revolutions); f32: w = e^alpha, libm sinf.
template <typename T>
__device__ __forceinline__ float snake_val(float u, float w, float inv) {
  float s;
  if constexpr (sizeof(T) == 2) s = __builtin_amdgcn_sinf(u * w);
  else s = sinf(u * w);
  return fmaf(inv, s * s, u);
}

template <typename T, int NCH>
__device__ __forceinline__ void snake_run(const char* src, char* dst, int tsrc0, int tdst0, int o0, int n, int Tlen,
                                          int cpair, const float (&fup2)[12], const float (&fdn)[12],
                                          float w0, float w1, float inv0, float inv1) {
  constexpr int KC = 16 / (int)sizeof(T);
  constexpr int ROWB = NCH * 16;
  typedef typename Pair<T>::raw raw;
  const int c = cpair * 2;
  const int chunk = c / KC;
  const int inoff = (c % KC) * (int)sizeof(T);
  auto addr = [&](int row) { return row * ROWB + ((chunk ^ aswz<NCH>(row)) << 4) + inoff; };
  auto ldx = [&](int t, float& a, float& b) {
    const int row = min(max(t, 0), Tlen - 1) - tsrc0;
    Pair<T>::unpack(*reinterpret_cast<const raw*>(src + addr(row)), a, b);
  };
  auto st = [&](int i, float a, float b) { *reinterpret_cast<raw*>(dst + addr(i)) = Pair<T>::pack(a, b); };

  const int ta = tdst0 + o0, tb = ta + n;
  const int tv0 = max(ta, 0), tv1 = min(tb, Tlen);
  for (int t = ta; t < min(tb, tv0); ++t) st(t - tdst0, 0.f, 0.f);
  for (int t = max(ta, tv1); t < tb; ++t) st(t - tdst0, 0.f, 0.f);
  if (tv1 <= tv0) return;

  // Step tp pushes s[2tp+7], s[2tp+8] (both read x[tp+1 .. tp+6]) into a 12-deep window and emits row tp+1 from
  // s[2tp-3 .. 2tp+8].  A run that starts inside the utterance warms the window up with the 6 steps tp = tv0-6 ..;
  // a run that starts at the left edge (tv0 < 3) starts at tp = -3 (the first s index >= 1) with the window
  // pre-filled with s[0], which is what every clamped index <= 0 reads.
  float xa[6], xb[6], sa[12], sb[12];
  const int tp0 = max(tv0 - 6, -3);
  float pa = 0.f, pb = 0.f;
  if (tv0 < 3) {  // s[0] = snake(u[0]), u[0] = sum_a x[clamp(a-3)] * 2 f[11-2a]
    float ua = 0.f, ub = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float va, vb;
      ldx(a - 3, va, vb);
      ua = fmaf(va, fup2[11 - 2 * a], ua);
      ub = fmaf(vb, fup2[11 - 2 * a], ub);
    }
    pa = snake_val<T>(ua, w0, inv0);
    pb = snake_val<T>(ub, w1, inv1);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) { sa[i] = pa; sb[i] = pb; }
  // entering step tp the x window holds x[tp .. tp+5]; the step replaces x[tp] by x[tp+6]
#pragma unroll
  for (int a = 0; a < 6; ++a) ldx(tp0 + a, xa[a], xb[a]);
  const int mlast = 2 * Tlen - 1;

#define AMP_SNAKE_STEP(K)                                                                   \
  {                                                                                         \
    const int tp = tg + (K);                                                                \
    ldx(tp + 6, xa[(K) % 6], xb[(K) % 6]);                                                  \
    float uoa = 0.f, uea = 0.f, uob = 0.f, ueb = 0.f;                                       \
    _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                         \
      const float va = xa[((K) + 1 + a) % 6], vb = xb[((K) + 1 + a) % 6];                   \
      uoa = fmaf(va, fup2[10 - 2 * a], uoa);                                                \
      uea = fmaf(va, fup2[11 - 2 * a], uea);                                                \
      uob = fmaf(vb, fup2[10 - 2 * a], uob);                                                \
      ueb = fmaf(vb, fup2[11 - 2 * a], ueb);                                                \
    }                                                                                       \
    float soa = snake_val<T>(uoa, w0, inv0), sea = snake_val<T>(uea, w0, inv0);             \
    float sob = snake_val<T>(uob, w1, inv1), seb = snake_val<T>(ueb, w1, inv1);             \
    const int m1 = 2 * tp + 7;                                                              \
    if (m1 > mlast) { soa = sa[(2 * (K) + 11) % 12]; sob = sb[(2 * (K) + 11) % 12]; }       \
    if (m1 + 1 > mlast) { sea = soa; seb = sob; }                                           \
    sa[(2 * (K)) % 12] = soa; sb[(2 * (K)) % 12] = sob;                                     \
    sa[(2 * (K) + 1) % 12] = sea; sb[(2 * (K) + 1) % 12] = seb;                             \
    const int t = tp + 1;                                                                   \
    if (t >= tv0 && t < tv1) {                                                              \
      float ya = 0.f, yb = 0.f, za = 0.f, zb = 0.f; /* two partial sums: shorter dependent chains */ \
      _Pragma("unroll") for (int j = 0; j < 12; j += 2) {                                   \
        ya = fmaf(sa[(2 * (K) + 2 + j) % 12], fdn[j], ya);                                  \
        yb = fmaf(sb[(2 * (K) + 2 + j) % 12], fdn[j], yb);                                  \
        za = fmaf(sa[(2 * (K) + 3 + j) % 12], fdn[j + 1], za);                              \
        zb = fmaf(sb[(2 * (K) + 3 + j) % 12], fdn[j + 1], zb);                              \
      }                                                                                     \
      st(t - tdst0, ya + za, yb + zb);                                                      \
    }                                                                                       \
  }

  for (int tg = tp0; tg + 1 < tv1; tg += 6) {
    AMP_SNAKE_STEP(0)
    AMP_SNAKE_STEP(1)
    AMP_SNAKE_STEP(2)
    AMP_SNAKE_STEP(3)
    AMP_SNAKE_STEP(4)
    AMP_SNAKE_STEP(5)
  }
#undef AMP_SNAKE_STEP
}

// out channel held by MFMA "A" row i of fragment f (so that a lane's accumulators are whole 16-byte chunks)
template <typename T>
__device__ __forceinline__ int frag_channel(int f, int i) {
  if constexpr (sizeof(T) == 2) return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3);
  else return f * 16 + i;
}

// Implicit-GEMM conv over an LDS-resident activation tile: for the wave's m-fragments (16 time rows each),
// acc[mi][f] = sum_{j, ci} W[co(f, .), j, ci] * act[row0 + 16 * mfrag + lr + j * dil][ci]
template <typename T, int C, int MG>
__device__ __forceinline__ void conv_frags(const char* act, const T* __restrict__ wp, int ks, int dil, int row0, int mf0,
                                           int nmf, int lane, f32x4 (&acc)[MG][C / 16]) {
  constexpr int KC = 16 / (int)sizeof(T);
  constexpr int NCH = C / KC;
  constexpr int ROWB = NCH * 16;
  constexpr int NF = C / 16;
  constexpr int KSTEP = 4 * KC;       // channels per MFMA K step (4 lane groups x one 16-byte chunk)
  constexpr int NKC = C / KSTEP;
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MG; ++mi)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[mi][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* wrow[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wrow[f] = wp + (int64_t)frag_channel<T>(f, lr) * ks * C + lg * KC;
  uint4 wf[NF], wn[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wf[f] = *reinterpret_cast<const uint4*>(wrow[f]);
  const int steps = ks * NKC;
  for (int s = 0; s < steps; ++s) {
    const int j = s / NKC, kc = s - j * NKC;
    if (s + 1 < steps) {
      const int j2 = (s + 1) / NKC, kc2 = (s + 1) - j2 * NKC;
#pragma unroll
      for (int f = 0; f < NF; ++f) wn[f] = *reinterpret_cast<const uint4*>(wrow[f] + j2 * C + kc2 * KSTEP);
    }
    uint4 xf[MG];
#pragma unroll
    for (int mi = 0; mi < MG; ++mi) {
      const int row = row0 + (mf0 + (mi < nmf ? mi : 0)) * 16 + lr + j * dil;
      xf[mi] = *reinterpret_cast<const uint4*>(act + row * ROWB + (((kc * 4 + lg) ^ aswz<NCH>(row)) << 4));
    }
#pragma unroll
    for (int mi = 0; mi < MG; ++mi)
#pragma unroll
      for (int f = 0; f < NF; ++f) AMma<T>::run(acc[mi][f], wf[f], xf[mi]);
#pragma unroll
    for (int f = 0; f < NF; ++f) wf[f] = wn[f];
  }
}

template <typename T, int C, int BT, int MG>
__global__ __launch_bounds__(256) void amp_layer_kernel(const AmpP p) {
  constexpr int NT = 256;
  constexpr int KC = 16 / (int)sizeof(T);
  constexpr int NCH = C / KC;
  constexpr int ROWB = NCH * 16;
  constexpr int NF = C / 16;
  constexpr int CP = C / 2;          // channel pairs
  constexpr int NRUN = NT / CP;      // row runs per snake phase
  static_assert(NT % CP == 0 && BT % 16 == 0, "geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ks = p.ks, dil = p.dil, Tlen = p.T;
  const int pad1 = dil * (ks - 1) / 2, pad2 = (ks - 1) / 2;
  const int n_c1 = BT + 2 * pad2 + 12;          // conv1 output rows snake 2 reads
  const int M1 = (n_c1 + 15) & ~15;             // ... rounded up to whole MFMA fragments
  const int n_a1 = n_c1 + 2 * pad1;             // snake-1 rows conv1 reads (without the fragment overhang)
  const int n_x = n_a1 + 12;
  const int rowsX = max(n_x, M1);
  char* Xs = smem;
  char* As = smem + rowsX * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid % p.nMT, b = lid / p.nMT;
  const int t0 = mt * BT;
  const int tc0 = t0 - pad2 - 6;     // time of c1 row 0
  const int ta0 = tc0 - pad1;        // time of a1 row 0
  const int tx0 = ta0 - 6;           // time of X row 0
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * Tlen * C;

  // ---- P0: x tile -> LDS (time index clamped: replicate padding of the first Snake) ----
  for (int idx = tid; idx < n_x * NCH; idx += NT) {
    const int r = idx / NCH, ch = idx - r * NCH;
    const int t = min(max(tx0 + r, 0), Tlen - 1);
    const uint4 v = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + ch * KC);
    *reinterpret_cast<uint4*>(Xs + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4)) = v;
  }
  const int cpair = tid % CP, run = tid / CP;
  // Snake constants of this thread's channel pair: w = e^alpha (bf16: / 2 pi, the v_sin_f32 argument unit)
  constexpr float WSC = sizeof(T) == 2 ? 0.15915494309189535f : 1.0f;
  float ea0, ea1, inv0, inv1;
  {
    const float a0 = __expf(p.la1[2 * cpair]), a1 = __expf(p.la1[2 * cpair + 1]);
    ea0 = a0 * WSC; ea1 = a1 * WSC; inv0 = 1.0f / (a0 + 1e-9f); inv1 = 1.0f / (a1 + 1e-9f);
  }
  __syncthreads();

  // ---- P1: snake 1: X -> A (a1 rows [0, n_a1); time ta0 + i) ----
  {
    const int R = (n_a1 + NRUN - 1) / NRUN;
    const int o0 = run * R, n = min(R, n_a1 - o0);
    float fu[12], fd[12];  // the taps as scalars (a pointer into the by-value argument block would spill it)
#pragma unroll
    for (int i = 0; i < 12; ++i) { fu[i] = 2.0f * p.up1[i]; fd[i] = p.dn1[i]; }
    if (n > 0) snake_run<T, NCH>(Xs, As, tx0, ta0, o0, n, Tlen, cpair, fu, fd, ea0, ea1, inv0, inv1);
  }
  __syncthreads();

  // ---- P2: conv1 (dilated): A -> c1 rows [0, M1) in X ----
  {
    const int nfr = M1 / 16;
    const int lr = lane & 15, lg = lane >> 4;
    for (int g = wave; g * MG < nfr; g += 4) {
      const int mf0 = g * MG, nmf = min(MG, nfr - mf0);
      f32x4 acc[MG][NF];
      conv_frags<T, C, MG>(As, reinterpret_cast<const T*>(p.w1p), ks, dil, 0, mf0, nmf, lane, acc);
#pragma unroll
      for (int mi = 0; mi < MG; ++mi) {
        if (mi < nmf) {
          const int row = (mf0 + mi) * 16 + lr;
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int h = 0; h < NF / 2; ++h) {
              const int co = h * 32 + lg * 8;
              const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b1 + co), bB = *reinterpret_cast<const f32x4*>(p.b1 + co + 4);
              const f32x4 v0 = acc[mi][2 * h] + bA, v1 = acc[mi][2 * h + 1] + bB;
              uint4 o;
              o.x = Pair<T>::pack(v0[0], v0[1]); o.y = Pair<T>::pack(v0[2], v0[3]);
              o.z = Pair<T>::pack(v1[0], v1[1]); o.w = Pair<T>::pack(v1[2], v1[3]);
              *reinterpret_cast<uint4*>(Xs + row * ROWB + (((h * 4 + lg) ^ aswz<NCH>(row)) << 4)) = o;
            }
          } else {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const int co = f * 16 + lg * 4;
              const f32x4 v = acc[mi][f] + *reinterpret_cast<const f32x4*>(p.b1 + co);
              *reinterpret_cast<f32x4*>(Xs + row * ROWB + (((f * 4 + lg) ^ aswz<NCH>(row)) << 4)) = v;
            }
          }
        }
      }
    }
  }
  {
    const float a0 = __expf(p.la2[2 * cpair]), a1 = __expf(p.la2[2 * cpair + 1]);
    ea0 = a0 * WSC; ea1 = a1 * WSC; inv0 = 1.0f / (a0 + 1e-9f); inv1 = 1.0f / (a1 + 1e-9f);
  }
  __syncthreads();

  // ---- P3: snake 2: c1 (X) -> a2 rows [0, BT + 2 pad2) in A; time t0 - pad2 + i ----
  {
    const int n_a2 = BT + 2 * pad2;
    const int R = (n_a2 + NRUN - 1) / NRUN;
    const int o0 = run * R, n = min(R, n_a2 - o0);
    float fu[12], fd[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { fu[i] = 2.0f * p.up2[i]; fd[i] = p.dn2[i]; }
    if (n > 0) snake_run<T, NCH>(Xs, As, tc0, t0 - pad2, o0, n, Tlen, cpair, fu, fd, ea0, ea1, inv0, inv1);
  }
  __syncthreads();

  // ---- P4: conv2: A -> y (+ bias, residual x, running mean res2) ----
  if constexpr (sizeof(T) == 2) {
    // bf16: through an LDS image of the output tile in X (free since the barrier above).  The MFMA layout gives a lane
    // 16 bytes of one row and its neighbours in lane order other rows: direct residual loads / stores are 64 separate
    // 16-byte requests per instruction, and -- the output may alias the running mean -- a load -> add -> store loop pays
    // one memory round trip per fragment pair.  Here the residual rows are fetched row-contiguous while the MFMAs run,
    // parked in the image, updated in place by the MFMA-layout lanes and stored row-contiguous.
    constexpr int nfr = BT / 16;
    static_assert((nfr + MG - 1) / MG <= 4, "one fragment group per wave");
    constexpr int NRV = BT * NCH / NT;  // 16-byte vectors of the tile per thread
    static_assert(NRV * NT == BT * NCH, "tile vectors must divide over the threads");
    const int lr = lane & 15, lg = lane >> 4;
    T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * Tlen * C;
    const T* r2b = p.res2 ? reinterpret_cast<const T*>(p.res2) + (int64_t)b * Tlen * C : nullptr;
    const float osc = p.out_scale, rsc = p.res_scale;
    uint4 rx[NRV];
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = t0 + r;
      rx[i] = t < Tlen ? *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + ch * KC) : make_uint4(0, 0, 0, 0);
    }
    const int mf0 = wave * MG, nmf = min(MG, nfr - mf0);  // this wave's fragment group (none: nmf <= 0)
    f32x4 acc[MG][NF];
    uint4 q2[MG][NF / 2];
    if (nmf > 0) {
      if (r2b) {  // (two of the nine layers of a stage: the running mean of the AMP blocks)
#pragma unroll
        for (int mi = 0; mi < MG; ++mi) {
          const int t = t0 + (mf0 + mi) * 16 + lr;
#pragma unroll
          for (int h = 0; h < NF / 2; ++h)
            q2[mi][h] = (mi < nmf && t < Tlen) ? *reinterpret_cast<const uint4*>(r2b + (int64_t)t * C + h * 32 + lg * 8)
                                               : make_uint4(0, 0, 0, 0);
        }
      }
      conv_frags<T, C, MG>(As, reinterpret_cast<const T*>(p.w2p), ks, 1, 0, mf0, nmf, lane, acc);
    }
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      *reinterpret_cast<uint4*>(Xs + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4)) = rx[i];
    }
    __syncthreads();
    if (nmf > 0) {
#pragma unroll
      for (int mi = 0; mi < MG; ++mi) {
        if (mi < nmf) {
          const int r = (mf0 + mi) * 16 + lr;
#pragma unroll
          for (int h = 0; h < NF / 2; ++h) {
            const int co = h * 32 + lg * 8;
            const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b2 + co), bB = *reinterpret_cast<const f32x4*>(p.b2 + co + 4);
            f32x4 v0 = (acc[mi][2 * h] + bA) * osc, v1 = (acc[mi][2 * h + 1] + bB) * osc;
            uint4* slot = reinterpret_cast<uint4*>(Xs + r * ROWB + (((co / KC) ^ aswz<NCH>(r)) << 4));
            const uint4 rr = *slot;
            v0[0] += __uint_as_float(rr.x << 16) * rsc; v0[1] += __uint_as_float(rr.x & 0xffff0000u) * rsc;
            v0[2] += __uint_as_float(rr.y << 16) * rsc; v0[3] += __uint_as_float(rr.y & 0xffff0000u) * rsc;
            v1[0] += __uint_as_float(rr.z << 16) * rsc; v1[1] += __uint_as_float(rr.z & 0xffff0000u) * rsc;
            v1[2] += __uint_as_float(rr.w << 16) * rsc; v1[3] += __uint_as_float(rr.w & 0xffff0000u) * rsc;
            if (r2b) {
              const uint4 q = q2[mi][h];
              v0[0] += __uint_as_float(q.x << 16); v0[1] += __uint_as_float(q.x & 0xffff0000u);
              v0[2] += __uint_as_float(q.y << 16); v0[3] += __uint_as_float(q.y & 0xffff0000u);
              v1[0] += __uint_as_float(q.z << 16); v1[1] += __uint_as_float(q.z & 0xffff0000u);
              v1[2] += __uint_as_float(q.w << 16); v1[3] += __uint_as_float(q.w & 0xffff0000u);
            }
            uint4 o;
            o.x = Pair<T>::pack(v0[0], v0[1]); o.y = Pair<T>::pack(v0[2], v0[3]);
            o.z = Pair<T>::pack(v1[0], v1[1]); o.w = Pair<T>::pack(v1[2], v1[3]);
            *slot = o;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = t0 + r;
      if (t < Tlen)
        *reinterpret_cast<uint4*>(yb + (int64_t)t * C + ch * KC) = *reinterpret_cast<const uint4*>(Xs + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4));
    }
  } else {
    constexpr int nfr = BT / 16;
    const int lr = lane & 15, lg = lane >> 4;
    T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * Tlen * C;
    const T* r2b = p.res2 ? reinterpret_cast<const T*>(p.res2) + (int64_t)b * Tlen * C : nullptr;
    const float osc = p.out_scale, rsc = p.res_scale;
    for (int g = wave; g * MG < nfr; g += 4) {
      const int mf0 = g * MG, nmf = min(MG, nfr - mf0);
      f32x4 acc[MG][NF];
      conv_frags<T, C, MG>(As, reinterpret_cast<const T*>(p.w2p), ks, 1, 0, mf0, nmf, lane, acc);
#pragma unroll
      for (int mi = 0; mi < MG; ++mi) {
        const int t = t0 + (mf0 + mi) * 16 + lr;
        if (mi < nmf && t < Tlen) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const int co = f * 16 + lg * 4;
            f32x4 v = (acc[mi][f] + *reinterpret_cast<const f32x4*>(p.b2 + co)) * osc;
            v += *reinterpret_cast<const f32x4*>(xb + (int64_t)t * C + co) * rsc;
            if (r2b) v += *reinterpret_cast<const f32x4*>(r2b + (int64_t)t * C + co);
            *reinterpret_cast<f32x4*>(yb + (int64_t)t * C + co) = v;
          }
        }
      }
    }
  }
}

// =====================================================================================================
// bf16: the two FIR stages of the anti-aliased Snake on the matrix cores.
//
// Measured on the kernel above (rocprofv3 PMC, profiles/r02_pmc_amp_valu.txt): SQ_ACTIVE_INST_VALU covers ~99 % of
// the launch -- the layer is VALU-bound by the Snake, 24 of whose ~45 VALU operations per output element are the
// two 12-tap FIRs (a wave64 VALU instruction issues in 4 cycles on gfx950).  Both FIRs are contractions over TIME
// with constant coefficients, i.e. small Toeplitz GEMMs:
//   up   (16 upsampled samples x 16 channels):  U[m, c] = sum_q Fup[m, q] X[q, c]      K = 16 input rows
//   down (16 output rows x 16 channels):        Y[c, t] = sum_m S[c, m] Fdn[m, t]      K = 48 upsampled samples
// One v_mfma_f32_16x16x16_bf16 per up tile and a K = 32 + a K = 16 MFMA per output cell replace 24 x 256 VALU FMAs.
// The operand layouts chain without any data movement: the X fragment comes out of LDS through the transposing
// read ds_read_b64_tr_b16; the up MFMA is oriented D[m][c] so that a lane ends up holding 4 consecutive samples m of
// ONE channel -- exactly the "A" fragment the down MFMA wants (the order of the contraction slots is free as long
// as the constant Toeplitz operand uses the same order), and the down MFMA is oriented D[c][t] so that a lane holds
// 4 consecutive channels of one row: one 8-byte LDS store.  What stays on the VALU is the Snake itself
// (mul, v_sin, mul, fma per upsampled sample) and the bf16 packs.  The f32 taps enter as hi + lo bf16 pairs (two
// MFMAs), so the filters carry no bf16 coefficient error; the upsampled Snake values are rounded to bf16 before the
// low-pass (they are f32 registers in the VALU version).
// Sequence edges: the replicate padding of x is in the tile (rows are stored time-clamped), and the clamped s
// index of the low-pass folds the outer taps onto s[0] / s[2T-1]: a different constant operand for the few cells
// at the ends of an utterance, computed on the fly from prefix / suffix sums of the taps.
// =====================================================================================================
typedef __attribute__((ext_vector_type(4))) short v4s_t;
typedef __attribute__((ext_vector_type(8))) short v8s_t;

// Wait states between a CHAIN of dependent MFMAs on one accumulator and the first VALU read of it.  Measured
// (tools/diag_amp_det*.py, profiles/r02_amp_mfma_result_hazard.txt): with the s_nop the compiler derives from the LAST
// MFMA of the chain alone, the first two accumulator registers read after a K=32, K=32, K=16, K=16 chain came back
// stale in ~0.5 % of the elements whenever a second workgroup shared the SIMD (never on an otherwise idle CU) -- the
// outputs were then not run-to-run reproducible.  The asm ties the accumulator so nothing is scheduled across it.
#define MFMA_SETTLE(acc) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc))

__device__ __forceinline__ v4s_t tr16_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ void split_bf16(float v, short& hi, short& lo) {
  const bf16_raw h = f32_to_bf16(v);
  hi = (short)h;
  lo = (short)f32_to_bf16(v - bf16_to_f32(h));
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

// taps table in LDS, per activation: [0,12) 2*up, [12,24) down, [24,36) prefix sums of down, [36,48) suffix sums
constexpr int TAB = 48;

struct SnakeOps {
  v4s_t up_hi, up_lo;   // up stage "A": [m_local = lane & 15][k = 4 kg + e]
  v8s_t d1_hi, d1_lo;   // down stage "B", K = 32: slots e < 4 -> tile 2b sample 4 kg + e, e >= 4 -> tile 2b+1
  v4s_t d2_hi, d2_lo;   // down stage "B", K = 16: tile 2b+2
};

// coefficient of upsampled sample m (cell-relative offset moff, cell base sample mb) in output row j of the cell
// (time t = tcell + j): sum of the taps fdn[jj] with clamp(2t + jj - 5, 0, mlast) == m
__device__ __forceinline__ float down_coef(const float* tab, int moff, int j, int mb, int mlast, bool edge) {
  const int j0 = moff - 2 * j - 3;  // the tap that reads sample m for row t when nothing is clamped
  if (!edge) return (j0 >= 0 && j0 < 12) ? tab[12 + j0] : 0.f;
  const int m = mb + moff;
  if (m < 0 || m > mlast) return 0.f;
  if (m == 0) return j0 >= 0 ? tab[24 + min(j0, 11)] : 0.f;
  if (m == mlast) return j0 <= 11 ? tab[36 + max(j0, 0)] : 0.f;
  return (j0 >= 0 && j0 < 12) ? tab[12 + j0] : 0.f;
}

__device__ __forceinline__ void down_ops(const float* tab, int lane, int mb, int mlast, bool edge, SnakeOps& k) {
  const int j = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int moff = e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4);
    short hi, lo;
    split_bf16(down_coef(tab, moff, j, mb, mlast, edge), hi, lo);
    k.d1_hi[e] = hi;
    k.d1_lo[e] = lo;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    short hi, lo;
    split_bf16(down_coef(tab, 32 + 4 * kg + e, j, mb, mlast, edge), hi, lo);
    k.d2_hi[e] = hi;
    k.d2_lo[e] = lo;
  }
}

__device__ __forceinline__ void up_ops(const float* tab, int lane, SnakeOps& k) {
  const int i = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kk = 4 * kg + e;
    // u[m0 + i] = sum_a x[q0 + kk] * f2[...]:  i even: kk = i/2 + a, tap 11 - 2a;  i odd: kk = (i+1)/2 + a, tap 10 - 2a
    const int a = (i & 1) ? kk - (i + 1) / 2 : kk - i / 2;
    const int idx = (i & 1) ? 10 - 2 * a : 11 - 2 * a;
    short hi, lo;
    split_bf16((a >= 0 && a < 6) ? tab[idx] : 0.f, hi, lo);
    k.up_hi[e] = hi;
    k.up_lo[e] = lo;
  }
}

// Snake of 16-row cells [b0, b1) x the 16 channels of group cg: src rows [8n, 8n + 16) feed up-tile n (src row 0
// is 7 rows before dst row 0 in time); dst row i is time tdst0 + i; rows outside [0, T) are written as zeros.
template <int NCH>
__device__ __forceinline__ void snake_cells(const char* src, char* dst, const float* tab, int tdst0, int b0, int b1,
                                            int cg, int Tlen, int lane, float w, float inv, const SnakeOps& kin) {
  constexpr int ROWB = NCH * 16;
  const int i16 = lane & 15, kg = lane >> 4;
  const int mlast = 2 * Tlen - 1;
  // transposing read: lane i of a 16-lane group points at row (i >> 2), 4 channels (i & 3) of the 4 x 16 block
  const int rrow = 4 * kg + (i16 >> 2);
  const int rcol = cg * 16 + 4 * (i16 & 3);           // channel of the 8-byte piece
  const int rchunk = rcol >> 3, roff = (rcol & 7) * 2;
  auto tile = [&](int n) {  // Snake of up-tile n: 4 consecutive samples of this lane's channel, bf16
    const int row = 8 * n + rrow;
    const v4s_t xb = tr16_read(src + row * ROWB + ((rchunk ^ aswz<NCH>(row)) << 4) + roff);
    f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
    u = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kin.up_hi, xb, u, 0, 0, 0);
    u = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kin.up_lo, xb, u, 0, 0, 0);
    MFMA_SETTLE(u);
    float sv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sn = __builtin_amdgcn_sinf(u[e] * w);
      sv[e] = fmaf(inv, sn * sn, u[e]);
    }
    return make_uint2(pack_bf16x2(sv[0], sv[1]), pack_bf16x2(sv[2], sv[3]));
  };
  // store: lane holds channels cg*16 + 4 kg .. +3 of row (16 b + i16)
  const int wcol = cg * 16 + 4 * kg;
  const int wchunk = wcol >> 3, woff = (wcol & 7) * 2;
  uint2 s0 = tile(2 * b0);
  for (int b = b0; b < b1; ++b) {
    const uint2 s1 = tile(2 * b + 1), s2 = tile(2 * b + 2);
    const int tcell = tdst0 + 16 * b, mb = 2 * tcell - 8;
    const bool edge = mb <= 0 || mb + 47 >= mlast;
    SnakeOps ke;
    if (edge) down_ops(tab, lane, mb, mlast, true, ke);
    const v8s_t d1h = edge ? ke.d1_hi : kin.d1_hi, d1l = edge ? ke.d1_lo : kin.d1_lo;
    const v4s_t d2h = edge ? ke.d2_hi : kin.d2_hi, d2l = edge ? ke.d2_lo : kin.d2_lo;
    const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, uint4{s0.x, s0.y, s1.x, s1.y});
    const v4s_t a2 = __builtin_bit_cast(v4s_t, s2);
    f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
    y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8_t, d1h), y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8_t, d1l), y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, d2h, y, 0, 0, 0);
    y = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a2, d2l, y, 0, 0, 0);
    MFMA_SETTLE(y);
    const int row = 16 * b + i16, t = tcell + i16;
    const bool ok = t >= 0 && t < Tlen;
    uint2 o;
    o.x = ok ? pack_bf16x2(y[0], y[1]) : 0u;
    o.y = ok ? pack_bf16x2(y[2], y[3]) : 0u;
    *reinterpret_cast<uint2*>(dst + row * ROWB + ((wchunk ^ aswz<NCH>(row)) << 4) + woff) = o;
    s0 = s2;
  }
}

// Persistent blocks: a block walks a contiguous range of tiles.  Measured on the one-tile-per-block form of this
// kernel (profiles/r02_amp_layer_ablation.txt): with every compute phase switched off the launch still took 45 % of
// its time -- 60 000 short-lived blocks, each paying its own chain of dependent memory round trips (tile load ->
// ... -> residual load -> store).  Here the next tile's x rows are requested at the top of a tile into registers
// and written to the second X slot after the first Snake (their HBM latency hides behind it), the tap tables and
// Toeplitz operands are built once per block, and a block's consecutive tiles share their halo rows in L2.
template <int C, int BT, int MG1, int MG2, bool PS>
__global__ __launch_bounds__(256) void amp_layer_mfma_kernel(const AmpP p) {
  typedef bf16_raw T;
  constexpr int NT = 256;
  constexpr int KC = 8;
  constexpr int NCH = C / KC;
  constexpr int ROWB = NCH * 16;
  constexpr int NF = C / 16;
  constexpr int NCG = C / 16;          // 16-channel groups
  constexpr int WPG = 4 / NCG > 0 ? 4 / NCG : 1;  // waves sharing one channel group (time split)
  constexpr int PF = C == 32 ? 6 : 8;  // 16-byte chunks of the x tile per thread (largest halo: ks = 11, dil = 5)
  static_assert(NCG <= 4 && BT % 16 == 0, "geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ks = p.ks, dil = p.dil, Tlen = p.T;
  const int pad1 = dil * (ks - 1) / 2, pad2 = (ks - 1) / 2;
  const int nc2 = (BT + 2 * pad2 + 15) >> 4;   // 16-row cells of a2 (dst of snake 2, rows t0 - pad2 + i)
  const int M1 = 16 * nc2 + 16;                // c1 rows snake 2 reads: times [t0 - pad2 - 7, ...)
  const int n_a1 = M1 + 2 * pad1;              // a1 rows conv1 reads
  const int nc1 = (n_a1 + 15) >> 4;
  const int rowsX = 16 * nc1 + 16;             // x rows snake 1 reads (>= M1)
  const int rowsA = 16 * max(nc1, nc2);
  char* Xbuf = smem;                                             // PS: 2 slots of rowsX rows, else 1
  char* As = smem + (PS ? 2 : 1) * rowsX * ROWB;
  float* tab = reinterpret_cast<float*>(As + rowsA * ROWB);      // 2 x TAB floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = p.B * p.nMT;
  const int tile_lo = (int)((int64_t)blockIdx.x * ntiles / gridDim.x), tile_hi = (int)((int64_t)(blockIdx.x + 1) * ntiles / gridDim.x);
  if (tile_lo >= tile_hi) return;
  if (p.skip & 256) {  // diagnostics: zero the whole allocation first
    const int total = ((PS ? 2 : 1) * rowsX + rowsA) * ROWB;
    for (int i = tid * 16; i < total; i += NT * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
  }

  if (tid < 24) {
    const int a = tid / 12, i = tid % 12;
    const float* up = a ? p.up2 : p.up1;
    const float* dn = a ? p.dn2 : p.dn1;
    float ps = 0.f, ss = 0.f;
    for (int q = 0; q <= i; ++q) ps += dn[q];
    for (int q = i; q < 12; ++q) ss += dn[q];
    float* tb = tab + a * TAB;
    tb[i] = 2.0f * up[i];
    tb[12 + i] = dn[i];
    tb[24 + i] = ps;
    tb[36 + i] = ss;
  }
  // x rows of a tile, time index clamped (replicate padding), as PF 16-byte chunks per thread
  auto fetch = [&](int tile, uint4 (&v)[PS ? PF : 1]) {
    const int mt = tile % p.nMT, b = tile / p.nMT;
    const int tx0 = mt * BT - pad2 - 7 - pad1 - 7;
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * Tlen * C;
#pragma unroll
    for (int q = 0; q < (PS ? PF : 1); ++q) {
      const int idx = tid + q * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = min(max(tx0 + min(r, rowsX - 1), 0), Tlen - 1);
      v[q] = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + ch * KC);
    }
  };
  auto put = [&](char* Xs, const uint4 (&v)[PS ? PF : 1]) {
#pragma unroll
    for (int q = 0; q < (PS ? PF : 1); ++q) {
      const int idx = tid + q * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      if (r < rowsX) *reinterpret_cast<uint4*>(Xs + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4)) = v[q];
    }
  };
  uint4 pf[PS ? PF : 1];
  if constexpr (PS) {
    fetch(tile_lo, pf);
    put(Xbuf, pf);
  } else {  // one tile per block: straight copy, no registers held
    const int mt = tile_lo % p.nMT, b = tile_lo / p.nMT;
    const int tx0 = mt * BT - pad2 - 7 - pad1 - 7;
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * Tlen * C;
    for (int idx = tid; idx < rowsX * NCH; idx += NT) {
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = min(max(tx0 + r, 0), Tlen - 1);
      *reinterpret_cast<uint4*>(Xbuf + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4)) =
          *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + ch * KC);
    }
  }
  // this wave's channel group and time share in the Snake phases; its Snake constants and FIR operands
  const int cg = wave % NCG, part = wave / NCG;
  const int chl = cg * 16 + (lane & 15);
  constexpr float INV2PI = 0.15915494309189535f;
  const float al1 = __expf(p.la1[chl]), al2 = __expf(p.la2[chl]);
  __syncthreads();
  const int lr = lane & 15, lg = lane >> 4;
  const float osc = p.out_scale, rsc = p.res_scale;

  int tile = tile_lo;
  do {
    char* Xs = Xbuf + (PS ? ((tile - tile_lo) & 1) * rowsX * ROWB : 0);
    char* Xn = Xbuf + (PS ? ((tile - tile_lo + 1) & 1) * rowsX * ROWB : 0);
    const int mt = tile % p.nMT, b = tile / p.nMT;
    const int t0 = mt * BT;
    const int ta2 = t0 - pad2;         // time of a2 row 0
    const int tc0 = ta2 - 7;           // time of c1 row 0
    const int ta0 = tc0 - pad1;        // time of a1 row 0 (x row 0 is 7 rows earlier)
    const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * Tlen * C;

    auto dump = [&](const char* buf, int row_of_t0) {  // diagnostics (PTPP_AMP_SKIP = 101..103): an intermediate -> y
      T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * Tlen * C;
      for (int idx = tid; idx < BT * NCH; idx += NT) {
        const int m = idx / NCH, ch = idx - m * NCH, r = row_of_t0 + m;
        if (t0 + m < Tlen)
          *reinterpret_cast<uint4*>(yb + (int64_t)(t0 + m) * C + ch * KC) = *reinterpret_cast<const uint4*>(buf + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4));
      }
    };
    const bool more = PS && tile + 1 < tile_hi;
    if constexpr (PS) {
      if (more) fetch(tile + 1, pf);   // in flight during the first Snake
    }

    // ---- P1: snake 1: X -> a1 cells [0, nc1) ----
    {
      SnakeOps k1;  // (rebuilt per tile: ~100 instructions, instead of 16 registers held across all phases)
      up_ops(tab, lane, k1);
      down_ops(tab, lane, 0, 0, false, k1);
      const int per = (nc1 + WPG - 1) / WPG;
      const int c0 = part * per, c1 = min(nc1, c0 + per);
      if (part < WPG && c0 < c1) snake_cells<NCH>(Xs, As, tab, ta0, c0, c1, cg, Tlen, lane, al1 * INV2PI, 1.0f / (al1 + 1e-9f), k1);
    }
    if constexpr (PS) {
      if (more) put(Xn, pf);           // (slot last read by the second Snake of the previous tile)
    }
    __syncthreads();
    if (p.skip == 101) { dump(As, t0 - ta0); return; }

    // ---- P2: conv1 (dilated): a1 -> c1 rows [0, M1) in X ----
    {
      const int nfr = M1 / 16;
      for (int g = wave; g * MG1 < nfr; g += 4) {
        const int mf0 = g * MG1, nmf = min(MG1, nfr - mf0);
        f32x4 acc[MG1][NF];
        conv_frags<T, C, MG1>(As, reinterpret_cast<const T*>(p.w1p), ks, dil, 0, mf0, nmf, lane, acc);
#pragma unroll
        for (int mi = 0; mi < MG1; ++mi) {
          if (mi < nmf) {
            const int row = (mf0 + mi) * 16 + lr;
#pragma unroll
            for (int h = 0; h < NF / 2; ++h) {
              const int co = h * 32 + lg * 8;
              const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b1 + co), bB = *reinterpret_cast<const f32x4*>(p.b1 + co + 4);
              const f32x4 v0 = acc[mi][2 * h] + bA, v1 = acc[mi][2 * h + 1] + bB;
              uint4 o;
              o.x = pack_bf16x2(v0[0], v0[1]); o.y = pack_bf16x2(v0[2], v0[3]);
              o.z = pack_bf16x2(v1[0], v1[1]); o.w = pack_bf16x2(v1[2], v1[3]);
              *reinterpret_cast<uint4*>(Xs + row * ROWB + (((h * 4 + lg) ^ aswz<NCH>(row)) << 4)) = o;
            }
          }
        }
      }
    }
    __syncthreads();
    // c1 rows whose time lies outside [0, T) hold sums over the zero padding: the second Snake replicates the edge
    // rows instead (only tiles at the ends of an utterance)
    if (tc0 < 0 || tc0 + M1 > Tlen) {
      for (int idx = tid; idx < M1 * NCH; idx += NT) {
        const int r = idx / NCH, ch = idx - r * NCH;
        const int t = tc0 + r;
        if (t < 0 || t >= Tlen) {
          const int rs = min(max(t, 0), Tlen - 1) - tc0;
          *reinterpret_cast<uint4*>(Xs + r * ROWB + ((ch ^ aswz<NCH>(r)) << 4)) =
              *reinterpret_cast<const uint4*>(Xs + rs * ROWB + ((ch ^ aswz<NCH>(rs)) << 4));
        }
      }
      __syncthreads();
    }

    if (p.skip == 102) { dump(Xs, t0 - tc0); return; }
    // ---- P3: snake 2: c1 (X) -> a2 cells [0, nc2) in A ----
    {
      SnakeOps k2;
      up_ops(tab + TAB, lane, k2);
      down_ops(tab + TAB, lane, 0, 0, false, k2);
      const int per = (nc2 + WPG - 1) / WPG;
      const int c0 = part * per, c1 = min(nc2, c0 + per);
      if (part < WPG && c0 < c1) snake_cells<NCH>(Xs, As, tab + TAB, ta2, c0, c1, cg, Tlen, lane, al2 * INV2PI, 1.0f / (al2 + 1e-9f), k2);
    }
    __syncthreads();

    if (p.skip == 103) { dump(As, t0 - ta2); return; }
    // ---- P4: conv2: a2 -> y (+ bias, residual x, running mean res2) ----
    {
      constexpr int nfr = BT / 16;
      T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * Tlen * C;
      const T* r2b = p.res2 ? reinterpret_cast<const T*>(p.res2) + (int64_t)b * Tlen * C : nullptr;
      for (int g = wave; g * MG2 < nfr; g += 4) {
        const int mf0 = g * MG2, nmf = min(MG2, nfr - mf0);
        // the residual rows (an L2 hit: fetched for this tile a few microseconds ago) are requested BEFORE the K
        // loop; loaded in the epilogue, each row group paid a full memory round trip in sequence
        uint4 rx[PS ? MG2 : 1][NF / 2];
        if constexpr (PS) {
#pragma unroll
          for (int mi = 0; mi < MG2; ++mi) {
            const int t = min(t0 + (mf0 + mi) * 16 + lr, Tlen - 1);
#pragma unroll
            for (int h = 0; h < NF / 2; ++h) rx[mi][h] = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + h * 32 + lg * 8);
          }
        }
        f32x4 acc[MG2][NF];
        conv_frags<T, C, MG2>(As, reinterpret_cast<const T*>(p.w2p), ks, 1, 0, mf0, nmf, lane, acc);
#pragma unroll
        for (int mi = 0; mi < MG2; ++mi) {
          const int t = t0 + (mf0 + mi) * 16 + lr;
          if (mi < nmf && t < Tlen) {
#pragma unroll
            for (int h = 0; h < NF / 2; ++h) {
              const int co = h * 32 + lg * 8;
              const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b2 + co), bB = *reinterpret_cast<const f32x4*>(p.b2 + co + 4);
              f32x4 v0 = (acc[mi][2 * h] + bA) * osc, v1 = (acc[mi][2 * h + 1] + bB) * osc;
              const uint4 r = PS ? rx[PS ? mi : 0][h] : *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + co);
              v0[0] += __uint_as_float(r.x << 16) * rsc; v0[1] += __uint_as_float(r.x & 0xffff0000u) * rsc;
              v0[2] += __uint_as_float(r.y << 16) * rsc; v0[3] += __uint_as_float(r.y & 0xffff0000u) * rsc;
              v1[0] += __uint_as_float(r.z << 16) * rsc; v1[1] += __uint_as_float(r.z & 0xffff0000u) * rsc;
              v1[2] += __uint_as_float(r.w << 16) * rsc; v1[3] += __uint_as_float(r.w & 0xffff0000u) * rsc;
              if (r2b) {  // (the last layer of a block only: 3 of 9 launches)
                const uint4 q = *reinterpret_cast<const uint4*>(r2b + (int64_t)t * C + co);
                v0[0] += __uint_as_float(q.x << 16); v0[1] += __uint_as_float(q.x & 0xffff0000u);
                v0[2] += __uint_as_float(q.y << 16); v0[3] += __uint_as_float(q.y & 0xffff0000u);
                v1[0] += __uint_as_float(q.z << 16); v1[1] += __uint_as_float(q.z & 0xffff0000u);
                v1[2] += __uint_as_float(q.w << 16); v1[3] += __uint_as_float(q.w & 0xffff0000u);
              }
              uint4 o;
              o.x = pack_bf16x2(v0[0], v0[1]); o.y = pack_bf16x2(v0[2], v0[3]);
              o.z = pack_bf16x2(v1[0], v1[1]); o.w = pack_bf16x2(v1[2], v1[3]);
              *reinterpret_cast<uint4*>(yb + (int64_t)t * C + co) = o;
            }
          }
        }
      }
    }
    if constexpr (PS) __syncthreads();  // the next tile's first Snake overwrites A
  } while (PS && ++tile < tile_hi);
}

template <int C, int BT, int MG1, int MG2, bool PS>
int launch_amp_mfma(AmpP& p, hipStream_t st) {
  constexpr int ROWB = (C / 8) * 16;
  const int pad1 = p.dil * (p.ks - 1) / 2, pad2 = (p.ks - 1) / 2;
  const int nc2 = (BT + 2 * pad2 + 15) >> 4, M1 = 16 * nc2 + 16;
  const int nc1 = (M1 + 2 * pad1 + 15) >> 4;
  const int rowsX = 16 * nc1 + 16, rowsA = 16 * (nc1 > nc2 ? nc1 : nc2);
  constexpr int PF = C == 32 ? 6 : 8;
  if (rowsX * (C / 8) > PF * 256) {
    ptpp_set_error("amp_layer: halo of ks=%d dil=%d exceeds the built prefetch depth", p.ks, p.dil);
    return PTPP_ENOTSUP;
  }
  const size_t smem = (size_t)((PS ? 2 : 1) * rowsX + rowsA) * ROWB + 2 * TAB * sizeof(float);
  if (smem > 160 * 1024) {
    ptpp_set_error("amp_layer: LDS tile too large (%zu B)", smem);
    return PTPP_ENOTSUP;
  }
  auto kern = amp_layer_mfma_kernel<C, BT, MG1, MG2, PS>;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  p.nMT = (p.T + BT - 1) / BT;
  const int64_t ntiles = (int64_t)p.B * p.nMT;
  const int per_cu = (int)((160 * 1024) / smem) > 0 ? (int)((160 * 1024) / smem) : 1;
  const char* g = getenv("PTPP_AMP_GRID");
  int64_t grid = !PS ? ntiles : (g ? atoi(g) : 256 * (int64_t)per_cu);  // PS: as many blocks as fit the 256 CUs at once
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, st, p);
  PTPP_CHECK_LAUNCH("amp_layer_fwd");
  return PTPP_OK;
}

template <typename T, int C, int BT, int MG>
int launch_amp(AmpP& p, hipStream_t st) {
  constexpr int KC = 16 / (int)sizeof(T);
  constexpr int ROWB = (C / KC) * 16;
  const int pad1 = p.dil * (p.ks - 1) / 2, pad2 = (p.ks - 1) / 2;
  const int n_c1 = BT + 2 * pad2 + 12, M1 = (n_c1 + 15) & ~15;
  const int n_x = n_c1 + 2 * pad1 + 12;
  const int rowsX = n_x > M1 ? n_x : M1;
  const int rowsA_1 = M1 + 2 * pad1, rowsA_2 = BT + 2 * pad2;
  const int rowsA = rowsA_1 > rowsA_2 ? rowsA_1 : rowsA_2;
  const size_t smem = (size_t)(rowsX + rowsA) * ROWB;
  if (smem > 160 * 1024) {
    ptpp_set_error("amp_layer: LDS tile too large (%zu B)", smem);
    return PTPP_ENOTSUP;
  }
  auto kern = amp_layer_kernel<T, C, BT, MG>;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  p.nMT = (p.T + BT - 1) / BT;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT)), dim3(256), smem, st, p);
  PTPP_CHECK_LAUNCH("amp_layer_fwd");
  return PTPP_OK;
}

}  // namespace

extern "C" int ptpp_amp_layer_supported(int C, int dtype) {
  if (dtype == PTPP_BF16) return C == 32 || C == 64;
  if (dtype == PTPP_F32) return C == 32 || C == 64;
  return 0;
}

extern "C" int ptpp_amp_layer_fwd(const ptpp_amp_layer_args* a, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->y && a->w1p && a->w2p && a->b1 && a->b2 && a->log_alpha1 && a->log_alpha2,
                 "amp_layer: null pointer");
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->ks >= 1 && (a->ks & 1) && a->ks <= 15 && a->dil >= 1 && a->dil <= 8,
                 "amp_layer: bad shape B=%d T=%d ks=%d dil=%d", a->B, a->T, a->ks, a->dil);
  PTPP_CHECK_ARG(a->x != a->y, "amp_layer: in-place not supported (neighbouring tiles read the halo)");
  PTPP_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->res2 | (uintptr_t)a->w1p | (uintptr_t)a->w2p |
                   (uintptr_t)a->b1 | (uintptr_t)a->b2) & 15) == 0, "amp_layer: pointers must be 16-byte aligned");
  if (!ptpp_amp_layer_supported(a->C, a->dtype)) {
    ptpp_set_error("amp_layer: C=%d dtype=%d not built (C in {32, 64})", a->C, a->dtype);
    return PTPP_ENOTSUP;
  }
  AmpP p;
  p.x = a->x; p.y = a->y; p.res2 = a->res2; p.w1p = a->w1p; p.w2p = a->w2p; p.b1 = a->b1; p.b2 = a->b2;
  p.la1 = a->log_alpha1; p.la2 = a->log_alpha2;
  for (int i = 0; i < 12; ++i) {
    p.up1[i] = a->up1[i]; p.dn1[i] = a->dn1[i]; p.up2[i] = a->up2[i]; p.dn2[i] = a->dn2[i];
  }
  p.B = a->B; p.T = a->T; p.ks = a->ks; p.dil = a->dil;
  p.out_scale = a->out_scale; p.res_scale = a->res_scale;
  p.nMT = 0;
  p.skip = getenv("PTPP_AMP_SKIP") ? atoi(getenv("PTPP_AMP_SKIP")) : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == PTPP_BF16) {
    // The product path is the VALU-FIR kernel (amp_layer_kernel): bit-reproducible and the faster one for C = 64.
    // EXPERIMENTAL, opt-in (PTPP_AMP_VARIANT = mfma | persist): the FIRs on the matrix cores.  Measured
    // (profiles/r02_amp_layer_variants.txt): 5-10 % faster than the VALU form at C = 32, slower at C = 64, and the
    // persistent form is slower than one tile per block (the phases are latency-bound: more resident waves win).
    // It matches the oracle, but whenever two workgroups share a CU its output is NOT run-to-run reproducible
    // (1-ulp bf16 differences in ~1 % of the elements; tools/diag_amp_det*.py; not an LDS-initialisation, barrier
    // or MFMA-result wait-state effect -- cause not found in round 2), which rules it out for the product.
    const char* var = getenv("PTPP_AMP_VARIANT");
    if (var && !strcmp(var, "persist")) {
      if (a->C == 32) return launch_amp_mfma<32, 256, 5, 4, true>(p, st);
      return launch_amp_mfma<64, 128, 3, 2, true>(p, st);
    }
    if (var && !strcmp(var, "mfma")) {
      if (a->C == 32) return launch_amp_mfma<32, 256, 5, 4, false>(p, st);
      return launch_amp_mfma<64, 128, 3, 2, false>(p, st);
    }
    if (a->C == 32) return launch_amp<bf16_raw, 32, 256, 5>(p, st);
    return launch_amp<bf16_raw, 64, 128, 3>(p, st);
  }
  if (a->C == 32) return launch_amp<float, 32, 128, 3>(p, st);
  return launch_amp<float, 64, 64, 2>(p, st);
}
