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
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "amp_layer_fwd")) return PTPP_ELAUNCH;
  p.nMT = (p.T + BT - 1) / BT;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT)), dim3(256), smem, st, p);
  PTPP_CHECK_LAUNCH("amp_layer_fwd");
  return PTPP_OK;
}

}  // namespace

int amp_fused_launch_16bit(const ptpp_amp_layer_args* a, void* stream);  // amp_fused.hip

extern "C" int ptpp_amp_layer_supported(int C, int dtype) {
  if (dtype == PTPP_BF16 || dtype == PTPP_F16) return C == 32 || C == 64;
  if (dtype == PTPP_F32) return C == 32 || C == 64;
  return 0;
}

extern "C" int ptpp_amp_layer_fwd(const ptpp_amp_layer_args* a, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->y && a->b1 && a->b2 && a->log_alpha1 && a->log_alpha2, "amp_layer: null pointer");
  const bool old16 = getenv("PTPP_AMP_OLD") && atoi(getenv("PTPP_AMP_OLD"));  // A/B: the round-2 kernel for 16-bit tensors
  const bool fused16 = (a->dtype == PTPP_BF16 && !old16) || a->dtype == PTPP_F16;
  PTPP_CHECK_ARG(fused16 ? (a->w1s && a->w2s) : (a->w1p && a->w2p), "amp_layer: null weight operand (%s)",
                 fused16 ? "16-bit tensors take the fragment streams w1s / w2s" : "w1p / w2p");
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->ks >= 1 && (a->ks & 1) && a->ks <= 15 && a->dil >= 1 && a->dil <= 8,
                 "amp_layer: bad shape B=%d T=%d ks=%d dil=%d", a->B, a->T, a->ks, a->dil);
  PTPP_CHECK_ARG(a->x != a->y, "amp_layer: in-place not supported (neighbouring tiles read the halo)");
  PTPP_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->res2 | (uintptr_t)a->w1p | (uintptr_t)a->w2p | (uintptr_t)a->w1s | (uintptr_t)a->w2s |
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
  if (fused16) return amp_fused_launch_16bit(a, stream);
  if (a->dtype == PTPP_BF16) {
    // (the FIRs-on-MFMA experiment of round 2 -- not run-to-run reproducible with two workgroups per CU -- left the library in
    //  round 4: tools/experiments/r02_amp_layer_mfma.hip.txt)
    if (a->C == 32) return launch_amp<bf16_raw, 32, 256, 5>(p, st);
    return launch_amp<bf16_raw, 64, 128, 3>(p, st);
  }
  if (a->C == 32) return launch_amp<float, 32, 128, 3>(p, st);
  return launch_amp<float, 64, 64, 2>(p, st);
}
