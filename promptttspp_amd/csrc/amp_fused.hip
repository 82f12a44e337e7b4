// One BigVGAN AMP layer in ONE kernel, 16-bit tensors, second generation (round 5).  Same arithmetic, rounding points and
// results (bit for bit) as amp_layer_kernel (amp_layer.hip, which stays the exact-f32 parity path):
//
//   y = res_scale * x + out_scale * (conv2(snake2(conv1(snake1(x)))) + b2) [+ res2]
//   (vocoders/bigvgan.py:42-47, layers/activations.py:22-44, 74-138)
//
// What changed, and the measurement behind each change (tools/experiments/valu_rate.hip, lds_read_rate.hip,
// profiles/r05_valu_rate.txt, r05_lds_read_rate.txt):
//   * 8 waves per workgroup, two workgroups per CU.  One wave alone issues a VALU instruction every ~4.7 cycles; v_fma_f32 /
//     v_mul_f32 / v_add_f32 retire in 2.4 once two waves of a SIMD feed it.  The Snake is ~50 FMAs per element pair: the
//     4-wave blocks of round 2 ran it at half rate whenever only one of the three resident blocks was in a Snake phase.
//   * LDS rows are PADDED (stride = row bytes + 16), not XOR-swizzled: addresses are affine, the Snake walks its rows with
//     immediate offsets (the swizzle cost ~16 integer instructions of 4.2 cycles per step), and interior tiles take a
//     clamp-free, select-free copy of the loop (v_cndmask_b32 measured at 16-20 cycles; the replicate padding only exists
//     in the first and last tile of an utterance).
//   * every run of the Snake has the same length (uniform control flow), x rows are requested one 6-step group ahead.
//   * conv phases: a wave owns TWO 16-channel fragments of the output and MG row fragments (round 2: all four channel
//     fragments and three row fragments, waves 3,3,3,1 / 3,3,2,0): weight fragments come straight from L2 (1 KiB per
//     fragment per wave per K step through the 64 B/clk texture path: 0.33 KB per MFMA then, 0.25-0.2 now), activation
//     fragments from LDS (0.5 KB per MFMA = half the 256 B/clk the LDS delivers), the row fragments divide evenly.
// Phases of a block (8 waves, __syncthreads between phases):
//   P1  Snake 1 (VALU): global x -> A   P2  conv1 (MFMA): A -> c1 (+ bias) in X
//   P3  Snake 2: X -> A                 P4  conv2; epilogue through an LDS image of the output tile (bias, residual x, running
//                                           AMP-block mean res2, 16-byte coalesced stores)
#include <stdlib.h>
#include <string.h>

#include "ptpp_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// element kinds of the 16-bit tensors
struct EB16 {  // bfloat16
  static __device__ __forceinline__ void unpack(uint32_t r, float& a, float& b) {
    a = __uint_as_float(r << 16);
    b = __uint_as_float(r & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {  // v_cvt_pk_bf16_f32: round to nearest even
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  }
  static __device__ __forceinline__ float unpack1(uint32_t r) { return __uint_as_float(r << 16); }
  static __device__ __forceinline__ uint16_t pack1(float a) { return (uint16_t)pack(a, a); }
  static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};
struct EF16 {  // IEEE half
  static __device__ __forceinline__ void unpack(uint32_t r, float& a, float& b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    const f16x2_t h = __builtin_bit_cast(f16x2_t, r);
    a = (float)h[0];
    b = (float)h[1];
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {  // round to nearest even
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    return __builtin_bit_cast(uint32_t, f16x2_t{(_Float16)a, (_Float16)b});
  }
  static __device__ __forceinline__ float unpack1(uint32_t r) { return (float)__builtin_bit_cast(_Float16, (uint16_t)r); }
  static __device__ __forceinline__ uint16_t pack1(float a) { return __builtin_bit_cast(uint16_t, (_Float16)a); }
  static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};

struct AmpF {
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
  int R1, R2;     // rows per Snake run (phase 1 / phase 3), the same for every thread
  int rowsX;      // rows of the X image (the A image follows it)
  int skip;       // diagnostics (PTPP_AMP_SKIP): bit 0 P1, 1 P2, 2 P3, 3 P4 MFMA loop
  int stagger;    // s_sleep(127) periods the second resident block of every CU waits before its first tile
};

// A wave-uniform value parked in a VGPR the compiler cannot move back to an SGPR: v_fmac_f32 / v_mul_f32 / v_add_f32 with an
// SGPR (or literal) operand issue in 4.2-4.4 cycles, with VGPR operands only in 2.3 (profiles/r05_valu_rate.txt: FMA_S, FMAAK
// against FMA, FMAC) -- the Snake's 48 FMAs per step all multiply by a filter tap.
__device__ __forceinline__ float in_vgpr(float s) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(s));
  return v;
}

__device__ __forceinline__ float snake_val16(float u, float w, float inv) {  // w = e^alpha / (2 pi): v_sin_f32 takes revolutions
  const float s = __builtin_amdgcn_sinf(u * w);
  return fmaf(inv, s * s, u);
}

// One Snake step for a channel pair.  Step K of a 6-step group replaces the oldest x by the row just loaded, pushes
// s[2tp+7], s[2tp+8] into the 12-deep window and (EMIT) writes row tp+1 = sum_j s[2tp-3+j] fdn[j].  (the FMA order is that of
// amp_layer.hip's AMP_SNAKE_STEP: results are bit-identical)
#define AF_STEP_CORE(K, RAW)                                                                \
  {                                                                                         \
    if constexpr (PAIR) E::unpack(RAW, xa[(K) % 6], xb[(K) % 6]);                           \
    else xa[(K) % 6] = E::unpack1(RAW);                                                     \
    float uoa = 0.f, uea = 0.f, uob = 0.f, ueb = 0.f;                                       \
    _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                         \
      const float va = xa[((K) + 1 + a) % 6];                                               \
      uoa = fmaf(va, fup2[10 - 2 * a], uoa);                                                \
      uea = fmaf(va, fup2[11 - 2 * a], uea);                                                \
      if constexpr (PAIR) {                                                                 \
        const float vb = xb[((K) + 1 + a) % 6];                                             \
        uob = fmaf(vb, fup2[10 - 2 * a], uob);                                              \
        ueb = fmaf(vb, fup2[11 - 2 * a], ueb);                                              \
      }                                                                                     \
    }                                                                                       \
    soa = snake_val16(uoa, w0, inv0); sea = snake_val16(uea, w0, inv0);                     \
    if constexpr (PAIR) { sob = snake_val16(uob, w1, inv1); seb = snake_val16(ueb, w1, inv1); } \
  }
#define AF_STEP_PUSH(K)                                                                     \
  {                                                                                         \
    sa[(2 * (K)) % 12] = soa; sa[(2 * (K) + 1) % 12] = sea;                                 \
    if constexpr (PAIR) { sb[(2 * (K)) % 12] = sob; sb[(2 * (K) + 1) % 12] = seb; }         \
  }
#define AF_STEP_DOWN(K, YA, YB)                                                             \
  {                                                                                         \
    float ya = 0.f, yb = 0.f, za = 0.f, zb = 0.f;                                           \
    _Pragma("unroll") for (int j = 0; j < 12; j += 2) {                                     \
      ya = fmaf(sa[(2 * (K) + 2 + j) % 12], fdn[j], ya);                                    \
      za = fmaf(sa[(2 * (K) + 3 + j) % 12], fdn[j + 1], za);                                \
      if constexpr (PAIR) {                                                                 \
        yb = fmaf(sb[(2 * (K) + 2 + j) % 12], fdn[j], yb);                                  \
        zb = fmaf(sb[(2 * (K) + 3 + j) % 12], fdn[j + 1], zb);                              \
      }                                                                                     \
    }                                                                                       \
    YA = ya + za; YB = yb + zb;                                                             \
  }
// one row position of a thread's channel pair (PAIR) or single channel: 4 / 2 bytes at byte offset c2 of the row
#define AF_LD(PTR) (PAIR ? *reinterpret_cast<const uint32_t*>(PTR) : (uint32_t)*reinterpret_cast<const uint16_t*>(PTR))
#define AF_ST(PTR, A, B)                                                        \
  {                                                                             \
    if constexpr (PAIR) *reinterpret_cast<uint32_t*>(PTR) = E::pack(A, B);      \
    else *reinterpret_cast<uint16_t*>(PTR) = E::pack1(A);                       \
  }

// Interior tile: no clamps, no selects.  The thread's run covers dst rows [o0, o0 + R); src row of time t is dst row + 6.
// Step s = 0 .. R + 4 loads src row o0 + 6 + s and (s >= 5) writes dst row o0 + s - 5.
template <typename E, int SS, int S, bool PAIR = true>
__device__ __forceinline__ void snake_fast(const char* src, char* dst, int o0, int R, int c2, const float (&fup2)[12],
                                           const float (&fdn)[12], float w0, float w1, float inv0, float inv1) {
  float xa[6], xb[6], sa[12], sb[12];
  float soa, sea, sob = 0.f, seb = 0.f;
  const char* px = src + o0 * SS + c2;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    if constexpr (PAIR) E::unpack(AF_LD(px + a * SS), xa[a], xb[a]);
    else xa[a] = E::unpack1(AF_LD(px + a * SS));
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
  px += 6 * SS;
  char* pd = dst + (o0 - 5) * S + c2;
  uint32_t cur[6], nxt[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cur[k] = AF_LD(px + k * SS);
  const int NS = R + 5;
  const int G = NS / 6, rem = NS - G * 6;
  // group 0: five warm-up steps, the sixth emits the run's first row
#pragma unroll
  for (int k = 0; k < 6; ++k) nxt[k] = AF_LD(px + (6 + k) * SS);
#define AF_FAST(K, EMIT)                                                          \
  {                                                                               \
    AF_STEP_CORE(K, cur[K])                                                       \
    AF_STEP_PUSH(K)                                                               \
    if (EMIT) {                                                                   \
      float oa, ob;                                                               \
      AF_STEP_DOWN(K, oa, ob)                                                     \
      AF_ST(pd + (K) * S, oa, ob)                                                 \
    }                                                                             \
  }
  AF_FAST(0, false) AF_FAST(1, false) AF_FAST(2, false) AF_FAST(3, false) AF_FAST(4, false) AF_FAST(5, true)
  for (int g = 1; g < G; ++g) {
    px += 6 * SS;
    pd += 6 * S;
#pragma unroll
    for (int k = 0; k < 6; ++k) cur[k] = nxt[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) nxt[k] = AF_LD(px + (6 + k) * SS);
    AF_FAST(0, true) AF_FAST(1, true) AF_FAST(2, true) AF_FAST(3, true) AF_FAST(4, true) AF_FAST(5, true)
  }
  if (rem) {  // (uniform: R is the same for every thread)
    pd += 6 * S;
#pragma unroll
    for (int k = 0; k < 6; ++k) cur[k] = nxt[k];
    AF_FAST(0, true)
    if (rem > 1) AF_FAST(1, true)
    if (rem > 2) AF_FAST(2, true)
    if (rem > 3) AF_FAST(3, true)
    if (rem > 4) AF_FAST(4, true)
  }
#undef AF_FAST
}

// First / last tiles of an utterance (and utterances shorter than a tile): rows read at clamp(t, 0, T-1), s read at
// clamp(m, 0, 2T-1), rows outside [0, T) written as zeros (the conv's zero padding).  dst rows [o0, o0 + n).
template <typename E, int SS, int S, bool PAIR = true>
__device__ __forceinline__ void snake_edge(const char* src, char* dst, int tsrc0, int tdst0, int o0, int n, int Tlen, int c2,
                                           const float (&fup2)[12], const float (&fdn)[12], float w0, float w1, float inv0,
                                           float inv1) {
  auto ldx = [&](int t, float& a, float& b) {
    const int row = min(max(t, 0), Tlen - 1) - tsrc0;
    if constexpr (PAIR) E::unpack(AF_LD(src + (int64_t)row * SS + c2), a, b);
    else { a = E::unpack1(AF_LD(src + (int64_t)row * SS + c2)); b = 0.f; }
  };
  auto st = [&](int i, float a, float b) { AF_ST(dst + i * S + c2, a, b) };
  const int ta = tdst0 + o0, tb = ta + n;
  const int tv0 = max(ta, 0), tv1 = min(tb, Tlen);
  for (int t = ta; t < min(tb, tv0); ++t) st(t - tdst0, 0.f, 0.f);
  for (int t = max(ta, tv1); t < tb; ++t) st(t - tdst0, 0.f, 0.f);
  if (tv1 <= tv0) return;
  float xa[6], xb[6], sa[12], sb[12];
  float soa, sea, sob = 0.f, seb = 0.f;
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
    pa = snake_val16(ua, w0, inv0);
    pb = snake_val16(ub, w1, inv1);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) { sa[i] = pa; sb[i] = pb; }
#pragma unroll
  for (int a = 0; a < 6; ++a) ldx(tp0 + a, xa[a], xb[a]);
  const int mlast = 2 * Tlen - 1;
#define AF_EDGE(K)                                                                          \
  {                                                                                         \
    const int tp = tg + (K);                                                                \
    const int rowc = min(max(tp + 6, 0), Tlen - 1) - tsrc0;                                 \
    const uint32_t raw = AF_LD(src + (int64_t)rowc * SS + c2);                              \
    AF_STEP_CORE(K, raw)                                                                    \
    const int m1 = 2 * tp + 7;                                                              \
    if (m1 > mlast) { soa = sa[(2 * (K) + 11) % 12]; sob = sb[(2 * (K) + 11) % 12]; }       \
    if (m1 + 1 > mlast) { sea = soa; seb = sob; }                                           \
    AF_STEP_PUSH(K)                                                                         \
    const int t = tp + 1;                                                                   \
    if (t >= tv0 && t < tv1) {                                                              \
      float oa, ob;                                                                         \
      AF_STEP_DOWN(K, oa, ob)                                                               \
      st(t - tdst0, oa, ob);                                                                \
    }                                                                                       \
  }
  for (int tg = tp0; tg + 1 < tv1; tg += 6) {
    AF_EDGE(0) AF_EDGE(1) AF_EDGE(2) AF_EDGE(3) AF_EDGE(4) AF_EDGE(5)
  }
#undef AF_EDGE
}

// out channel held by MFMA "A" row i of fragment f (a lane's accumulators of a fragment PAIR are one 16-byte chunk)
__device__ __forceinline__ int frag_channel16(int f, int i) { return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3); }

// Implicit-GEMM conv over the LDS-resident activation image for the wave's MG row fragments x the fragment pair h:
//   acc[mi][u] = sum_{j, ci} W[co(2h + u, .), j, ci] * act[(mf0 + mi) * 16 + lr + j * dil][ci]
// Weight fragments straight from global memory into registers, requested two K steps ahead, from the FRAGMENT STREAM
// (ptpp_amp_pack_wstream): vector (s * C/16 + f) * 64 + lane of the stream is lane's 16 bytes of the MFMA "A" fragment f of
// K step s = (tap j, 32-channel block kc) -- a wave reads 2 KiB contiguous per step.  (Reading the same fragments from the
// [Cout][ks][Cin] operand touches 32 separate 64-byte pieces per wave-load, half a cache line each, which the 32 KiB L1 has
// evicted before the next K step asks for the other half: measured 2.8 ms of a 3.9 ms layer at C = 64, k = 11, against
// 0.98 ms with the loads removed -- profiles/r05_amp_phases.txt.)  Activation fragments from LDS.
template <typename E, int C, int S, int MG>
__device__ __forceinline__ void conv_pair(const char* act, const uint4* __restrict__ ws, int ks, int dil, int mf0, int nmf,
                                          int h, int lane, f32x4 (&acc)[MG][2], int dbg = 0) {
  constexpr int NKC = C / 32;
  constexpr int NF = C / 16;
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MG; ++mi) { acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (dbg & 128) __builtin_amdgcn_s_setprio(3);
  const uint4* wr = ws + 2 * h * 64 + lane;  // + s * NF * 64 per step; the pair's second fragment 64 vectors on
  const int steps = ks * NKC;
  const char* abase[MG];
#pragma unroll
  for (int mi = 0; mi < MG; ++mi) abase[mi] = act + ((mf0 + (mi < nmf ? mi : 0)) * 16 + lr) * S + lg * 16;
  const int dS = dil * S;
  // three weight register sets in rotation, no moves: step s reads set s % 3 and requests step s + 2 into set (s + 2) % 3.
  // The loads are UNCONDITIONAL (past the end the last step's fragments are requested again): with a branch around them the
  // compiler's vmcnt bookkeeping gives up at the merge and drains the queue (s_waitcnt vmcnt(0)) every step.
  uint4 w[3][2];
  const int last = steps - 1;
  w[0][0] = wr[0];
  w[0][1] = wr[64];
  w[1][0] = wr[min(1, last) * NF * 64];
  w[1][1] = wr[min(1, last) * NF * 64 + 64];
#define AF_CONV_STEP(CUR, NXT, SI)                                                                  \
  {                                                                                                 \
    const int s_ = (SI);                                                                            \
    const int sn_ = (dbg & 16) ? 0 : min(s_ + 2, last);                                             \
    w[NXT][0] = wr[sn_ * NF * 64];                                                                  \
    w[NXT][1] = wr[sn_ * NF * 64 + 64];                                                             \
    const int off = (dbg & 32) ? 0 : (s_ / NKC) * dS + (s_ % NKC) * 64;                             \
    uint4 xf[MG];                                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < MG; ++mi) xf[mi] = *reinterpret_cast<const uint4*>(abase[mi] + off); \
    _Pragma("unroll") for (int mi = 0; mi < MG; ++mi) {                                             \
      E::mma(acc[mi][0], w[CUR][0], xf[mi]);                                                        \
      E::mma(acc[mi][1], w[CUR][1], xf[mi]);                                                        \
    }                                                                                               \
  }
  int s = 0;
  for (; s + 3 <= steps; s += 3) {
    AF_CONV_STEP(0, 2, s)
    AF_CONV_STEP(1, 0, s + 1)
    AF_CONV_STEP(2, 1, s + 2)
  }
  if (s < steps) {
    AF_CONV_STEP(0, 2, s)
    if (s + 1 < steps) AF_CONV_STEP(1, 0, s + 1)
  }
#undef AF_CONV_STEP
  if (dbg & 128) __builtin_amdgcn_s_setprio(0);
}

// [Cout][ks][Cin] operand -> fragment stream (one 16-byte vector per thread)
__global__ __launch_bounds__(256) void amp_wstream_kernel(const uint16_t* __restrict__ wp, uint4* __restrict__ out, int C, int ks) {
  const int NKC = C / 32, NF = C / 16;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= ks * NKC * NF * 64) return;
  const int lane = q & 63, f = (q >> 6) % NF, s = (q >> 6) / NF;
  const int lr = lane & 15, lg = lane >> 4;
  const int j = s / NKC, kc = s - j * NKC;
  const int co = frag_channel16(f, lr);
  out[q] = *reinterpret_cast<const uint4*>(wp + ((int64_t)co * ks + j) * C + kc * 32 + lg * 8);
}

template <typename E, int C, int BT, int S, int MG1, int MG2, bool PAIR>
__global__ __launch_bounds__(512, 4) void amp_fused_kernel(const AmpF p) {
  constexpr int NT = 512;
  constexpr int NCH = C / 8;          // 16-byte chunks per row
  constexpr int CP = PAIR ? C / 2 : C;  // Snake work items per row: channel pairs, or single channels (twice the run length:
                                        // the 5 warm-up steps of a run weigh half as much)
  constexpr int CB = PAIR ? 4 : 2;    // bytes of a work item in a row
  constexpr int NRUN = NT / CP;       // row runs per Snake phase
  constexpr int WN = C / 32;          // waves along the output channels (a wave owns 32 of them)
  constexpr int WM = 8 / WN;          // waves along the rows
  static_assert(NT % CP == 0 && BT % 16 == 0 && S % 16 == 0 && S >= C * 2, "geometry");
  static_assert(MG2 * WM * 16 >= BT, "conv2: one fragment group per wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ks = p.ks, dil = p.dil, Tlen = p.T;
  const int pad1 = dil * (ks - 1) / 2, pad2 = (ks - 1) / 2;
  const int n_c1 = BT + 2 * pad2 + 12;          // conv1 output rows snake 2 reads
  const int M1 = (n_c1 + 15) & ~15;             // ... rounded up to whole MFMA fragments
  const int n_a1 = n_c1 + 2 * pad1;             // snake-1 rows conv1 reads (without the fragment overhang)
  char* Xs = smem;
  char* As = smem + p.rowsX * S;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Two workgroups share a CU and walk the same phases: started together they stay in lock step (both in a VALU phase, then
  // both in an MFMA phase).  The second block of every CU starts half a tile late; the offset persists, one block's Snake
  // then runs beside the other's conv.
  const int slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) >> 1;  // HW_ID.wave_id / 2: 0 = first, 1 = second block of the CU
  if (p.stagger && slot && blockIdx.x < 512)
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  int skip = p.skip;
  if (skip & 64) skip = slot ? (skip | 5) : (skip | 10);  // experiment: one block of the CU only Snakes, the other only convs
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid % p.nMT, b = lid / p.nMT;
  const int t0 = mt * BT;
  const int tc0 = t0 - pad2 - 6;     // time of c1 row 0
  const int ta0 = tc0 - pad1;        // time of a1 row 0
  const int tx0 = ta0 - 6;           // time of X row 0
  // interior: every row a Snake-1 run reads (its overhang and the prefetched group included) lies inside the utterance
  const bool interior = tx0 >= 0 && tx0 + NRUN * p.R1 + 17 <= Tlen;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)b * Tlen * C;

  const int cpair = tid % CP, run = tid / CP;
  constexpr float WSC = 0.15915494309189535f;  // v_sin_f32 takes revolutions
  float ea0, ea1, inv0, inv1;
  {
    const float a0 = __expf(p.la1[PAIR ? 2 * cpair : cpair]), a1 = __expf(p.la1[PAIR ? 2 * cpair + 1 : cpair]);
    ea0 = a0 * WSC; ea1 = a1 * WSC; inv0 = 1.0f / (a0 + 1e-9f); inv1 = 1.0f / (a1 + 1e-9f);
  }

  // ---- P1: snake 1: global x -> A (a1 rows [0, n_a1); time ta0 + i).  No x image and no load phase: a thread's rows come
  // straight from global memory, one 6-step group ahead of their use (a wave-load is two full 128-byte row pieces) ----
  if (!(skip & 1)) {
    float fu[12], fd[12];  // the taps in VECTOR registers (in_vgpr)
#pragma unroll
    for (int i = 0; i < 12; ++i) { fu[i] = in_vgpr(2.0f * p.up1[i]); fd[i] = in_vgpr(p.dn1[i]); }
    const int o0 = run * p.R1;
    const char* src = reinterpret_cast<const char*>(xb) + (int64_t)tx0 * (C * 2);
    if (interior) {
      snake_fast<E, C * 2, S, PAIR>(src, As, o0, p.R1, cpair * CB, fu, fd, ea0, ea1, inv0, inv1);
    } else {
      const int n = min(p.R1, n_a1 - o0);
      if (n > 0) snake_edge<E, C * 2, S, PAIR>(src, As, tx0, ta0, o0, n, Tlen, cpair * CB, fu, fd, ea0, ea1, inv0, inv1);
    }
  }
  __syncthreads();

  const int wn = wave % WN, wm = wave / WN;
  const int lr = lane & 15, lg = lane >> 4;
  // ---- P2: conv1 (dilated): A -> c1 rows [0, M1) in X ----
  if (!(skip & 2)) {
    const int nfr = M1 / 16;
    const int per = (nfr + WM - 1) / WM;  // <= MG1 (host)
    const int mf0 = wm * per, nmf = min(per, nfr - mf0);
    if (nmf > 0) {
      f32x4 acc[MG1][2];
      conv_pair<E, C, S, MG1>(As, reinterpret_cast<const uint4*>(p.w1p), ks, dil, mf0, nmf, wn, lane, acc, skip);
      const int co = wn * 32 + lg * 8;
      const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b1 + co), bB = *reinterpret_cast<const f32x4*>(p.b1 + co + 4);
#pragma unroll
      for (int mi = 0; mi < MG1; ++mi) {
        if (mi < nmf) {
          const int row = (mf0 + mi) * 16 + lr;
          const f32x4 v0 = acc[mi][0] + bA, v1 = acc[mi][1] + bB;
          uint4 o;
          o.x = E::pack(v0[0], v0[1]); o.y = E::pack(v0[2], v0[3]);
          o.z = E::pack(v1[0], v1[1]); o.w = E::pack(v1[2], v1[3]);
          *reinterpret_cast<uint4*>(Xs + row * S + (wn * 4 + lg) * 16) = o;
        }
      }
    }
  }
  {
    const float a0 = __expf(p.la2[PAIR ? 2 * cpair : cpair]), a1 = __expf(p.la2[PAIR ? 2 * cpair + 1 : cpair]);
    ea0 = a0 * WSC; ea1 = a1 * WSC; inv0 = 1.0f / (a0 + 1e-9f); inv1 = 1.0f / (a1 + 1e-9f);
  }
  __syncthreads();

  // ---- P3: snake 2: c1 (X) -> a2 rows [0, BT + 2 pad2) in A; time t0 - pad2 + i ----
  if (!(skip & 4)) {
    const int n_a2 = BT + 2 * pad2;
    float fu[12], fd[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { fu[i] = in_vgpr(2.0f * p.up2[i]); fd[i] = in_vgpr(p.dn2[i]); }
    const int o0 = run * p.R2;
    if (interior) {
      snake_fast<E, S, S, PAIR>(Xs, As, o0, p.R2, cpair * CB, fu, fd, ea0, ea1, inv0, inv1);
    } else {
      const int n = min(p.R2, n_a2 - o0);
      if (n > 0) snake_edge<E, S, S, PAIR>(Xs, As, tc0, t0 - pad2, o0, n, Tlen, cpair * CB, fu, fd, ea0, ea1, inv0, inv1);
    }
  }
  __syncthreads();

  // ---- P4: conv2: A -> y (+ bias, residual x, running mean res2) through an LDS image of the output tile in X ----
  {
    constexpr int nfr = BT / 16;
    constexpr int per = (nfr + WM - 1) / WM;
    static_assert(per <= MG2, "conv2 fragment groups");
    constexpr int NRV = (BT * NCH + NT - 1) / NT;  // 16-byte vectors of the tile per thread
    uint16_t* yb = reinterpret_cast<uint16_t*>(p.y) + (int64_t)b * Tlen * C;
    const uint16_t* r2b = p.res2 ? reinterpret_cast<const uint16_t*>(p.res2) + (int64_t)b * Tlen * C : nullptr;
    const float osc = p.out_scale, rsc = p.res_scale;
    uint4 rx[NRV];
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = t0 + r;
      rx[i] = (idx < BT * NCH && t < Tlen) ? *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + ch * 8) : make_uint4(0, 0, 0, 0);
    }
    const int mf0 = wm * per, nmf = min(per, nfr - mf0);
    const int co = wn * 32 + lg * 8;
    f32x4 acc[MG2][2];
    uint4 q2[MG2];
    if (nmf > 0) {
      if (r2b) {  // (two of the nine layers of a stage: the running mean of the AMP blocks)
#pragma unroll
        for (int mi = 0; mi < MG2; ++mi) {
          const int t = t0 + (mf0 + mi) * 16 + lr;
          q2[mi] = (mi < nmf && t < Tlen) ? *reinterpret_cast<const uint4*>(r2b + (int64_t)t * C + co) : make_uint4(0, 0, 0, 0);
        }
      }
      if (!(skip & 8)) conv_pair<E, C, S, MG2>(As, reinterpret_cast<const uint4*>(p.w2p), ks, 1, mf0, nmf, wn, lane, acc, skip);
      else {
#pragma unroll
        for (int mi = 0; mi < MG2; ++mi) { acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mi][1] = acc[mi][0]; }
      }
    }
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      if (idx < BT * NCH) *reinterpret_cast<uint4*>(Xs + r * S + ch * 16) = rx[i];
    }
    __syncthreads();
    if (nmf > 0) {
      const f32x4 bA = *reinterpret_cast<const f32x4*>(p.b2 + co), bB = *reinterpret_cast<const f32x4*>(p.b2 + co + 4);
#pragma unroll
      for (int mi = 0; mi < MG2; ++mi) {
        if (mi < nmf) {
          const int r = (mf0 + mi) * 16 + lr;
          f32x4 v0 = (acc[mi][0] + bA) * osc, v1 = (acc[mi][1] + bB) * osc;
          uint4* slot = reinterpret_cast<uint4*>(Xs + r * S + (wn * 4 + lg) * 16);
          const uint4 rr = *slot;
          float e0, e1;
          E::unpack(rr.x, e0, e1); v0[0] += e0 * rsc; v0[1] += e1 * rsc;
          E::unpack(rr.y, e0, e1); v0[2] += e0 * rsc; v0[3] += e1 * rsc;
          E::unpack(rr.z, e0, e1); v1[0] += e0 * rsc; v1[1] += e1 * rsc;
          E::unpack(rr.w, e0, e1); v1[2] += e0 * rsc; v1[3] += e1 * rsc;
          if (r2b) {
            const uint4 q = q2[mi];
            E::unpack(q.x, e0, e1); v0[0] += e0; v0[1] += e1;
            E::unpack(q.y, e0, e1); v0[2] += e0; v0[3] += e1;
            E::unpack(q.z, e0, e1); v1[0] += e0; v1[1] += e1;
            E::unpack(q.w, e0, e1); v1[2] += e0; v1[3] += e1;
          }
          uint4 o;
          o.x = E::pack(v0[0], v0[1]); o.y = E::pack(v0[2], v0[3]);
          o.z = E::pack(v1[0], v1[1]); o.w = E::pack(v1[2], v1[3]);
          *slot = o;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = t0 + r;
      if (idx < BT * NCH && t < Tlen)
        *reinterpret_cast<uint4*>(yb + (int64_t)t * C + ch * 8) = *reinterpret_cast<const uint4*>(Xs + r * S + ch * 16);
    }
  }
}

// ---- wide stages (C = 128, 256): the anti-aliased Snake applied while the conv's input tile is staged ----------------------
//   y = res_scale * res + out_scale * (conv(snake(x)) + bias) [+ res2]      (bigvgan.py:42-47, one of the two convs of a layer)
// The Snake reads x STRAIGHT FROM GLOBAL MEMORY (a wave-load is two 128-byte row pieces, fully used; the next 6-step group is
// requested one group ahead) and writes its output rows into the LDS image the MFMAs read: no x image, no separate Snake launch,
// the activated tensor never exists in HBM (round 4: 37 aa_snake launches x 822 MB).  Full fusion of the layer (amp_fused_kernel)
// does not pay here: the conv1 halo would be recomputed on the matrix cores (x 1.5 at 64-row tiles) and the weights (0.4-1.4 MB per
// conv) stream from L2 either way.
struct SnkP {
  const void* x;
  void* y;
  const void* res;
  const void* res2;
  const void* ws;      // fragment stream of the conv weights
  const float* bias;
  const float* la;
  float up[12], dn[12];
  int B, T, ks, dil;
  float out_scale, res_scale;
  int nMT, R, rowsA;
  int skip;
};

template <typename E, int C, int BT, int S, int MG, bool PAIR>
__global__ __launch_bounds__(512, 4) void snake_conv_kernel(const SnkP p) {
  constexpr int NT = 512;
  constexpr int NCH = C / 8;
  constexpr int CP = PAIR ? C / 2 : C;  // Snake work items per row (channel pairs / single channels)
  constexpr int CB = PAIR ? 4 : 2;
  constexpr int NRUN = NT / CP;       // row runs of the Snake
  constexpr int WN = C / 32 > 8 ? 8 : C / 32;   // waves along the output channels
  constexpr int WM = 8 / WN;          // waves along the rows
  constexpr int NPW = C / 32 / WN;    // channel-fragment pairs per wave (1)
  static_assert(NPW == 1 && MG * WM * 16 == BT && S % 16 == 0 && S >= C * 2, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  const int ks = p.ks, dil = p.dil, Tlen = p.T;
  const int pad = dil * (ks - 1) / 2;
  const int n_a = BT + 2 * pad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid % p.nMT, b = lid / p.nMT;
  const int t0 = mt * BT;
  const int ta0 = t0 - pad;          // time of A row 0
  const int tx0 = ta0 - 6;           // time of the first x row the Snake reads
  // interior: every row a run reads (its overhang and the prefetched group included) lies inside the utterance
  const bool interior = tx0 >= 0 && tx0 + NRUN * p.R + 17 <= Tlen;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)b * Tlen * C;
  const int cpair = tid % CP, run = tid / CP;
  constexpr float WSC = 0.15915494309189535f;
  float ea0, ea1, inv0, inv1;
  {
    const float a0 = __expf(p.la[PAIR ? 2 * cpair : cpair]), a1 = __expf(p.la[PAIR ? 2 * cpair + 1 : cpair]);
    ea0 = a0 * WSC; ea1 = a1 * WSC; inv0 = 1.0f / (a0 + 1e-9f); inv1 = 1.0f / (a1 + 1e-9f);
  }
  // ---- P1: Snake: global x -> A rows [0, n_a) ----
  if (!(p.skip & 1)) {
    float fu[12], fd[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { fu[i] = in_vgpr(2.0f * p.up[i]); fd[i] = in_vgpr(p.dn[i]); }
    const int o0 = run * p.R;
    const char* src = reinterpret_cast<const char*>(xb) + (int64_t)tx0 * (C * 2);
    if (interior) {
      snake_fast<E, C * 2, S, PAIR>(src, As, o0, p.R, cpair * CB, fu, fd, ea0, ea1, inv0, inv1);
    } else {
      const int n = min(p.R, n_a - o0);
      if (n > 0) snake_edge<E, C * 2, S, PAIR>(src, As, tx0, ta0, o0, n, Tlen, cpair * CB, fu, fd, ea0, ea1, inv0, inv1);
    }
  }
  // the residual rows of the tile, row-contiguous, parked in the output image after the MFMAs; requested before them where
  // the registers allow (4 row fragments per wave), after them with 6 (the other resident workgroup covers the round trip)
  constexpr int NRV = BT * NCH / NT;
  static_assert(NRV * NT == BT * NCH, "tile vectors must divide over the threads");
  constexpr bool LATE = MG > 4;
  uint4 rx[NRV];
  auto load_res = [&]() {
    const uint16_t* rb = reinterpret_cast<const uint16_t*>(p.res) + (int64_t)b * Tlen * C;
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      const int t = t0 + r;
      rx[i] = t < Tlen ? *reinterpret_cast<const uint4*>(rb + (int64_t)t * C + ch * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  if (!LATE && p.res) load_res();
  __syncthreads();

  // ---- P2: conv on the matrix cores ----
  const int wn = wave % WN, wm = wave / WN;
  const int lr = lane & 15, lg = lane >> 4;
  const int mf0 = wm * MG;
  const int co = wn * 32 + lg * 8;
  const uint16_t* r2b = p.res2 ? reinterpret_cast<const uint16_t*>(p.res2) + (int64_t)b * Tlen * C : nullptr;
  uint4 q2[MG];
  auto load_res2 = [&]() {
#pragma unroll
    for (int mi = 0; mi < MG; ++mi) {
      const int t = t0 + (mf0 + mi) * 16 + lr;
      q2[mi] = t < Tlen ? *reinterpret_cast<const uint4*>(r2b + (int64_t)t * C + co) : make_uint4(0, 0, 0, 0);
    }
  };
  if (!LATE && r2b) load_res2();
  f32x4 acc[MG][2];
  if (!(p.skip & 2)) conv_pair<E, C, S, MG>(As, reinterpret_cast<const uint4*>(p.ws), ks, dil, mf0, MG, wn, lane, acc, p.skip);
  else {
#pragma unroll
    for (int mi = 0; mi < MG; ++mi) { acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mi][1] = acc[mi][0]; }
  }
  if (LATE && p.res) load_res();
  if (LATE && r2b) load_res2();
  __syncthreads();  // every wave is done with the A image: it becomes the output tile

  // ---- P3: epilogue through the LDS image of the output tile ----
  if (p.res) {
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / NCH, ch = idx - r * NCH;
      *reinterpret_cast<uint4*>(As + r * S + ch * 16) = rx[i];
    }
    __syncthreads();
  }
  {
    const float osc = p.out_scale, rsc = p.res_scale;
    const f32x4 bA = *reinterpret_cast<const f32x4*>(p.bias + co), bB = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
#pragma unroll
    for (int mi = 0; mi < MG; ++mi) {
      const int r = (mf0 + mi) * 16 + lr;
      f32x4 v0 = (acc[mi][0] + bA) * osc, v1 = (acc[mi][1] + bB) * osc;
      uint4* slot = reinterpret_cast<uint4*>(As + r * S + (wn * 4 + lg) * 16);
      float e0, e1;
      if (p.res) {
        const uint4 rr = *slot;
        E::unpack(rr.x, e0, e1); v0[0] += e0 * rsc; v0[1] += e1 * rsc;
        E::unpack(rr.y, e0, e1); v0[2] += e0 * rsc; v0[3] += e1 * rsc;
        E::unpack(rr.z, e0, e1); v1[0] += e0 * rsc; v1[1] += e1 * rsc;
        E::unpack(rr.w, e0, e1); v1[2] += e0 * rsc; v1[3] += e1 * rsc;
      }
      if (r2b) {
        const uint4 q = q2[mi];
        E::unpack(q.x, e0, e1); v0[0] += e0; v0[1] += e1;
        E::unpack(q.y, e0, e1); v0[2] += e0; v0[3] += e1;
        E::unpack(q.z, e0, e1); v1[0] += e0; v1[1] += e1;
        E::unpack(q.w, e0, e1); v1[2] += e0; v1[3] += e1;
      }
      uint4 o;
      o.x = E::pack(v0[0], v0[1]); o.y = E::pack(v0[2], v0[3]);
      o.z = E::pack(v1[0], v1[1]); o.w = E::pack(v1[2], v1[3]);
      *slot = o;
    }
  }
  __syncthreads();
  uint16_t* yb = reinterpret_cast<uint16_t*>(p.y) + (int64_t)b * Tlen * C;
#pragma unroll
  for (int i = 0; i < NRV; ++i) {
    const int idx = tid + i * NT;
    const int r = idx / NCH, ch = idx - r * NCH;
    const int t = t0 + r;
    if (t < Tlen) *reinterpret_cast<uint4*>(yb + (int64_t)t * C + ch * 8) = *reinterpret_cast<const uint4*>(As + r * S + ch * 16);
  }
}

template <int C, int BT, int S, bool PAIR>
size_t snake_conv_geometry(SnkP& p) {
  constexpr int NRUN = 512 / (PAIR ? C / 2 : C);
  const int pad = p.dil * (p.ks - 1) / 2;
  const int n_a = BT + 2 * pad;
  p.R = (n_a + NRUN - 1) / NRUN;
  int rowsA = n_a > NRUN * p.R ? n_a : NRUN * p.R;
  if (rowsA < BT) rowsA = BT;
  p.rowsA = rowsA;
  return (size_t)rowsA * S;
}

template <typename E, int C, int BT, int S, int MG, bool PAIR>
int launch_snake_conv(SnkP& p, hipStream_t st) {
  const size_t smem = snake_conv_geometry<C, BT, S, PAIR>(p);
  if (smem > 160 * 1024) {
    ptpp_set_error("snake_conv: LDS tile too large (%zu B)", smem);
    return PTPP_ENOTSUP;
  }
  auto kern = snake_conv_kernel<E, C, BT, S, MG, PAIR>;
  if (smem > 64 * 1024 && !lds_limit_raised(reinterpret_cast<const void*>(kern))) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      ptpp_set_error("snake_conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return PTPP_ELAUNCH;
    }
    lds_limit_mark(reinterpret_cast<const void*>(kern));
  }
  p.nMT = (p.T + BT - 1) / BT;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT)), dim3(512), smem, st, p);
  PTPP_CHECK_LAUNCH("snake_conv1d_fwd");
  return PTPP_OK;
}

// ---- act_post + conv_post + tanh (vocoders/bigvgan.py:129-131): the last anti-aliased Snake while conv_post's input is staged --
//   y[b, t] = tanh(bias + sum_{j < ks, c < C} w[j, c] * snake(x)[b, t + j - ks/2, c])        (zero padding), y f32
struct SnkPostP {
  const void* x;
  float* y;
  const float* w;      // (ks, C) f32
  const float* la;
  float up[12], dn[12];
  float bias;
  int B, T, ks;
  int nMT, R, rowsA;
};

template <typename E, int C, int BT, int S>
__global__ __launch_bounds__(512, 4) void snake_post_kernel(const SnkPostP p) {
  constexpr int NT = 512;
  constexpr int NRUN = NT / C;  // one channel per thread
  static_assert(BT == NT && C % 8 == 0 && S % 16 == 0 && S >= C * 2, "one output sample per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  float* wsm = reinterpret_cast<float*>(smem + p.rowsA * S);
  const int ks = p.ks, Tlen = p.T;
  const int pad = ks / 2;
  const int n_a = BT + 2 * pad;
  const int tid = threadIdx.x;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid % p.nMT, b = lid / p.nMT;
  const int t0 = mt * BT;
  const int ta0 = t0 - pad, tx0 = ta0 - 6;
  const bool interior = tx0 >= 0 && tx0 + NRUN * p.R + 17 <= Tlen;
  const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)b * Tlen * C;
  const int cidx = tid % C, run = tid / C;
  constexpr float WSC = 0.15915494309189535f;
  const float a0 = __expf(p.la[cidx]);
  const float ea = a0 * WSC, iv = 1.0f / (a0 + 1e-9f);
  for (int i = tid; i < ks * C; i += NT) wsm[i] = p.w[i];
  {
    float fu[12], fd[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { fu[i] = in_vgpr(2.0f * p.up[i]); fd[i] = in_vgpr(p.dn[i]); }
    const int o0 = run * p.R;
    const char* src = reinterpret_cast<const char*>(xb) + (int64_t)tx0 * (C * 2);
    if (interior) {
      snake_fast<E, C * 2, S, false>(src, As, o0, p.R, cidx * 2, fu, fd, ea, ea, iv, iv);
    } else {
      const int n = min(p.R, n_a - o0);
      if (n > 0) snake_edge<E, C * 2, S, false>(src, As, tx0, ta0, o0, n, Tlen, cidx * 2, fu, fd, ea, ea, iv, iv);
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  float acc = p.bias;
  for (int j = 0; j < ks; ++j) {
    const char* row = As + (tid + j) * S;
    const float* wr = wsm + j * C;
#pragma unroll
    for (int q = 0; q < C / 8; ++q) {
      const uint4 v = *reinterpret_cast<const uint4*>(row + q * 16);
      const f32x4 wa = *reinterpret_cast<const f32x4*>(wr + q * 8), wb = *reinterpret_cast<const f32x4*>(wr + q * 8 + 4);
      float e0, e1;
      E::unpack(v.x, e0, e1); acc = fmaf(e0, wa[0], acc); acc = fmaf(e1, wa[1], acc);
      E::unpack(v.y, e0, e1); acc = fmaf(e0, wa[2], acc); acc = fmaf(e1, wa[3], acc);
      E::unpack(v.z, e0, e1); acc = fmaf(e0, wb[0], acc); acc = fmaf(e1, wb[1], acc);
      E::unpack(v.w, e0, e1); acc = fmaf(e0, wb[2], acc); acc = fmaf(e1, wb[3], acc);
    }
  }
  if (t < Tlen) p.y[(int64_t)b * Tlen + t] = tanhf(acc);
}

// LDS geometry of one block (shared by the size query and the launch)
template <int C, int BT, int S, bool PAIR>
size_t amp_fused_geometry(AmpF& p) {
  constexpr int NRUN = 512 / (PAIR ? C / 2 : C);
  const int pad1 = p.dil * (p.ks - 1) / 2, pad2 = (p.ks - 1) / 2;
  const int n_c1 = BT + 2 * pad2 + 12, M1 = (n_c1 + 15) & ~15;
  const int n_a1 = n_c1 + 2 * pad1, n_a2 = BT + 2 * pad2;
  p.R1 = (n_a1 + NRUN - 1) / NRUN;
  p.R2 = (n_a2 + NRUN - 1) / NRUN;
  // the X image holds c1 (M1 rows), then the output tile.  Interior tiles walk NRUN equal runs: the last run of Snake 2 may
  // overhang (rows it reads: up to o0 + R + 16 with the prefetched group; rows it writes: up to NRUN * R)
  int rowsX = M1;
  const int need2 = NRUN * p.R2 + 17;
  if (rowsX < need2) rowsX = need2;
  if (rowsX < BT) rowsX = BT;
  int rowsA = M1 + 2 * pad1;
  if (rowsA < n_a2) rowsA = n_a2;
  if (rowsA < NRUN * p.R1) rowsA = NRUN * p.R1;
  if (rowsA < NRUN * p.R2) rowsA = NRUN * p.R2;
  p.rowsX = rowsX;
  return (size_t)(rowsX + rowsA) * S;
}

template <typename E, int C, int BT, int S, int MG1, int MG2, bool PAIR>
int launch_amp_fused(AmpF& p, hipStream_t st) {
  constexpr int WM = 8 / (C / 32);
  const int pad2 = (p.ks - 1) / 2;
  const int n_c1 = BT + 2 * pad2 + 12, M1 = (n_c1 + 15) & ~15;
  if ((M1 / 16 + WM - 1) / WM > MG1) {
    ptpp_set_error("amp_fused: ks=%d needs %d conv1 row fragments per wave (built for %d)", p.ks, (M1 / 16 + WM - 1) / WM, MG1);
    return PTPP_ENOTSUP;
  }
  size_t smem = amp_fused_geometry<C, BT, S, PAIR>(p);
  if (getenv("PTPP_AMP_ONE_BLOCK")) smem = 100 * 1024;  // experiment: one workgroup per CU
  if (smem > 160 * 1024) {
    ptpp_set_error("amp_fused: LDS tile too large (%zu B)", smem);
    return PTPP_ENOTSUP;
  }
  auto kern = amp_fused_kernel<E, C, BT, S, MG1, MG2, PAIR>;
  if (smem > 64 * 1024 && !lds_limit_raised(reinterpret_cast<const void*>(kern))) {  // once per (device, kernel)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      ptpp_set_error("amp_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return PTPP_ELAUNCH;
    }
    lds_limit_mark(reinterpret_cast<const void*>(kern));
  }
  p.nMT = (p.T + BT - 1) / BT;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT)), dim3(512), smem, st, p);
  PTPP_CHECK_LAUNCH("amp_layer_fwd (fused, 16-bit)");
  return PTPP_OK;
}

// the tallest tile whose LDS image leaves room for a second workgroup on the CU (2 x 80 KiB)
template <typename E, int C, int S, bool PAIR, int BT0, int BT1, int BT2, int BT3, int MGA0, int MGB0, int MGA1, int MGB1, int MGA2,
          int MGB2, int MGA3, int MGB3>
int launch_amp_fused_pick(AmpF& p, hipStream_t st, int variant) {
  constexpr size_t HALF = 80 * 1024;
  constexpr int WM = 8 / (C / 32);
  auto fits = [&](int BT, int MGA, size_t smem) {  // LDS for two workgroups per CU, conv1 row fragments per wave as built
    const int M1 = (BT + 2 * ((p.ks - 1) / 2) + 12 + 15) & ~15;
    return smem <= HALF && (M1 / 16 + WM - 1) / WM <= MGA;
  };
  if (variant == 1 || (variant == 0 && fits(BT0, MGA0, amp_fused_geometry<C, BT0, S, PAIR>(p)))) return launch_amp_fused<E, C, BT0, S, MGA0, MGB0, PAIR>(p, st);
  if (variant == 2 || (variant == 0 && fits(BT1, MGA1, amp_fused_geometry<C, BT1, S, PAIR>(p)))) return launch_amp_fused<E, C, BT1, S, MGA1, MGB1, PAIR>(p, st);
  if (variant == 3 || (variant == 0 && fits(BT2, MGA2, amp_fused_geometry<C, BT2, S, PAIR>(p)))) return launch_amp_fused<E, C, BT2, S, MGA2, MGB2, PAIR>(p, st);
  return launch_amp_fused<E, C, BT3, S, MGA3, MGB3, PAIR>(p, st);
}

}  // namespace

// called by ptpp_amp_layer_fwd (amp_layer.hip) for 16-bit tensors; variant: PTPP_AMP_VARIANT (experiments)
int amp_fused_launch_16bit(const ptpp_amp_layer_args* a, void* stream) {
  AmpF p;
  p.x = a->x; p.y = a->y; p.res2 = a->res2; p.w1p = a->w1s; p.w2p = a->w2s; p.b1 = a->b1; p.b2 = a->b2;
  p.la1 = a->log_alpha1; p.la2 = a->log_alpha2;
  for (int i = 0; i < 12; ++i) {
    p.up1[i] = a->up1[i]; p.dn1[i] = a->dn1[i]; p.up2[i] = a->up2[i]; p.dn2[i] = a->dn2[i];
  }
  p.B = a->B; p.T = a->T; p.ks = a->ks; p.dil = a->dil;
  p.out_scale = a->out_scale; p.res_scale = a->res_scale;
  p.nMT = 0; p.R1 = p.R2 = 0; p.rowsX = 0;
  const int skip = getenv("PTPP_AMP_SKIP") ? atoi(getenv("PTPP_AMP_SKIP")) : 0;
  const int variant = getenv("PTPP_AMP_VARIANT") ? atoi(getenv("PTPP_AMP_VARIANT")) : 0;
  p.skip = skip;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  p.stagger = getenv("PTPP_AMP_STAGGER") ? atoi(getenv("PTPP_AMP_STAGGER")) : 0;
  // variant 0: tallest tile that keeps two workgroups per CU; 1 / 2 / 3: force the first / second / third height
  if (a->dtype == PTPP_BF16) {
    if (variant >= 5) {  // (experiment) the Snake on channel PAIRS per thread: half the run length
      if (a->C == 64) return launch_amp_fused_pick<EB16, 64, 144, true, 256, 224, 192, 128, 5, 4, 4, 4, 4, 3, 3, 2>(p, st, variant - 5);
      if (a->C == 32) return launch_amp_fused_pick<EB16, 32, 80, true, 448, 384, 320, 256, 4, 4, 4, 3, 3, 3, 3, 2>(p, st, variant - 5);
    }
    if (a->C == 64) return launch_amp_fused_pick<EB16, 64, 144, false, 256, 224, 192, 128, 5, 4, 4, 4, 4, 3, 3, 2>(p, st, variant);
    if (a->C == 32) return launch_amp_fused_pick<EB16, 32, 80, false, 448, 384, 320, 256, 4, 4, 4, 3, 3, 3, 3, 2>(p, st, variant);
  }
  if (a->dtype == PTPP_F16) {
    if (a->C == 64) return launch_amp_fused_pick<EF16, 64, 144, false, 256, 224, 192, 128, 5, 4, 4, 4, 4, 3, 3, 2>(p, st, variant);
    if (a->C == 32) return launch_amp_fused_pick<EF16, 32, 80, false, 448, 384, 320, 256, 4, 4, 4, 3, 3, 3, 3, 2>(p, st, variant);
  }
  ptpp_set_error("amp_fused: C=%d dtype=%d not built", a->C, a->dtype);
  return PTPP_ENOTSUP;
}

extern "C" int ptpp_snake_conv1d_supported(int C, int dtype) { return (dtype == PTPP_BF16 || dtype == PTPP_F16) && (C == 128 || C == 256); }

extern "C" int ptpp_snake_conv1d_fwd(const ptpp_snake_conv_args* a, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->y && a->ws && a->bias && a->log_alpha, "snake_conv: null pointer");
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->ks >= 1 && (a->ks & 1) && a->ks <= 15 && a->dil >= 1 && a->dil <= 8,
                 "snake_conv: bad shape B=%d T=%d ks=%d dil=%d", a->B, a->T, a->ks, a->dil);
  PTPP_CHECK_ARG(a->x != a->y, "snake_conv: in-place not supported (neighbouring tiles read the halo)");
  PTPP_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->res | (uintptr_t)a->res2 | (uintptr_t)a->ws | (uintptr_t)a->bias) & 15) == 0,
                 "snake_conv: pointers must be 16-byte aligned");
  if (!ptpp_snake_conv1d_supported(a->C, a->dtype)) {
    ptpp_set_error("snake_conv: C=%d dtype=%d not built (C in {128, 256}, 16-bit)", a->C, a->dtype);
    return PTPP_ENOTSUP;
  }
  SnkP p;
  p.x = a->x; p.y = a->y; p.res = a->res; p.res2 = a->res2; p.ws = a->ws; p.bias = a->bias; p.la = a->log_alpha;
  for (int i = 0; i < 12; ++i) { p.up[i] = a->up[i]; p.dn[i] = a->dn[i]; }
  p.B = a->B; p.T = a->T; p.ks = a->ks; p.dil = a->dil;
  p.out_scale = a->out_scale; p.res_scale = a->res_scale;
  p.nMT = 0; p.R = 0; p.rowsA = 0;
  p.skip = getenv("PTPP_AMP_SKIP") ? atoi(getenv("PTPP_AMP_SKIP")) : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int variant = getenv("PTPP_AMP_VARIANT") ? atoi(getenv("PTPP_AMP_VARIANT")) : 0;
  if (variant >= 4) {  // (experiment) the Snake on channel pairs
    if (a->C == 128) return launch_snake_conv<EB16, 128, 128, 272, 4, true>(p, st);
    return launch_snake_conv<EB16, 256, 64, 528, 4, true>(p, st);
  }
  // the tallest tile that leaves room for a second workgroup on the CU (the halo of 2 pad rows and the 5 warm-up steps of a
  // Snake run weigh less on a taller tile: Snake rows per output row 1.55 -> 1.36 at C = 128, k = 11, d = 5)
  constexpr size_t HALF = 80 * 1024;
  if (a->dtype == PTPP_F16) {
    if (a->C == 128) {
      if (variant != 1 && snake_conv_geometry<128, 192, 272, false>(p) <= HALF) return launch_snake_conv<EF16, 128, 192, 272, 6, false>(p, st);
      return launch_snake_conv<EF16, 128, 128, 272, 4, false>(p, st);
    }
    if (variant != 1 && snake_conv_geometry<256, 96, 528, false>(p) <= HALF) return launch_snake_conv<EF16, 256, 96, 528, 6, false>(p, st);
    return launch_snake_conv<EF16, 256, 64, 528, 4, false>(p, st);
  }
  if (a->C == 128) {
    if (variant != 1 && snake_conv_geometry<128, 192, 272, false>(p) <= HALF) return launch_snake_conv<EB16, 128, 192, 272, 6, false>(p, st);
    return launch_snake_conv<EB16, 128, 128, 272, 4, false>(p, st);
  }
  if (variant != 1 && snake_conv_geometry<256, 96, 528, false>(p) <= HALF) return launch_snake_conv<EB16, 256, 96, 528, 6, false>(p, st);
  return launch_snake_conv<EB16, 256, 64, 528, 4, false>(p, st);
}

extern "C" int ptpp_snake_conv_post_supported(int C, int ks, int dtype) {
  return (dtype == PTPP_BF16 || dtype == PTPP_F16) && C == 32 && ks >= 1 && ks <= 15 && (ks & 1);
}

extern "C" int ptpp_snake_conv_post_tanh(const void* x, const float* log_alpha, const float* filt_up, const float* filt_down,
                                         const float* w, float bias, float* y, int B, int T, int C, int ks, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && log_alpha && filt_up && filt_down && w && y, "snake_conv_post_tanh: null pointer");
  PTPP_CHECK_ARG(B > 0 && T > 0, "snake_conv_post_tanh: bad shape B=%d T=%d", B, T);
  PTPP_CHECK_ARG((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "snake_conv_post_tanh: pointers must be 16-byte aligned");
  if (!ptpp_snake_conv_post_supported(C, ks, dtype)) {
    ptpp_set_error("snake_conv_post_tanh: C=%d ks=%d dtype=%d not built (C = 32, odd ks, 16-bit)", C, ks, dtype);
    return PTPP_ENOTSUP;
  }
  constexpr int BT = 512, S = 80, CC = 32, NRUN = 512 / CC;
  SnkPostP p;
  p.x = x; p.y = y; p.w = w; p.la = log_alpha; p.bias = bias;
  for (int i = 0; i < 12; ++i) { p.up[i] = filt_up[i]; p.dn[i] = filt_down[i]; }
  p.B = B; p.T = T; p.ks = ks;
  const int n_a = BT + 2 * (ks / 2);
  p.R = (n_a + NRUN - 1) / NRUN;
  p.rowsA = n_a > NRUN * p.R ? n_a : NRUN * p.R;
  p.nMT = (T + BT - 1) / BT;
  const size_t smem = (size_t)p.rowsA * S + (size_t)ks * CC * sizeof(float);
  if (dtype == PTPP_F16)
    hipLaunchKernelGGL((snake_post_kernel<EF16, CC, BT, S>), dim3((unsigned)((int64_t)B * p.nMT)), dim3(512), smem,
                       reinterpret_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL((snake_post_kernel<EB16, CC, BT, S>), dim3((unsigned)((int64_t)B * p.nMT)), dim3(512), smem,
                       reinterpret_cast<hipStream_t>(stream), p);
  PTPP_CHECK_LAUNCH("snake_conv_post_tanh");
  return PTPP_OK;
}

extern "C" int ptpp_amp_pack_wstream(const void* wp, void* out, int C, int ks, int dtype, void* stream) {
  PTPP_CHECK_ARG(wp && out && wp != out, "amp_pack_wstream: null / aliased pointer");
  PTPP_CHECK_ARG((dtype == PTPP_BF16 || dtype == PTPP_F16) && (C == 32 || C == 64 || C == 128 || C == 256) && ks >= 1 && ks <= 15,
                 "amp_pack_wstream: C=%d ks=%d dtype=%d not built", C, ks, dtype);
  PTPP_CHECK_ARG((((uintptr_t)wp | (uintptr_t)out) & 15) == 0, "amp_pack_wstream: pointers must be 16-byte aligned");
  const int n = ks * (C / 32) * (C / 16) * 64;
  hipLaunchKernelGGL(amp_wstream_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint16_t*>(wp), reinterpret_cast<uint4*>(out), C, ks);
  PTPP_CHECK_LAUNCH("amp_pack_wstream");
  return PTPP_OK;
}
