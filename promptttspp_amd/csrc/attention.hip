// Multi-head attention with Transformer-XL style relative positions
// (esp/transformer/attention.py:63-93,142-206,237-305) for SHORT sequences
// (phones per utterance ~70, <= 1024): latency/launch bound, not FLOP bound
// (8 kFLOP x T per phone), so the design goal is ONE launch per layer for all
// (b, h, row) with the rel-shift folded into an index computation and the
// softmax reductions done with wavefront shuffles.
//
//   score[i,j] = ((q_i + u) . k_j + bd[i,j]) / sqrt(dk)
//   new    : bd[i,j] = (q_i + v) . p[T-1-i+j]                 p: (2T-1, C)
//   legacy : flat = (i+1) T + j, (r,c) = divmod(flat, T+1);    p: (T, C)
//            bd[i,j] = c == 0 ? 0 : (q_r + v) . p[c-1]         (r in {i, i+1})
//   plain  : bd = 0 (u = v = 0): standard scaled dot-product attention (BERT)
//   masked keys (j >= len_b) get probability 0; masked query rows output 0.
//
// One wave owns one query row: lanes run over keys j for the scores (q+u, q+v in
// LDS, broadcast reads; k/p rows streamed with 8/16-byte loads out of L2), a
// shuffle max/sum softmax, then lanes run over 4-channel vectors for P.V.
#include <stdlib.h>

#include "ptpp_common.h"

namespace {

enum { VAR_NEW = 0, VAR_LEGACY = 1, VAR_PLAIN = 2, VAR_WINDOW = 3 };
// VAR_WINDOW (modules/transformer.py:59-137, the FFT-block encoder plug-in): scores += q_i . emb_k[j - i + w] and
// ctx_i += sum_r P[i, i + r - w] emb_v[r] for |j - i| <= w; emb_k / emb_v: (2w + 1, dk) f32, shared by the heads; no u / v biases.
__device__ __forceinline__ bool has_uv(int variant) { return variant == VAR_NEW || variant == VAR_LEGACY; }
__device__ __forceinline__ float dot_f32(const float* __restrict__ a, const float* __restrict__ row, int dk) {
  float s = 0.f;
  for (int d = 0; d < dk; d += 4) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(row + d), av = *reinterpret_cast<const f32x4*>(a + d);
    s += av[0] * r[0] + av[1] * r[1] + av[2] * r[2] + av[3] * r[3];
  }
  return s;
}

struct AttnP {
  const void *q, *k, *v, *pos;
  const float *bias_u, *bias_v;
  void* ctx;
  float* probs;
  const int* lengths;
  int B, T, H, dk, ld, ldpos, ldctx, variant;
  float scale;
  unsigned drop_thresh16;  // attention-probability dropout, 0 = off
  float drop_inv_keep;
  unsigned long long drop_seed;
  const float *emb_k, *emb_v;  // VAR_WINDOW
  int window;
};

template <typename T>
__device__ __forceinline__ float dot_row(const float* __restrict__ a, const T* __restrict__ row, int dk) {
  float s = 0.f;
  for (int d = 0; d < dk; d += 4) {
    const f32x4 r = Elem<T>::ld4(row + d);
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + d);
    s += av[0] * r[0] + av[1] * r[1] + av[2] * r[2] + av[3] * r[3];
  }
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  const int Tn = p.T, dk = p.dk;
  const int Tpad = (Tn + 3) & ~3;  // keep every per-wave LDS region 16-byte aligned
  float* qu = lds + w * (3 * dk + Tpad);
  float* qv = qu + dk;
  float* qv2 = qv + dk;
  float* sc = qv2 + dk;
  if (i >= Tn) return;
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * dk;
  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + hc;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + hc;
  const T* pb = p.pos ? reinterpret_cast<const T*>(p.pos) + hc : nullptr;
  T* ob = reinterpret_cast<T*>(p.ctx) + ((int64_t)b * Tn + i) * p.ldctx + hc;
  float* prow = p.probs ? p.probs + (((int64_t)b * p.H + h) * Tn + i) * Tn : nullptr;

  if (i >= len) {  // padded query: probabilities are all zeroed by the mask
    for (int d = lane * 4; d < dk; d += 256) Elem<T>::st4(ob + d, f32x4{0.f, 0.f, 0.f, 0.f});
    if (prow) for (int j = lane; j < Tn; j += 64) prow[j] = 0.f;
    return;
  }
  for (int d = lane * 4; d < dk; d += 256) {
    const f32x4 qi = Elem<T>::ld4(qb + (int64_t)i * p.ld + d);
    f32x4 bu = f32x4{0.f, 0.f, 0.f, 0.f}, bv = bu;
    if (has_uv(p.variant)) {
      bu = *reinterpret_cast<const f32x4*>(p.bias_u + hc + d);
      bv = *reinterpret_cast<const f32x4*>(p.bias_v + hc + d);
    }
    *reinterpret_cast<f32x4*>(qu + d) = qi + bu;
    *reinterpret_cast<f32x4*>(qv + d) = qi + bv;
    if (p.variant == VAR_LEGACY) {
      f32x4 qn = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i + 1 < Tn) qn = Elem<T>::ld4(qb + (int64_t)(i + 1) * p.ld + d) + bv;
      *reinterpret_cast<f32x4*>(qv2 + d) = qn;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes are visible

  float mx = -3.0e38f;
  for (int j = lane; j < len; j += 64) {
    float s = dot_row<T>(qu, kb + (int64_t)j * p.ld, dk);
    if (p.variant == VAR_NEW) {
      s += dot_row<T>(qv, pb + (int64_t)(Tn - 1 - i + j) * p.ldpos, dk);
    } else if (p.variant == VAR_LEGACY) {
      const int flat = (i + 1) * Tn + j;
      const int r = flat / (Tn + 1), c = flat - r * (Tn + 1);
      if (c != 0) s += dot_row<T>(r == i ? qv : qv2, pb + (int64_t)(c - 1) * p.ldpos, dk);
    } else if (p.variant == VAR_WINDOW) {
      const int r = j - i + p.window;
      if (r >= 0 && r <= 2 * p.window) s += dot_f32(qu, p.emb_k + (int64_t)r * dk, dk);
    }
    s *= p.scale;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const uint64_t drow = (((uint64_t)b * p.H + h) * Tn + i) * (uint64_t)Tn;  // element index of P[b,h,i,0]
  for (int j = lane; j < Tn; j += 64) {
    float pr = j < len ? sc[j] * inv : 0.f;
    if (prow) prow[j] = pr;  // the softmax itself: the backward regenerates the dropout mask from (seed, index)
    if (p.drop_thresh16 && j < len) {
      const uint64_t e = drow + j;
      const uint32_t bits = (uint32_t)(drop_hash(p.drop_seed, e >> 2) >> (16 * (e & 3))) & 0xffffu;
      pr = bits >= p.drop_thresh16 ? pr * p.drop_inv_keep : 0.f;
    }
    if (j < len) sc[j] = pr;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);

  // ctx[d] = sum_j P[j] V[j][d]: lanes = (key part) x (4-channel vector)
  const int nvec = dk >> 2;            // vectors per head (<= 64)
  const int parts = 64 / nvec;         // power of two for dk in {64, 128, 256}
  const int dv = (lane % nvec) * 4, part = lane / nvec;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (part < parts)
    for (int j = part; j < len; j += parts) acc += Elem<T>::ld4(vb + (int64_t)j * p.ld + dv) * sc[j];
  if (p.variant == VAR_WINDOW && part == 0)  // + sum_r P[i, i + r - w] emb_v[r]
    for (int r = 0; r <= 2 * p.window; ++r) {
      const int j = i + r - p.window;
      if (j >= 0 && j < len) acc += *reinterpret_cast<const f32x4*>(p.emb_v + (int64_t)r * dk + dv) * sc[j];
    }
  for (int o = nvec; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane < nvec) Elem<T>::st4(ob + dv, acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward on the matrix cores with K and V of one (utterance, head) resident in LDS (round 3; bf16, T <= 256,
// dk in {64, 128}, the "new" rel-pos table or no table).  One block = 64 query rows of one (b, h), one wave = 16 of them:
//   K, V (T x dk)                -> LDS once per block (swizzled rows; V in the layout of the transposing LDS read)
//   S^T  = K_frag  x (q + u)^T   v_mfma_f32_16x16x32_bf16: A = 16 keys x 32 channels from LDS, B = the wave's 16 query rows
//                                from registers; a lane ends up with ONE query row (lane & 15) and 4 consecutive keys per
//                                16-key fragment, so the softmax reductions are register loops + two shuffles
//   bd   : R^T = P_frag x (q + v)^T over the 32 table rows a 16 x 16 (query, key) block can touch (A straight from
//                                global memory: the table is shared by every block and lives in L2), then the skew
//                                bd[i, j] = R[i, T-1-i+j] through a 16 x 32 f32 LDS patch per wave
//   O^T  = V^T_frag x P          A = 16 channels x 32 keys by ds_read_b64_tr_b16 (the transposing read of the weight-
//                                gradient kernel), B = the lane's own probabilities: the K order inside an MFMA step is
//                                free, so it is chosen to be the order the lane already holds them in
// Same masking, probability output (f32, before dropout) and dropout mask as attn_fwd_kernel; q + u / q + v are rounded
// to bf16 operands (the row kernel keeps them in f32), which is within the bf16 tolerance of the tests.
constexpr int MF_QT = 64;  // query rows per block

template <int CPR>
__device__ __forceinline__ int vsw(int row);  // chunk swizzle of the V image (as conv1d_wgrad_bf16.hip::sw)
template <>
__device__ __forceinline__ int vsw<16>(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }
template <>
__device__ __forceinline__ int vsw<8>(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }
template <int CPR>
__device__ __forceinline__ int ksw(int row);  // chunk swizzle of the K image: the 16 lanes of a b128 group read 16 rows
template <>
__device__ __forceinline__ int ksw<16>(int row) { return row & 15; }
template <>
__device__ __forceinline__ int ksw<8>(int row) { return (row >> 1) & 7; }

typedef __attribute__((ext_vector_type(4))) short mf_v4s;
__device__ __forceinline__ mf_v4s mf_tr_read(const bf16_raw* p) {
  mf_v4s r;
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) bf16_raw*)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  return r;
}
__device__ __forceinline__ uint4 mf_pack8(const float (&v)[8]) {
  uint4 o;
  o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  return o;
}

template <int DK>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const AttnP p) {
  typedef bf16_raw T;
  constexpr int CPR = DK / 8;      // 16-byte chunks per K / V row
  constexpr int KS = DK / 32;      // MFMA K steps over the channels
  constexpr int MAXNF = 16;        // key fragments of 16 (T <= 256)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tn = p.T;
  const int Tp = (Tn + 31) & ~31;  // keys padded to whole 32-key MFMA steps (zero rows)
  uint4* Ks = reinterpret_cast<uint4*>(smem);                 // [Tp][CPR] swizzled chunks
  T* Vs = reinterpret_cast<T*>(Ks + (size_t)Tp * CPR);        // [Tp][DK], chunk-swizzled for the transposing read
  float* Rs = reinterpret_cast<float*>(Vs + (size_t)Tp * DK); // [4 waves][16][33] skew patches
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int i0 = blockIdx.x * MF_QT + w * 16;  // the wave's first query row
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * DK;
  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + hc;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + hc;
  const T* pb = p.pos ? reinterpret_cast<const T*>(p.pos) + hc : nullptr;

  // ---- K, V -> LDS (rows >= len are never used with a non-zero weight, rows >= T are zero) ----
  for (int idx = tid; idx < Tp * CPR; idx += 256) {
    const int row = idx / CPR, c = idx % CPR;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
    if (row < Tn) {
      kv = *reinterpret_cast<const uint4*>(kb + (int64_t)row * p.ld + c * 8);
      vv = *reinterpret_cast<const uint4*>(vb + (int64_t)row * p.ld + c * 8);
    }
    Ks[row * CPR + (c ^ ksw<CPR>(row))] = kv;
    *reinterpret_cast<uint4*>(Vs + row * DK + ((c ^ vsw<CPR>(row)) << 3)) = vv;
  }
  // ---- the wave's query operands: lane (query lq, channel chunk lg of each K step) ----
  const int iq = i0 + lq;
  const bool qvalid = iq < Tn;
  uint4 qu[KS], qv[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qvalid) {
      const int ch = ks * 32 + lg * 8;
      const uint4 r = *reinterpret_cast<const uint4*>(qb + (int64_t)iq * p.ld + ch);
      const float qf[8] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u), __uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
                           __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] = qf[e] + (p.variant != VAR_PLAIN ? p.bias_u[hc + ch + e] : 0.f);
        c[e] = qf[e] + (p.variant != VAR_PLAIN ? p.bias_v[hc + ch + e] : 0.f);
      }
    }
    qu[ks] = mf_pack8(a);
    qv[ks] = mf_pack8(c);
  }
  __syncthreads();
  if (i0 >= Tn) return;  // (after the only block-wide barrier)

  // ---- scores: s[nf][r] = score of query iq against key 16 nf + 4 lg + r ----
  const int nfr = (len + 15) >> 4;  // key fragments that hold unmasked keys
  f32x4 sc[MAXNF];
  float* Rw = Rs + w * (16 * 33);
#pragma unroll
  for (int nf = 0; nf < MAXNF; ++nf) {
    sc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nf < nfr) {
      const int krow = nf * 16 + lq;  // A operand: lane (key krow, channel chunk lg)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = Ks[krow * CPR + ((ks * 4 + lg) ^ ksw<CPR>(krow))];
        sc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qu[ks]), sc[nf], 0, 0, 0);
      }
      if (p.variant == VAR_NEW) {
        // table rows this (16 queries x 16 keys) block touches: m = T-1-i+j, i in [i0, i0+15], j in [16 nf, 16 nf + 15]
        const int m0 = Tn - 1 - (i0 + 15) + nf * 16;  // >= -15 + ... ; rows outside [0, 2T-2] belong to masked / absent pairs
        f32x4 r0 = f32x4{0.f, 0.f, 0.f, 0.f}, r1 = r0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int ma = m0 + lq, mb = m0 + 16 + lq;
          uint4 pa = make_uint4(0, 0, 0, 0), pbv = pa;
          if (ma >= 0 && ma < 2 * Tn - 1) pa = *reinterpret_cast<const uint4*>(pb + (int64_t)ma * p.ldpos + ks * 32 + lg * 8);
          if (mb >= 0 && mb < 2 * Tn - 1) pbv = *reinterpret_cast<const uint4*>(pb + (int64_t)mb * p.ldpos + ks * 32 + lg * 8);
          r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa), __builtin_bit_cast(bf16x8_t, qv[ks]), r0, 0, 0, 0);
          r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pbv), __builtin_bit_cast(bf16x8_t, qv[ks]), r1, 0, 0, 0);
        }
        // skew through the wave's LDS patch: R[query lq][c], c = table row - m0 in [0, 32); wanted c = 15 - lq + (4 lg + r)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Rw[lq * 33 + 4 * lg + r] = r0[r];
          Rw[lq * 33 + 16 + 4 * lg + r] = r1[r];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[nf][r] += Rw[lq * 33 + 15 - lq + 4 * lg + r];
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // ---- softmax over the keys of the lane's query row (registers + the 4 lane groups) ----
  float mx = -3.0e38f;
#pragma unroll
  for (int nf = 0; nf < MAXNF; ++nf)
    if (nf < nfr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = nf * 16 + 4 * lg + r;
        sc[nf][r] = j < len ? sc[nf][r] * p.scale : -3.0e38f;
        mx = fmaxf(mx, sc[nf][r]);
      }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int nf = 0; nf < MAXNF; ++nf)
    if (nf < nfr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = nf * 16 + 4 * lg + r;
        const float e = j < len ? __expf(sc[nf][r] - mx) : 0.f;
        sc[nf][r] = e;
        sum += e;
      }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const bool row_on = qvalid && iq < len;  // padded query rows: all probabilities zero, output zero
  const float inv = row_on ? 1.f / sum : 0.f;
  float* prow = (p.probs && qvalid) ? p.probs + (((int64_t)b * p.H + h) * Tn + iq) * Tn : nullptr;
  const uint64_t drow = (((uint64_t)b * p.H + h) * Tn + (uint64_t)iq) * (uint64_t)Tn;
#pragma unroll
  for (int nf = 0; nf < MAXNF; ++nf) {
    if (nf * 16 < Tn) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = nf * 16 + 4 * lg + r;
        float pr = nf < nfr ? sc[nf][r] * inv : 0.f;
        if (prow && j < Tn) prow[j] = pr;  // the softmax itself (the backward regenerates the dropout mask)
        if (p.drop_thresh16 && j < len) {
          const uint64_t e = drow + j;
          const uint32_t bits = (uint32_t)(drop_hash(p.drop_seed, e >> 2) >> (16 * (e & 3))) & 0xffffu;
          pr = bits >= p.drop_thresh16 ? pr * p.drop_inv_keep : 0.f;
        }
        if (nf < nfr) sc[nf][r] = pr;
      }
    }
  }
  // ---- context: O^T[channel][query] = sum_j V[j][channel] P[query][j]; the 32 keys of an MFMA step are taken in the order
  // the lane holds them: slots 0-3 = keys kb0 + 4 lg + (0..3), slots 4-7 = keys kb0 + 16 + 4 lg + (0..3) ----
  f32x4 oc[DK / 16];
#pragma unroll
  for (int mf = 0; mf < DK / 16; ++mf) oc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (nfr + 1) >> 1;
#pragma unroll
  for (int st = 0; st < MAXNF / 2; ++st) {
    if (st < nsteps) {
      float pv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = sc[2 * st][r];
        pv[4 + r] = (2 * st + 1 < nfr) ? sc[2 * st + 1][r] : 0.f;
      }
      const uint4 pf = mf_pack8(pv);
      const int kb0 = st * 32;
#pragma unroll
      for (int mf = 0; mf < DK / 16; ++mf) {
        // transposing reads: lane i of a 16-lane group addresses row (base + (i >> 2)), columns mf*16 + 4 (i & 3) .. +3 and
        // receives column mf*16 + i of the 4 rows base .. base + 3
        const int col = mf * 16 + 4 * (lq & 3);
        const int ra = kb0 + 4 * lg + (lq >> 2), rb = ra + 16;
        const mf_v4s lo = mf_tr_read(Vs + ra * DK + (((col >> 3) ^ vsw<CPR>(ra)) << 3) + (col & 7));
        const mf_v4s hi = mf_tr_read(Vs + rb * DK + (((col >> 3) ^ vsw<CPR>(rb)) << 3) + (col & 7));
        typedef __attribute__((ext_vector_type(8))) short v8s_;
        v8s_ av;
        av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3];
        av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
        oc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, pf), oc[mf], 0, 0, 0);
      }
    }
  }
  if (qvalid) {
    T* ob = reinterpret_cast<T*>(p.ctx) + ((int64_t)b * Tn + iq) * p.ldctx + hc;
#pragma unroll
    for (int mf = 0; mf < DK / 16; ++mf) Elem<T>::st4(ob + mf * 16 + 4 * lg, row_on ? oc[mf] : f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

// ---- backward, kernel 1: per query row -> dS row, dq, du, dv ------------------
struct AttnBwdP {
  const void *q, *k, *v, *pos, *dctx;
  const float *bias_u, *bias_v, *probs;
  float* dS;       // (B,H,T,T)
  void* dq;        // (B,T,*) with stride lddq
  float *du, *dvb; // (H*dk) accumulators (filled by red_sum_kernel from the scratch)
  float* scratch;  // reduction scratch: 32 replicas of [du | dvb]
  const int* lengths;
  int B, T, H, dk, ld, ldpos, lddctx, lddq, variant;
  float scale;
  unsigned drop_thresh16;  // the forward's probability dropout (0 = off): the mask is regenerated here
  float drop_inv_keep;
  unsigned long long drop_seed;
  const float *emb_k, *emb_v;  // VAR_WINDOW
  int window;
};

// keep-scale of element (b, h, i, j) of the attention probabilities: 1/(1-p) if kept, 0 if dropped (as the forward)
__device__ __forceinline__ float attn_keep(const AttnBwdP& p, uint64_t e) {
  const uint32_t bits = (uint32_t)(drop_hash(p.drop_seed, e >> 2) >> (16 * (e & 3))) & 0xffffu;
  return bits >= p.drop_thresh16 ? p.drop_inv_keep : 0.f;
}

// Query rows per wave in attn_bwd_row_kernel.  One: the row loops are chains of dependent L2 reads, so
// the kernel wants many resident waves (4 rows per wave left 1.75 waves per SIMD on the phone-level
// shapes); the du / dvb sums, which once forced several rows per wave to cut same-address atomics, now go
// through the replicated reduction scratch (ptpp_common.h).
constexpr int ROWS_PER_WAVE = 1;

// dS row i of (b, h) into LDS `ds` (and to global memory when `store`): dS = P * (dP - sum_j P dP) * scale with
// dP[j] = dctx_i . v_j (times the regenerated dropout keep-scale); `go` receives dctx_i.  Rows i >= len are all zero.
template <typename T>
__device__ __forceinline__ void attn_ds_row(const AttnBwdP& p, int b, int h, int i, int len, float* go, float* ds, bool store,
                                            int lane) {
  const int Tn = p.T, dk = p.dk, hc = h * dk;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + hc;
  const T* gb = reinterpret_cast<const T*>(p.dctx) + ((int64_t)b * Tn + i) * p.lddctx + hc;
  const float* prow = p.probs + (((int64_t)b * p.H + h) * Tn + i) * Tn;
  float* dsrow = p.dS + (((int64_t)b * p.H + h) * Tn + i) * Tn;
  __builtin_amdgcn_wave_barrier();  // the previous row's LDS reads are done
  for (int d = lane * 4; d < dk; d += 256) *reinterpret_cast<f32x4*>(go + d) = Elem<T>::ld4(gb + d);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float dsum = 0.f;
  const uint64_t drow = (((uint64_t)b * p.H + h) * Tn + i) * (uint64_t)Tn;
  for (int j = lane; j < len; j += 64) {
    float dp = dot_row<T>(go, vb + (int64_t)j * p.ld, dk);  // d(dropped P)
    if (p.variant == VAR_WINDOW) {
      const int r = j - i + p.window;
      if (r >= 0 && r <= 2 * p.window) dp += dot_f32(go, p.emb_v + (int64_t)r * dk, dk);
    }
    if (p.drop_thresh16) dp *= attn_keep(p, drow + j);        // dP = mask/(1-p) * d(dropped P)
    ds[j] = dp;
    dsum += prow[j] * dp;
  }
  dsum = wave_sum(dsum);
  for (int j = lane; j < Tn; j += 64) {
    const float v = j < len ? prow[j] * (ds[j] - dsum) * p.scale : 0.f;
    ds[j] = v;
    if (store) dsrow[j] = v;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_row_kernel(const AttnBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int Tn = p.T, dk = p.dk;
  const int Tpad = (Tn + 3) & ~3;
  const bool legacy = p.variant == VAR_LEGACY;
  const int per = dk + Tpad + (legacy ? Tpad : 0);
  float* go = lds + w * per;  // dctx_i
  float* ds = go + dk;
  float* ds2 = ds + Tpad;     // legacy: dS row i - 1 (see below)
  float* red = lds + 4 * per;  // [4 waves][2][dk] partial du / dvb of the block
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * dk;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + hc;
  const T* pb = p.pos ? reinterpret_cast<const T*>(p.pos) + hc : nullptr;
  const int nvec = dk >> 2, parts = 64 / nvec;
  const int dv = (lane % nvec) * 4, part = lane / nvec;
  // du / dvb: summed over the wave's rows in registers, over the block's waves in LDS, then ONE
  // atomic per channel and block (per-row atomics onto the same H*dk addresses serialised the kernel)
  f32x4 su = f32x4{0.f, 0.f, 0.f, 0.f}, sv = su;
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int i = (blockIdx.x * 4 + w) * ROWS_PER_WAVE + rr;
    if (i >= Tn) break;
    T* dqr = reinterpret_cast<T*>(p.dq) + ((int64_t)b * Tn + i) * p.lddq + hc;
    float* dsrow = p.dS + (((int64_t)b * p.H + h) * Tn + i) * Tn;
    // legacy: the pad/view shift of the reference (attention.py:142-162) reads row i's positional scores for the keys
    // j <= i from (q_i + v) . pos[T-1-i+j], and for the keys j >= i+2 from (q_{i+1} + v) . pos[j-i-2] -- so dq_i also
    // receives dS[i-1, j] pos[j-i-1] for j >= i+1.  Row i-1 of dS belongs to another wave: it is recomputed here
    // (the legacy table is the demo configuration: simplicity over speed).
    const bool prev = legacy && i >= 1 && i - 1 < len;
    if (prev) attn_ds_row<T>(p, b, h, i - 1, len, go, ds2, false, lane);
    if (i >= len) {
      for (int j = lane; j < Tn; j += 64) dsrow[j] = 0.f;
      for (int j = lane; j < Tn; j += 64) ds[j] = 0.f;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (!prev) {
        for (int d = lane * 4; d < dk; d += 256) Elem<T>::st4(dqr + d, f32x4{0.f, 0.f, 0.f, 0.f});
        continue;
      }
    } else {
      attn_ds_row<T>(p, b, h, i, len, go, ds, true, lane);
    }
    f32x4 au = f32x4{0.f, 0.f, 0.f, 0.f}, av = au;
    if (part < parts) {
      if (i < len)
        for (int j = part; j < len; j += parts) {
          au += Elem<T>::ld4(kb + (int64_t)j * p.ld + dv) * ds[j];
          if (p.variant == VAR_NEW) av += Elem<T>::ld4(pb + (int64_t)(Tn - 1 - i + j) * p.ldpos + dv) * ds[j];
        }
      if (p.variant == VAR_WINDOW && i < len && part == 0)
        for (int r = 0; r <= 2 * p.window; ++r) {
          const int j = i + r - p.window;
          if (j >= 0 && j < len) av += *reinterpret_cast<const f32x4*>(p.emb_k + (int64_t)r * dk + dv) * ds[j];
        }
      if (legacy) {
        if (i < len)
          for (int j = part; j <= i && j < len; j += parts) av += Elem<T>::ld4(pb + (int64_t)(Tn - 1 - i + j) * p.ldpos + dv) * ds[j];
        if (prev)
          for (int j = i + 1 + part; j < len; j += parts) av += Elem<T>::ld4(pb + (int64_t)(j - i - 1) * p.ldpos + dv) * ds2[j];
      }
    }
    for (int o = nvec; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        au[e] += __shfl_xor(au[e], o, 64);
        av[e] += __shfl_xor(av[e], o, 64);
      }
    }
    if (lane < nvec) Elem<T>::st4(dqr + dv, au + av);
    su += au;
    sv += av;
  }
  if (has_uv(p.variant)) {
    if (lane < nvec) {
      *reinterpret_cast<f32x4*>(red + (w * 2 + 0) * dk + dv) = su;
      *reinterpret_cast<f32x4*>(red + (w * 2 + 1) * dk + dv) = sv;
    }
    __syncthreads();
    float* rep = p.scratch + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % PTPP_RED_NREP) * (2 * p.H * dk);
    for (int c = threadIdx.x; c < 2 * dk; c += 256) {
      const float v = red[c] + red[2 * dk + c] + red[4 * dk + c] + red[6 * dk + c];
      atomicAdd(rep + (c < dk ? hc + c : p.H * dk + hc + (c - dk)), v);  // [du (H*dk) | dvb (H*dk)]
    }
  }
}

// VAR_WINDOW: gradients of the two relative-position tables, summed over utterances, heads and query rows:
//   demb_k[r] = sum dS[b,h,i,i+r-w] q[b,i,h]      demb_v[r] = sum P_dropped[b,h,i,i+r-w] dctx[b,i,h]
// grid (2w + 1, H, batch groups); the 4 waves of a block split the rows, lanes = (row part) x (4-channel vector)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_win_emb_kernel(const AttnBwdP p, float* __restrict__ demb_k, float* __restrict__ demb_v) {
  __shared__ f32x4 red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x, h = blockIdx.y;
  const int nbg = gridDim.z, bper = (p.B + nbg - 1) / nbg;
  const int b0 = blockIdx.z * bper, b1 = min(p.B, b0 + bper);
  const int Tn = p.T, dk = p.dk, hc = h * dk;
  const int nvec = dk >> 2, parts = 64 / nvec;
  const int dv = (lane % nvec) * 4, part = lane / nvec;
  f32x4 ak = f32x4{0.f, 0.f, 0.f, 0.f}, av = ak;
  if (part < parts)
    for (int b = b0; b < b1; ++b) {
      const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
      const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
      const T* gb = reinterpret_cast<const T*>(p.dctx) + (int64_t)b * Tn * p.lddctx + hc;
      const int64_t base = ((int64_t)b * p.H + h) * Tn * Tn;
      for (int i = w * parts + part; i < len; i += 4 * parts) {
        const int j = i + r - p.window;
        if (j < 0 || j >= len) continue;
        const int64_t e = base + (int64_t)i * Tn + j;
        float pv = p.probs[e];
        if (p.drop_thresh16) pv *= attn_keep(p, (uint64_t)e);
        ak += Elem<T>::ld4(qb + (int64_t)i * p.ld + dv) * p.dS[e];
        av += Elem<T>::ld4(gb + (int64_t)i * p.lddctx + dv) * pv;
      }
    }
  for (int o = nvec; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ak[e] += __shfl_xor(ak[e], o, 64);
      av[e] += __shfl_xor(av[e], o, 64);
    }
  }
  if (lane < nvec) { red[0][w][lane] = ak; red[1][w][lane] = av; }
  __syncthreads();
  if (w == 0 && lane < nvec) {
    const f32x4 sk = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    const f32x4 sv = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(demb_k + (int64_t)r * dk + dv + e, sk[e]);
      atomicAdd(demb_v + (int64_t)r * dk + dv + e, sv[e]);
    }
  }
}

// The same for the tables without the legacy shift, FOUR consecutive query rows per wave that share every K / V / pos row
// they read: with one row per wave each of the B H T rows streams all of K, V and the positional table of its (b, h) through
// the caches (150 KB per row, 1.2 GB per launch at phone level: the kernel ran at the L2's pace, 125 us).  Row r of the wave
// reads pos[T-1-(i0+r)+j] for key j, i.e. the rows of a wave walk the SAME table rows one key apart: the loop runs over the
// table index m and hands pos[m] to every row whose key m - (T-1-i0-r) exists.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_row4_kernel(const AttnBwdP p) {
  constexpr int R = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int Tn = p.T, dk = p.dk;
  const int Tpad = (Tn + 3) & ~3;
  const int per = R * (dk + Tpad);
  float* go = lds + w * per;       // [R][dk]   dctx rows
  float* ds = go + R * dk;         // [R][Tpad] dS rows
  float* red = lds + 4 * per;      // [4 waves][2][dk] partial du / dvb of the block
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * dk;
  const int i0 = (blockIdx.x * 4 + w) * R;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + hc;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + hc;
  const T* pb = p.pos ? reinterpret_cast<const T*>(p.pos) + hc : nullptr;
  const int nvec = dk >> 2, parts = 64 / nvec;
  const int dv = (lane % nvec) * 4, part = lane / nvec;
  f32x4 su = f32x4{0.f, 0.f, 0.f, 0.f}, sv = su;
  if (i0 < Tn) {
    // rows i0 + r < len are live; rows in [len, T) only get their zeros
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r;
      const bool live = i < len;
      const T* gb = reinterpret_cast<const T*>(p.dctx) + ((int64_t)b * Tn + min(i, Tn - 1)) * p.lddctx + hc;
      for (int d = lane * 4; d < dk; d += 256)
        *reinterpret_cast<f32x4*>(go + r * dk + d) = live ? Elem<T>::ld4(gb + d) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int64_t prow0 = (((int64_t)b * p.H + h) * Tn + i0) * Tn;  // P / dS row i0 of (b, h); row r follows at + r T
    float dsum[R] = {0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < (i0 < len ? len : 0); j += 64) {
      const T* vrow = vb + (int64_t)j * p.ld;
      float acc[R] = {0.f, 0.f, 0.f, 0.f};
      for (int d = 0; d < dk; d += 4) {
        const f32x4 vv = Elem<T>::ld4(vrow + d);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(go + r * dk + d);
          acc[r] += g[0] * vv[0] + g[1] * vv[1] + g[2] * vv[2] + g[3] * vv[3];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float dp = acc[r];
        if (i0 + r < len) {
          if (p.drop_thresh16) dp *= attn_keep(p, (uint64_t)(prow0 + (int64_t)r * Tn) + j);
          dsum[r] += p.probs[prow0 + (int64_t)r * Tn + j] * dp;
        }
        ds[r * Tpad + j] = dp;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) dsum[r] = wave_sum(dsum[r]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r;
      if (i >= Tn) break;
      const bool live = i < len;
      float* dsrow = p.dS + prow0 + (int64_t)r * Tn;
      for (int j = lane; j < Tn; j += 64) {
        const float v = (live && j < len) ? p.probs[prow0 + (int64_t)r * Tn + j] * (ds[r * Tpad + j] - dsum[r]) * p.scale : 0.f;
        ds[r * Tpad + j] = v;
        dsrow[j] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    f32x4 au[R], av[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { au[r] = f32x4{0.f, 0.f, 0.f, 0.f}; av[r] = au[r]; }
    if (part < parts && i0 < len) {
      for (int j = part; j < len; j += parts) {
        const f32x4 kv = Elem<T>::ld4(kb + (int64_t)j * p.ld + dv);
#pragma unroll
        for (int r = 0; r < R; ++r) au[r] += kv * ds[r * Tpad + j];
      }
      if (p.variant == VAR_NEW) {
        const int mlo = Tn - 1 - i0 - (R - 1), mhi = Tn - 1 - i0 + len - 1;  // table rows any of the R rows reads
        for (int m = max(mlo, 0) + part; m <= mhi; m += parts) {
          const f32x4 pv = Elem<T>::ld4(pb + (int64_t)m * p.ldpos + dv);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int j = m - (Tn - 1 - i0 - r);
            if (j >= 0 && j < len) av[r] += pv * ds[r * Tpad + j];
          }
        }
      }
    }
    for (int o = nvec; o < 64; o <<= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          au[r][e] += __shfl_xor(au[r][e], o, 64);
          av[r][e] += __shfl_xor(av[r][e], o, 64);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r;
      if (i >= Tn) break;
      T* dqr = reinterpret_cast<T*>(p.dq) + ((int64_t)b * Tn + i) * p.lddq + hc;
      if (lane < nvec) Elem<T>::st4(dqr + dv, au[r] + av[r]);
      su += au[r];
      sv += av[r];
    }
  }
  if (p.variant != VAR_PLAIN) {
    if (lane < nvec) {
      *reinterpret_cast<f32x4*>(red + (w * 2 + 0) * dk + dv) = su;
      *reinterpret_cast<f32x4*>(red + (w * 2 + 1) * dk + dv) = sv;
    }
    __syncthreads();
    float* rep = p.scratch + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % PTPP_RED_NREP) * (2 * p.H * dk);
    for (int c = threadIdx.x; c < 2 * dk; c += 256) {
      const float v = red[c] + red[2 * dk + c] + red[4 * dk + c] + red[6 * dk + c];
      atomicAdd(rep + (c < dk ? hc + c : p.H * dk + hc + (c - dk)), v);  // [du (H*dk) | dvb (H*dk)]
    }
  }
}

// ---- backward, kernel 2: per key row -> dK, dV --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_col_kernel(const AttnBwdP p, void* dk_out, void* dv_out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + w, h = blockIdx.y, b = blockIdx.z;
  const int Tn = p.T, dk = p.dk;
  if (j >= Tn) return;
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * dk;
  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
  const T* gb = reinterpret_cast<const T*>(p.dctx) + (int64_t)b * Tn * p.lddctx + hc;
  const float* pcol = p.probs + ((int64_t)b * p.H + h) * Tn * Tn + j;
  const float* dcol = p.dS + ((int64_t)b * p.H + h) * Tn * Tn + j;
  T* dkr = reinterpret_cast<T*>(dk_out) + ((int64_t)b * Tn + j) * p.lddq + hc;
  T* dvr = reinterpret_cast<T*>(dv_out) + ((int64_t)b * Tn + j) * p.lddq + hc;
  const int nvec = dk >> 2, parts = 64 / nvec;
  const int dv = (lane % nvec) * 4, part = lane / nvec;
  f32x4 ak = f32x4{0.f, 0.f, 0.f, 0.f}, av = ak;
  if (j < len && part < parts) {
    f32x4 bu = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_uv(p.variant)) bu = *reinterpret_cast<const f32x4*>(p.bias_u + hc + dv);
    for (int i = part; i < len; i += parts) {
      const float dsv = dcol[(int64_t)i * Tn];
      float pv = pcol[(int64_t)i * Tn];
      if (p.drop_thresh16) pv *= attn_keep(p, (((uint64_t)b * p.H + h) * Tn + i) * (uint64_t)Tn + j);  // dV uses the dropped P
      ak += (Elem<T>::ld4(qb + (int64_t)i * p.ld + dv) + bu) * dsv;
      av += Elem<T>::ld4(gb + (int64_t)i * p.lddctx + dv) * pv;
    }
  }
  for (int o = nvec; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ak[e] += __shfl_xor(ak[e], o, 64);
      av[e] += __shfl_xor(av[e], o, 64);
    }
  }
  if (lane < nvec) {
    Elem<T>::st4(dkr + dv, ak);
    Elem<T>::st4(dvr + dv, av);
  }
}

// ---- backward, kernel 3 ("new" variant): gradient of the projected positional
// table, summed over the batch: dpos[m] = sum_{b,i} dS[b,h,i,j=m-(T-1)+i] (q_i + v)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_pos_kernel(const AttnBwdP p, float* dpos) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + w, h = blockIdx.y;
  const int nbg = gridDim.z, bper = (p.B + nbg - 1) / nbg;  // batch groups: more blocks than (L/4) x H
  const int b0 = blockIdx.z * bper, b1 = min(p.B, b0 + bper);
  const int Tn = p.T, dk = p.dk, L = p.variant == VAR_LEGACY ? Tn : 2 * Tn - 1;
  if (m >= L) return;
  const int hc = h * dk;
  const int nvec = dk >> 2, parts = 64 / nvec;
  const int dv = (lane % nvec) * 4, part = lane / nvec;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias_v + hc + dv);
  const int ilo = max(0, Tn - 1 - m), ihi = min(Tn - 1, 2 * Tn - 2 - m);  // 0 <= j = m-(T-1)+i < T
  if (part < parts)
    for (int b = b0; b < b1; ++b) {
      const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
      const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
      const float* dsb = p.dS + ((int64_t)b * p.H + h) * Tn * Tn;
      for (int i = ilo + part; i <= ihi && i < len; i += parts) {
        const int j = m - (Tn - 1) + i;
        acc += (Elem<T>::ld4(qb + (int64_t)i * p.ld + dv) + bv) * dsb[(int64_t)i * Tn + j];
      }
      if (p.variant == VAR_LEGACY)  // keys j = m + i + 2 of query row i took (q_{i+1} + v) . pos[m] (see attn_bwd_row_kernel)
        for (int i = part; i <= Tn - 3 - m && i < len; i += parts)
          acc += (Elem<T>::ld4(qb + (int64_t)(i + 1) * p.ld + dv) + bv) * dsb[(int64_t)i * Tn + m + i + 2];
    }
  for (int o = nvec; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane < nvec) {
    float* d = dpos + (int64_t)m * (p.H * dk) + hc + dv;
    if (nbg == 1) *reinterpret_cast<f32x4*>(d) = acc;
    else
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(d + e, acc[e]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward on the matrix cores (round 4; bf16, dk in {64, 128}, the "new" rel-pos table or no table, T small enough for the
// LDS images below).  Same tiling idiom as attn_fwd_mfma_kernel -- a wave owns 16 rows, the "other" sequence dimension is
// the MFMA K or N -- three kernels replacing attn_bwd_row4 / col / pos:
//   q  (64 query rows per block): dP^T = V_frag x dctx^T, dS from the saved probabilities in the softmax layout of the
//      forward (a lane: one query row, 4 consecutive keys per fragment), dq^T = K^T_frag x dS (transposing LDS reads, as the
//      forward's context product) + Pos^T_frag x D with D[i][m] = dS[i][m - (T-1) + i] -- the inverse of the forward's skew,
//      through the same 16 x 32 f32 patch per wave; du / dvb = column sums of the two dq parts.
//   k  (64 keys per block): dK^T = (q+u)^T_frag x dS, dV^T = dctx^T_frag x P_dropped; the B operands (a key, 8 query rows)
//      come straight from global memory (coalesced over the keys).
//   pos (16 table rows per block, all utterances of a batch group): dpos^T = (q+v)^T_frag x D.
// dS / P enter the MFMAs rounded to bf16 (the row kernels keep them in f32): within the bf16 tolerance of the tests.
template <int DK>
__device__ __forceinline__ bf16x8_t mf_tr_frag(const bf16_raw* img, int base, int mf, int lq, int lg) {
  // A operand [16 channels mf*16 ..][32 rows base ..] of a row-major [row][DK] image with the vsw swizzle: slots 0-3 = rows
  // base + 4 lg + (0..3), slots 4-7 = rows base + 16 + 4 lg + (0..3)
  constexpr int CPR = DK / 8;
  const int col = mf * 16 + 4 * (lq & 3);
  const int ra = base + 4 * lg + (lq >> 2), rb = ra + 16;
  const mf_v4s lo = mf_tr_read(img + ra * DK + (((col >> 3) ^ vsw<CPR>(ra)) << 3) + (col & 7));
  const mf_v4s hi = mf_tr_read(img + rb * DK + (((col >> 3) ^ vsw<CPR>(rb)) << 3) + (col & 7));
  typedef __attribute__((ext_vector_type(8))) short v8s_;
  v8s_ av;
  av[0] = lo[0]; av[1] = lo[1]; av[2] = lo[2]; av[3] = lo[3];
  av[4] = hi[0]; av[5] = hi[1]; av[6] = hi[2]; av[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, av);
}

constexpr int BQ_MAXNF = 16;  // key fragments of 16 held in registers: T <= 256

template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_q_mfma_kernel(const AttnBwdP p) {
  typedef bf16_raw T;
  constexpr int CPR = DK / 8, KS = DK / 32, MAXNF = BQ_MAXNF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tn = p.T, Tp = (Tn + 31) & ~31, PW = Tp + 64;
  // LDS: region A = max(Tp, PW) rows -- V for the dP product, then (after a barrier) the block's window of the positional
  // table; region B = K (Tp rows).  All three at once would be 208 KB at T = 256, dk = 128.
  const int rowsA = p.variant == VAR_NEW ? PW : Tp;
  uint4* Vk = reinterpret_cast<uint4*>(smem);                   // V  [Tp][CPR], row reads (ksw swizzle)
  T* Pt = reinterpret_cast<T*>(smem);                           // table rows mlo .. mlo + PW - 1, transposing reads (vsw swizzle)
  T* Kt = reinterpret_cast<T*>(smem) + (size_t)rowsA * DK;      // K  [Tp][DK], transposing reads
  float* Rs = reinterpret_cast<float*>(Kt + (size_t)Tp * DK);   // [4 waves][16][33]
  float* red = Rs + 4 * 16 * 33;                                // [4 waves][2][DK]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int ib0 = blockIdx.x * 64, i0 = ib0 + w * 16;
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * DK;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)b * Tn * p.ld + hc;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)b * Tn * p.ld + hc;
  const T* pb = p.pos ? reinterpret_cast<const T*>(p.pos) + hc : nullptr;
  const int mlo = Tn - 1 - (ib0 + 63);
  for (int idx = tid; idx < Tp * CPR; idx += 256) {
    const int row = idx / CPR, c = idx % CPR;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
    if (row < len) {
      kv = *reinterpret_cast<const uint4*>(kb + (int64_t)row * p.ld + c * 8);
      vv = *reinterpret_cast<const uint4*>(vb + (int64_t)row * p.ld + c * 8);
    }
    Vk[row * CPR + (c ^ ksw<CPR>(row))] = vv;
    *reinterpret_cast<uint4*>(Kt + row * DK + ((c ^ vsw<CPR>(row)) << 3)) = kv;
  }
  const int iq = i0 + lq;
  const bool qvalid = iq < Tn, live = iq < len;
  uint4 g[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    g[ks] = make_uint4(0, 0, 0, 0);
    if (live) g[ks] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.dctx) + ((int64_t)b * Tn + iq) * p.lddctx + hc + ks * 32 + lg * 8);
  }
  __syncthreads();
  f32x4 oc[DK / 16], op[DK / 16];
#pragma unroll
  for (int mf = 0; mf < DK / 16; ++mf) oc[mf] = op[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 ds[MAXNF];
#pragma unroll
  for (int nf = 0; nf < MAXNF; ++nf) ds[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nfr = (i0 < Tn && i0 < len) ? (len + 15) >> 4 : 0;
  if (i0 < Tn) {
    // ---- dP, dS ----
    f32x4 pr[MAXNF];
    const int64_t prow = (((int64_t)b * p.H + h) * Tn + iq) * Tn;
    float dsum = 0.f;
#pragma unroll
    for (int nf = 0; nf < MAXNF; ++nf) {
      pr[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (nf < nfr) {
        const int krow = nf * 16 + lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4 vf = Vk[krow * CPR + ((ks * 4 + lg) ^ ksw<CPR>(krow))];
          ds[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, g[ks]), ds[nf], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nf * 16 + 4 * lg + r;
          if (live && j < len) {
            pr[nf][r] = p.probs[prow + j];
            if (p.drop_thresh16) ds[nf][r] *= attn_keep(p, (uint64_t)prow + j);
            dsum += pr[nf][r] * ds[nf][r];
          }
        }
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
#pragma unroll
    for (int nf = 0; nf < MAXNF; ++nf) {
      if (nf * 16 < Tn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nf * 16 + 4 * lg + r;
          const float v = (nf < nfr && live && j < len) ? pr[nf][r] * (ds[nf][r] - dsum) * p.scale : 0.f;
          ds[nf][r] = v;
          if (qvalid && j < Tn) p.dS[prow + j] = v;
        }
      }
    }
    // ---- dq, content part: K^T_frag x dS ----
    const int nsteps = (nfr + 1) >> 1;
#pragma unroll
    for (int st = 0; st < MAXNF / 2; ++st) {
      if (st < nsteps) {
        float pv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pv[r] = ds[2 * st][r];
          pv[4 + r] = ds[2 * st + 1][r];
        }
        const uint4 pf = mf_pack8(pv);
#pragma unroll
        for (int mf = 0; mf < DK / 16; ++mf)
          oc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_tr_frag<DK>(Kt, st * 32, mf, lq, lg), __builtin_bit_cast(bf16x8_t, pf), oc[mf], 0, 0, 0);
      }
    }
  }
  // ---- dq, positional part: the table window takes V's place; per key fragment the 16 x 32 patch
  // D[query][table row - m0], m0 = T-1-(i0+15) + 16 nf ----
  if (p.variant == VAR_NEW) {
    __syncthreads();  // every wave is done with V
    for (int idx = tid; idx < PW * CPR; idx += 256) {
      const int row = idx / CPR, c = idx % CPR, m = mlo + row;
      uint4 pv = make_uint4(0, 0, 0, 0);
      if (m >= 0 && m < 2 * Tn - 1) pv = *reinterpret_cast<const uint4*>(pb + (int64_t)m * p.ldpos + c * 8);
      *reinterpret_cast<uint4*>(Pt + row * DK + ((c ^ vsw<CPR>(row)) << 3)) = pv;
    }
    __syncthreads();
  }
  if (i0 < Tn) {
    if (p.variant == VAR_NEW) {
      float* Rw = Rs + w * (16 * 33);
#pragma unroll
      for (int nf = 0; nf < MAXNF; ++nf) {
        if (nf < nfr) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 15 - lq + 4 * lg + r;  // in [0, 31)
            Rw[lq * 33 + c] = ds[nf][r];
            Rw[lq * 33 + ((c + 16) & 31)] = 0.f;
          }
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_s_waitcnt(0xc07f);
          float dv[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dv[r] = Rw[lq * 33 + 4 * lg + r];
            dv[4 + r] = Rw[lq * 33 + 16 + 4 * lg + r];
          }
          const uint4 df = mf_pack8(dv);
          const int r0 = 48 - 16 * w + 16 * nf;  // window row of table row m0
#pragma unroll
          for (int mf = 0; mf < DK / 16; ++mf)
            op[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_tr_frag<DK>(Pt, r0, mf, lq, lg), __builtin_bit_cast(bf16x8_t, df), op[mf], 0, 0, 0);
        }
      }
    }
    if (qvalid) {
      T* dqr = reinterpret_cast<T*>(p.dq) + ((int64_t)b * Tn + iq) * p.lddq + hc;
#pragma unroll
      for (int mf = 0; mf < DK / 16; ++mf) Elem<T>::st4(dqr + mf * 16 + 4 * lg, oc[mf] + op[mf]);
    }
  }
  // ---- du / dvb: column sums of the two parts over the block's query rows ----
  if (p.variant != VAR_PLAIN) {
#pragma unroll
    for (int mf = 0; mf < DK / 16; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = oc[mf][r], c = op[mf][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          c += __shfl_xor(c, o, 64);
        }
        if (lq == 0) {
          red[(w * 2 + 0) * DK + mf * 16 + 4 * lg + r] = a;
          red[(w * 2 + 1) * DK + mf * 16 + 4 * lg + r] = c;
        }
      }
    __syncthreads();
    float* rep = p.scratch + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % PTPP_RED_NREP) * (2 * p.H * DK);
    for (int c = tid; c < 2 * DK; c += 256) {
      const float v = red[c] + red[2 * DK + c] + red[4 * DK + c] + red[6 * DK + c];
      atomicAdd(rep + (c < DK ? hc + c : p.H * DK + hc + (c - DK)), v);
    }
  }
}

template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_k_mfma_kernel(const AttnBwdP p, void* dk_out, void* dv_out) {
  typedef bf16_raw T;
  constexpr int CPR = DK / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tn = p.T, Tp = (Tn + 31) & ~31;
  T* Qt = reinterpret_cast<T*>(smem);   // (q + u) [Tp][DK], transposing reads
  T* Gt = Qt + (size_t)Tp * DK;         // dctx    [Tp][DK]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int j0 = blockIdx.x * 64 + w * 16, jq = j0 + lq;
  const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
  const int hc = h * DK;
  const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
  const T* gb = reinterpret_cast<const T*>(p.dctx) + (int64_t)b * Tn * p.lddctx + hc;
  for (int idx = tid; idx < Tp * CPR; idx += 256) {
    const int row = idx / CPR, c = idx % CPR;
    uint4 qv = make_uint4(0, 0, 0, 0), gv = qv;
    if (row < len) {
      qv = *reinterpret_cast<const uint4*>(qb + (int64_t)row * p.ld + c * 8);
      gv = *reinterpret_cast<const uint4*>(gb + (int64_t)row * p.lddctx + c * 8);
      if (p.variant != VAR_PLAIN) {
        const float qf[8] = {__uint_as_float(qv.x << 16), __uint_as_float(qv.x & 0xffff0000u), __uint_as_float(qv.y << 16),
                             __uint_as_float(qv.y & 0xffff0000u), __uint_as_float(qv.z << 16), __uint_as_float(qv.z & 0xffff0000u),
                             __uint_as_float(qv.w << 16), __uint_as_float(qv.w & 0xffff0000u)};
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = qf[e] + p.bias_u[hc + c * 8 + e];
        qv = mf_pack8(a);
      }
    }
    *reinterpret_cast<uint4*>(Qt + row * DK + ((c ^ vsw<CPR>(row)) << 3)) = qv;
    *reinterpret_cast<uint4*>(Gt + row * DK + ((c ^ vsw<CPR>(row)) << 3)) = gv;
  }
  __syncthreads();
  if (j0 >= Tn) return;
  f32x4 ak[DK / 16], av[DK / 16];
#pragma unroll
  for (int mf = 0; mf < DK / 16; ++mf) ak[mf] = av[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t base = ((int64_t)b * p.H + h) * Tn * Tn;
  const int nsteps = j0 < len ? (len + 31) >> 5 : 0;
  for (int st = 0; st < nsteps; ++st) {
    float dsv[8], pv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int i = st * 32 + (s >> 2) * 16 + 4 * lg + (s & 3);
      dsv[s] = pv[s] = 0.f;
      if (i < len && jq < len) {
        const int64_t e = base + (int64_t)i * Tn + jq;
        dsv[s] = p.dS[e];
        pv[s] = p.probs[e];
        if (p.drop_thresh16) pv[s] *= attn_keep(p, (uint64_t)e);
      }
    }
    const uint4 dsf = mf_pack8(dsv), pf = mf_pack8(pv);
#pragma unroll
    for (int mf = 0; mf < DK / 16; ++mf) {
      ak[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_tr_frag<DK>(Qt, st * 32, mf, lq, lg), __builtin_bit_cast(bf16x8_t, dsf), ak[mf], 0, 0, 0);
      av[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_tr_frag<DK>(Gt, st * 32, mf, lq, lg), __builtin_bit_cast(bf16x8_t, pf), av[mf], 0, 0, 0);
    }
  }
  if (jq < Tn) {
    T* dkr = reinterpret_cast<T*>(dk_out) + ((int64_t)b * Tn + jq) * p.lddq + hc;
    T* dvr = reinterpret_cast<T*>(dv_out) + ((int64_t)b * Tn + jq) * p.lddq + hc;
#pragma unroll
    for (int mf = 0; mf < DK / 16; ++mf) {
      Elem<T>::st4(dkr + mf * 16 + 4 * lg, ak[mf]);
      Elem<T>::st4(dvr + mf * 16 + 4 * lg, av[mf]);
    }
  }
}

template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_pos_mfma_kernel(const AttnBwdP p, float* dpos) {
  typedef bf16_raw T;
  constexpr int CPR = DK / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tn = p.T, Tp = (Tn + 31) & ~31;
  T* Qv = reinterpret_cast<T*>(smem);  // (q + v) of the current utterance [Tp][DK]; afterwards the waves' partial tiles (f32)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int h = blockIdx.y, hc = h * DK;
  const int m0 = blockIdx.x * 16, m = m0 + lq;
  const int nbg = gridDim.z, bper = (p.B + nbg - 1) / nbg;
  const int b0 = blockIdx.z * bper, b1 = min(p.B, b0 + bper);
  f32x4 acc[DK / 16];
#pragma unroll
  for (int mf = 0; mf < DK / 16; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  // query rows that meet this block's table rows: 0 <= j = m - (T-1) + i < T
  const int ilo = max(0, Tn - 1 - (m0 + 15)), ihi = min(Tn - 1, 2 * Tn - 2 - m0);
  for (int b = b0; b < b1; ++b) {
    const int len = p.lengths ? min(p.lengths[b], Tn) : Tn;
    const T* qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * Tn * p.ld + hc;
    __syncthreads();  // the previous utterance's reads are done
    for (int idx = tid; idx < Tp * CPR; idx += 256) {
      const int row = idx / CPR, c = idx % CPR;
      uint4 qv = make_uint4(0, 0, 0, 0);
      if (row < len && row >= (ilo & ~31) && row <= ihi) {
        qv = *reinterpret_cast<const uint4*>(qb + (int64_t)row * p.ld + c * 8);
        const float qf[8] = {__uint_as_float(qv.x << 16), __uint_as_float(qv.x & 0xffff0000u), __uint_as_float(qv.y << 16),
                             __uint_as_float(qv.y & 0xffff0000u), __uint_as_float(qv.z << 16), __uint_as_float(qv.z & 0xffff0000u),
                             __uint_as_float(qv.w << 16), __uint_as_float(qv.w & 0xffff0000u)};
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = qf[e] + p.bias_v[hc + c * 8 + e];
        qv = mf_pack8(a);
      }
      *reinterpret_cast<uint4*>(Qv + row * DK + ((c ^ vsw<CPR>(row)) << 3)) = qv;
    }
    __syncthreads();
    const float* dsb = p.dS + ((int64_t)b * p.H + h) * Tn * Tn;
    const int s_lo = ilo >> 5, s_hi = min(ihi, len - 1) >> 5;
    for (int st = s_lo + w; st <= s_hi; st += 4) {
      float dv[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int i = st * 32 + (s >> 2) * 16 + 4 * lg + (s & 3);
        const int j = m - (Tn - 1) + i;
        dv[s] = (i < len && j >= 0 && j < len) ? dsb[(int64_t)i * Tn + j] : 0.f;
      }
      const uint4 df = mf_pack8(dv);
#pragma unroll
      for (int mf = 0; mf < DK / 16; ++mf)
        acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf_tr_frag<DK>(Qv, st * 32, mf, lq, lg), __builtin_bit_cast(bf16x8_t, df), acc[mf], 0, 0, 0);
    }
  }
  __syncthreads();
  float* part = reinterpret_cast<float*>(smem);  // [4 waves][16 table rows][DK]
#pragma unroll
  for (int mf = 0; mf < DK / 16; ++mf)
    *reinterpret_cast<f32x4*>(part + ((w * 16 + lq) * DK) + mf * 16 + 4 * lg) = acc[mf];
  __syncthreads();
  const int L = 2 * Tn - 1;
  for (int idx = tid; idx < 16 * DK; idx += 256) {
    const int r = idx / DK, c = idx % DK;
    if (m0 + r >= L) continue;
    const float v = part[idx] + part[16 * DK + idx] + part[2 * 16 * DK + idx] + part[3 * 16 * DK + idx];
    float* d = dpos + (int64_t)(m0 + r) * (p.H * DK) + hc + c;
    if (nbg == 1) *d = v;
    else atomicAdd(d, v);
  }
}

bool shape_ok(int B, int T, int H, int dk) {
  return B > 0 && T > 0 && T <= 2048 && H > 0 && (dk == 64 || dk == 128 || dk == 256);
}

}  // namespace

extern "C" int ptpp_attention_fwd(const void* q, const void* k, const void* v, const void* pos, const float* bias_u,
                                  const float* bias_v, void* ctx, float* probs, const int32_t* lengths, int B, int T_,
                                  int H, int dk, int ld, int ldpos, int ldctx, int variant, float drop_p,
                                  uint64_t drop_seed, int dtype, void* stream) {
  PTPP_CHECK_ARG(q && k && v && ctx, "attention_fwd: null pointer");
  PTPP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attention_fwd: bad dropout p");
  PTPP_CHECK_ARG(shape_ok(B, T_, H, dk), "attention_fwd: unsupported shape B=%d T=%d H=%d dk=%d", B, T_, H, dk);
  PTPP_CHECK_ARG(variant >= 0 && variant <= 2, "attention_fwd: bad variant");
  PTPP_CHECK_ARG(variant == VAR_PLAIN || (pos && bias_u && bias_v), "attention_fwd: rel-pos variant needs pos/u/v");
  PTPP_CHECK_ARG(ld % 4 == 0 && ldctx % 4 == 0 && (variant == VAR_PLAIN || ldpos % 4 == 0), "attention_fwd: strides");
  AttnP p{q, k, v, pos, bias_u, bias_v, ctx, probs, lengths, B, T_, H, dk, ld, ldpos, ldctx, variant,
          1.0f / sqrtf((float)dk)};
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // bf16, T <= 256, dk 64 / 128, no legacy table: K / V resident in LDS, products on the matrix cores (PTPP_ATTN_MFMA=0: row kernel)
  static const char* mfma_env = getenv("PTPP_ATTN_MFMA");
  if (dtype == PTPP_BF16 && T_ <= 256 && (dk == 64 || dk == 128) && variant != VAR_LEGACY && ld % 8 == 0 && ldctx % 4 == 0 &&
      (variant == VAR_PLAIN || ldpos % 8 == 0) && ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
      (!pos || ((uintptr_t)pos % 16) == 0) && !(mfma_env && mfma_env[0] == '0')) {
    const int Tp = (T_ + 31) & ~31;
    const size_t sm = (size_t)Tp * dk * 2 * 2 + 4 * 16 * 33 * sizeof(float);
    dim3 g((T_ + MF_QT - 1) / MF_QT, H, B);
    if (dk == 128) {
      if (sm > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<128>), (int)sm, "attention_fwd")) return PTPP_ELAUNCH;
      hipLaunchKernelGGL(attn_fwd_mfma_kernel<128>, g, dim3(256), sm, st, p);
    } else {
      if (sm > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<64>), (int)sm, "attention_fwd")) return PTPP_ELAUNCH;
      hipLaunchKernelGGL(attn_fwd_mfma_kernel<64>, g, dim3(256), sm, st, p);
    }
    PTPP_CHECK_LAUNCH("attention_fwd (mfma)");
    return PTPP_OK;
  }
  const size_t smem = (size_t)4 * (3 * dk + ((T_ + 3) & ~3)) * sizeof(float);
  dim3 grid((T_ + 3) / 4, H, B);
  if (dtype == PTPP_F32) hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), smem, st, p);
  else if (dtype == PTPP_BF16) hipLaunchKernelGGL(attn_fwd_kernel<bf16_raw>, grid, dim3(256), smem, st, p);
  else PTPP_CHECK_ARG(false, "attention_fwd: bad dtype");
  PTPP_CHECK_LAUNCH("attention_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_attention_bwd(const void* q, const void* k, const void* v, const void* pos, const float* bias_u,
                                  const float* bias_v, const float* probs, const void* dctx, float* dS, void* dq,
                                  void* dk_out, void* dv_out, float* dpos, float* du, float* dvb,
                                  const int32_t* lengths, int B, int T_, int H, int dk, int ld, int ldpos, int lddctx,
                                  int lddq, int variant, float drop_p, uint64_t drop_seed, int dtype, void* scratch,
                                  size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(q && k && v && probs && dctx && dS && dq && dk_out && dv_out, "attention_bwd: null pointer");
  PTPP_CHECK_ARG(shape_ok(B, T_, H, dk), "attention_bwd: unsupported shape");
  PTPP_CHECK_ARG(variant >= 0 && variant <= 2, "attention_bwd: bad variant");
  PTPP_CHECK_ARG(variant == VAR_PLAIN || (pos && bias_u && bias_v && dpos && du && dvb), "attention_bwd: rel-pos args");
  PTPP_CHECK_ARG(variant == VAR_PLAIN || red_scratch_ok(scratch, scratch_bytes, 2 * H * dk),
                 "attention_bwd: reduction scratch missing or too small");
  AttnBwdP p{q, k, v, pos, dctx, bias_u, bias_v, probs, dS, dq, du, dvb, reinterpret_cast<float*>(scratch), lengths,
             B, T_, H, dk, ld, ldpos, lddctx, lddq, variant, 1.0f / sqrtf((float)dk)};
  PTPP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attention_bwd: bad dropout p");
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  RedSlot slot{nullptr, 0};
  if (variant != VAR_PLAIN) {
    slot = red_take(scratch, scratch_bytes, 2 * H * dk, st);  // (a private arena slice when the finishing launch is deferred)
    p.scratch = reinterpret_cast<float*>(slot.ptr);
  }
  dim3 grid((T_ + 3) / 4, H, B);
  dim3 grid_row((T_ + 4 * ROWS_PER_WAVE - 1) / (4 * ROWS_PER_WAVE), H, B);
  const int Tpad = (T_ + 3) & ~3;
  const size_t smem = (size_t)(4 * (dk + Tpad + (variant == VAR_LEGACY ? Tpad : 0)) + 8 * dk) * sizeof(float);
  // dpos ((2T-1) rows, legacy: T rows): batch groups so that (L/4) x H x groups >= ~1024 blocks; groups > 1 accumulate
  // with atomics into the buffer zeroed here
  const int L = variant == VAR_LEGACY ? T_ : 2 * T_ - 1;
  // bf16, dk 64 / 128, no legacy table, the LDS images fit: the three products on the matrix cores (PTPP_ATTN_BWD_MFMA=0: the
  // row / column kernels below)
  {
    static const char* bm_env = getenv("PTPP_ATTN_BWD_MFMA");
    const int Tp = (T_ + 31) & ~31;
    const size_t sm_q = (size_t)((variant == VAR_NEW ? Tp + 64 : Tp) + Tp) * dk * 2 + (4 * 16 * 33 + 8 * dk) * sizeof(float);
    const size_t sm_k = (size_t)Tp * dk * 2 * 2;
    const size_t sm_p = (size_t)Tp * dk * 2 > (size_t)4 * 16 * dk * 4 ? (size_t)Tp * dk * 2 : (size_t)4 * 16 * dk * 4;
    const uintptr_t al = (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dctx | (uintptr_t)pos;
    if (dtype == PTPP_BF16 && (dk == 64 || dk == 128) && variant != VAR_LEGACY && T_ <= 16 * BQ_MAXNF && sm_q <= 160 * 1024 && ld % 8 == 0 &&
        lddctx % 8 == 0 && lddq % 4 == 0 && (variant == VAR_PLAIN || ldpos % 8 == 0) && (al & 15) == 0 && !(bm_env && bm_env[0] == '0')) {
      const dim3 gq((T_ + 63) / 64, H, B);
      int nbg2 = (256 + ((L + 15) / 16) * H - 1) / (((L + 15) / 16) * H);
      if (nbg2 > B) nbg2 = B;
      if (variant != VAR_PLAIN && nbg2 > 1) (void)hipMemsetAsync(dpos, 0, (size_t)L * H * dk * sizeof(float), st);
#define ATTN_BWD_MF(DKV)                                                                                                              \
  {                                                                                                                                   \
    static bool attr_done = false;                                                                                                    \
    if (!attr_done) {                                                                                                                 \
      if (!ptpp_lds_limit(reinterpret_cast<const void*>(attn_bwd_q_mfma_kernel<DKV>), 160 * 1024, "attention_bwd") ||                         \
          !ptpp_lds_limit(reinterpret_cast<const void*>(attn_bwd_k_mfma_kernel<DKV>), 160 * 1024, "attention_bwd") ||                         \
          !ptpp_lds_limit(reinterpret_cast<const void*>(attn_bwd_pos_mfma_kernel<DKV>), 160 * 1024, "attention_bwd"))                         \
        return PTPP_ELAUNCH;                                                                                                            \
      attr_done = true;                                                                                                               \
    }                                                                                                                                 \
    hipLaunchKernelGGL(attn_bwd_q_mfma_kernel<DKV>, gq, dim3(256), sm_q, st, p);                                                      \
    hipLaunchKernelGGL(attn_bwd_k_mfma_kernel<DKV>, gq, dim3(256), sm_k, st, p, dk_out, dv_out);                                      \
    if (variant != VAR_PLAIN)                                                                                                         \
      hipLaunchKernelGGL(attn_bwd_pos_mfma_kernel<DKV>, dim3((L + 15) / 16, H, nbg2), dim3(256), sm_p, st, p, dpos);                  \
  }
      if (dk == 128) ATTN_BWD_MF(128) else ATTN_BWD_MF(64)
#undef ATTN_BWD_MF
      if (variant != VAR_PLAIN) red_finish(slot, 2 * H * dk, du, H * dk, dvb, 1, st);
      PTPP_CHECK_LAUNCH("attention_bwd (mfma)");
      return PTPP_OK;
    }
  }
  const int lblk = (L + 3) / 4 * H;
  int nbg = (1024 + lblk - 1) / lblk;
  if (nbg > B) nbg = B;
  if (variant != VAR_PLAIN && nbg > 1) (void)hipMemsetAsync(dpos, 0, (size_t)L * H * dk * sizeof(float), st);
  // (PTPP_ATTN_BWD_ROW4=0: one query row per wave for every table)
  static const char* row4_env = getenv("PTPP_ATTN_BWD_ROW4");
  const bool row4 = variant != VAR_LEGACY && !(row4_env && row4_env[0] == '0');
  const dim3 grid_row4((T_ + 15) / 16, H, B);
  const size_t smem4 = (size_t)(4 * 4 * (dk + Tpad) + 8 * dk) * sizeof(float);
#define ATTN_BWD(TT)                                                                                  \
  if (row4) {                                                                                         \
    if (smem4 > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(attn_bwd_row4_kernel<TT>), (int)smem4, "attention_bwd")) \
      return PTPP_ELAUNCH;                                                                            \
    hipLaunchKernelGGL(attn_bwd_row4_kernel<TT>, grid_row4, dim3(256), smem4, st, p);                 \
  } else                                                                                              \
    hipLaunchKernelGGL(attn_bwd_row_kernel<TT>, grid_row, dim3(256), smem, st, p);                    \
  hipLaunchKernelGGL(attn_bwd_col_kernel<TT>, grid, dim3(256), 0, st, p, dk_out, dv_out);             \
  if (variant != VAR_PLAIN)                                                                           \
    hipLaunchKernelGGL(attn_bwd_pos_kernel<TT>, dim3((L + 3) / 4, H, nbg), dim3(256), 0, st, p, dpos);
  if (dtype == PTPP_F32) { ATTN_BWD(float) }
  else if (dtype == PTPP_BF16) { ATTN_BWD(bf16_raw) }
  else PTPP_CHECK_ARG(false, "attention_bwd: bad dtype");
#undef ATTN_BWD
  if (variant != VAR_PLAIN) red_finish(slot, 2 * H * dk, du, H * dk, dvb, 1, st);
  PTPP_CHECK_LAUNCH("attention_bwd");
  return PTPP_OK;
}

// ---- windowed relative-position attention (modules/transformer.py:59-137): the row kernels with VAR_WINDOW ----------------
extern "C" int ptpp_attention_win_fwd(const void* q, const void* k, const void* v, const float* emb_k, const float* emb_v, void* ctx,
                                      float* probs, const int32_t* lengths, int B, int T_, int H, int dk, int ld, int ldctx, int window,
                                      float drop_p, uint64_t drop_seed, int dtype, void* stream) {
  PTPP_CHECK_ARG(q && k && v && ctx && emb_k && emb_v, "attention_win_fwd: null pointer");
  PTPP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attention_win_fwd: bad dropout p");
  PTPP_CHECK_ARG(shape_ok(B, T_, H, dk) && window >= 0 && window <= 64, "attention_win_fwd: unsupported shape B=%d T=%d H=%d dk=%d window=%d", B,
                 T_, H, dk, window);
  PTPP_CHECK_ARG(ld % 4 == 0 && ldctx % 4 == 0 && (((uintptr_t)emb_k | (uintptr_t)emb_v) & 15) == 0, "attention_win_fwd: strides / alignment");
  AttnP p{q, k, v, nullptr, nullptr, nullptr, ctx, probs, lengths, B, T_, H, dk, ld, 0, ldctx, VAR_WINDOW, 1.0f / sqrtf((float)dk)};
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  p.emb_k = emb_k; p.emb_v = emb_v; p.window = window;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t smem = (size_t)4 * (3 * dk + ((T_ + 3) & ~3)) * sizeof(float);
  dim3 grid((T_ + 3) / 4, H, B);
  if (dtype == PTPP_F32) hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), smem, st, p);
  else if (dtype == PTPP_BF16) hipLaunchKernelGGL(attn_fwd_kernel<bf16_raw>, grid, dim3(256), smem, st, p);
  else PTPP_CHECK_ARG(false, "attention_win_fwd: bad dtype");
  PTPP_CHECK_LAUNCH("attention_win_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_attention_win_bwd(const void* q, const void* k, const void* v, const float* emb_k, const float* emb_v, const float* probs,
                                      const void* dctx, float* dS, void* dq, void* dk_out, void* dv_out, float* demb_k, float* demb_v,
                                      const int32_t* lengths, int B, int T_, int H, int dk, int ld, int lddctx, int lddq, int window,
                                      float drop_p, uint64_t drop_seed, int dtype, void* stream) {
  PTPP_CHECK_ARG(q && k && v && emb_k && emb_v && probs && dctx && dS && dq && dk_out && dv_out && demb_k && demb_v,
                 "attention_win_bwd: null pointer");
  PTPP_CHECK_ARG(shape_ok(B, T_, H, dk) && window >= 0 && window <= 64, "attention_win_bwd: unsupported shape");
  PTPP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "attention_win_bwd: bad dropout p");
  AttnBwdP p{q, k, v, nullptr, dctx, nullptr, nullptr, probs, dS, dq, nullptr, nullptr, nullptr, lengths,
             B, T_, H, dk, ld, 0, lddctx, lddq, VAR_WINDOW, 1.0f / sqrtf((float)dk)};
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  p.emb_k = emb_k; p.emb_v = emb_v; p.window = window;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nr = 2 * window + 1;
  (void)hipMemsetAsync(demb_k, 0, (size_t)nr * dk * sizeof(float), st);
  (void)hipMemsetAsync(demb_v, 0, (size_t)nr * dk * sizeof(float), st);
  const int Tpad = (T_ + 3) & ~3;
  const size_t smem = (size_t)(4 * (dk + Tpad) + 8 * dk) * sizeof(float);
  dim3 grid((T_ + 3) / 4, H, B);
  int nbg = (256 + nr * H - 1) / (nr * H);
  if (nbg > B) nbg = B;
#define ATTN_WIN_BWD(TT)                                                                                  \
  hipLaunchKernelGGL(attn_bwd_row_kernel<TT>, grid, dim3(256), smem, st, p);                              \
  hipLaunchKernelGGL(attn_bwd_col_kernel<TT>, grid, dim3(256), 0, st, p, dk_out, dv_out);                 \
  hipLaunchKernelGGL(attn_bwd_win_emb_kernel<TT>, dim3(nr, H, nbg), dim3(256), 0, st, p, demb_k, demb_v);
  if (dtype == PTPP_F32) { ATTN_WIN_BWD(float) }
  else if (dtype == PTPP_BF16) { ATTN_WIN_BWD(bf16_raw) }
  else PTPP_CHECK_ARG(false, "attention_win_bwd: bad dtype");
#undef ATTN_WIN_BWD
  PTPP_CHECK_LAUNCH("attention_win_bwd");
  return PTPP_OK;
}
