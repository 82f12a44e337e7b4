// Shared pieces of the channels-last Conv1d kernels: the argument block, the LDS swizzle, the
// weight-row permutation of the bf16 tiles and the fused epilogue.
#pragma once
#include "ptpp_common.h"

namespace {

// XOR swizzle of the 16-byte chunk index inside an LDS row so that the
// ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS table) hit distinct slots.
template <int NCH>
__device__ __forceinline__ int swz(int row);
template <>
__device__ __forceinline__ int swz<8>(int row) { return (row >> 1) & 7; }
template <>
__device__ __forceinline__ int swz<4>(int row) { return (-(row >> 2)) & 3; }

// LDS row of weight-tile row n.  MFMA tile fn of the wave takes LDS rows fn*16 .. fn*16+15
// and hands lane group lg, register r the row fn*16 + 4*lg + r.  bf16: place channel
// u = h*32 + lg*8 + f*4 + r (within the wave's FN*16 span) at the row of tile fn = 2h + f.
template <typename T, int FN>
__device__ __forceinline__ int wperm(int n) {
  if constexpr (sizeof(T) == 2 && FN % 2 == 0) {
    const int u = n % (16 * FN);
    const int h = u >> 5, lg = (u >> 3) & 3, f = (u >> 2) & 1, r = u & 3;
    return (n - u) + (2 * h + f) * 16 + 4 * lg + r;
  } else {
    return n;
  }
}

struct ConvP {
  const void* x;
  const void* wp;
  const float* bias;
  const void* res;
  const void* res2;
  void* y;
  const int* lengths;
  int B, T, Cin, Cout, ks, dil, pad;
  int ldx, ldy, ldr, ldr2;
  int cinp;
  int act, in_mask, out_mask;
  float out_scale, res_scale;
  int nMT, nNT;
  unsigned drop_thresh16;  // 0 = no dropout
  float drop_inv_keep;
  unsigned long long drop_seed;
  float* ws;   // split-K: f32 partial sums [nsplit][B][T][Cout] (no epilogue), summed by conv_splitk_finish_kernel
  int nsplit;  // 1 = no split
  // fused DiffNet "post" epilogue (conv1d_glds.h, ptpp_conv1d_diffnet_post): Cout = 2 * post_C, res = x, y = xn
  float* post_skip;         // nullptr = ordinary epilogue
  const float* post_dnext;  // (B, post_C) or nullptr
  void* post_yin;           // (B, T, post_C) or nullptr
  int post_C, post_init;
  // fused DiffNet gate backward (conv1d_glds.h, ptpp_conv1d_gate_bwd): Cout = C channels of dg, never stored;
  // da[:, c] / da[:, C + c] (row stride gate_ldda) from the saved pre-activation gate_a (B, T, 2C)
  const void* gate_a;  // nullptr = not this mode
  void* gate_da;
  int gate_ldda;
  // fused DiffNet gate FORWARD that also keeps the pre-activation for the backward (training; conv1d_glds.h,
  // ptpp_conv1d_gate_fwd_save): act = PTPP_ACT_GATE, y = g (Cout / 2 channels), gate_save (B, T, Cout) in the STANDARD
  // [gate | filter] channel order, row stride gate_lds
  void* gate_save;  // nullptr = not this mode
  int gate_lds;
};

// Fused epilogue of one (BM x BN) tile: acc[fm][fn] is the MFMA accumulator of the wave's
// fragment (fm, fn) with the WEIGHTS as the "A" operand, so a lane holds consecutive output
// channels of output row t0 + (wm*FM + fm)*16 + (lane & 15).
template <typename T, int FM, int FN, int ACT>
__device__ __forceinline__ void conv_epilogue_act(const ConvP& p, f32x4 (&acc)[FM][FN], int b, int t0, int n0, int wm, int wn,
                                                  int lane, int len) {
  const int lr = lane & 15, lg = lane >> 4;
  // ---- epilogue: lane holds channels co..co+3 of row t for each fragment ----
  T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.T * p.ldy;
  const T* rb = p.res ? reinterpret_cast<const T*>(p.res) + (int64_t)b * p.T * p.ldr : nullptr;
  const T* r2b = p.res2 ? reinterpret_cast<const T*>(p.res2) + (int64_t)b * p.T * p.ldr2 : nullptr;
  // (epilogue parameters as local scalars: lambdas that capture the by-value argument block
  //  itself make the compiler keep a copy of it in scratch memory)
  const float* const e_bias = p.bias;
  const int e_T = p.T, e_Cout = p.Cout, e_ldy = p.ldy, e_ldr = p.ldr, e_ldr2 = p.ldr2;
  const float e_scale = p.out_scale, e_rscale = p.res_scale, e_dinv = p.drop_inv_keep;
  const unsigned e_dth = p.drop_thresh16;
  const unsigned long long e_dseed = p.drop_seed;
  const bool e_omask = p.out_mask != 0;
  const bool e_al16 = ((uintptr_t)p.y & 15) == 0 && (!p.res || ((uintptr_t)p.res & 15) == 0) &&
                      (!p.res2 || ((uintptr_t)p.res2 & 15) == 0);
  const bool vec_ok = ((e_Cout & 3) == 0);
  // conv term of 4 consecutive channels: bias, activation, mask, scale, dropout
  auto finish4 = [&](f32x4 v, int t, int co, bool keep) {
    if (e_bias) v += *reinterpret_cast<const f32x4*>(e_bias + co);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep ? act_apply_c<ACT>(v[e], p.act) * e_scale : 0.f;
    if (e_dth)
      v *= drop_mask4(e_dseed, (uint64_t)(((int64_t)b * e_T + t) * e_Cout + co) >> 2, e_dth, e_dinv);
    return v;
  };
  // 4 channels starting at co, vector path when Cout % 4 == 0, per-element otherwise
  auto out4 = [&](f32x4 acc4, int t, int co, bool keep) {
    if (vec_ok) {
      f32x4 v = finish4(acc4, t, co, keep);
      if (rb) v += Elem<T>::ld4(rb + (int64_t)t * e_ldr + co) * e_rscale;
      if (r2b) v += Elem<T>::ld4(r2b + (int64_t)t * e_ldr2 + co);
      Elem<T>::st4(yb + (int64_t)t * e_ldy + co, v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < e_Cout) {
          float u = acc4[e] + (e_bias ? e_bias[co + e] : 0.f);
          u = keep ? act_apply_c<ACT>(u, p.act) * e_scale : 0.f;
          if (rb) u += Elem<T>::ld(rb + (int64_t)t * e_ldr + co + e) * e_rscale;
          if (r2b) u += Elem<T>::ld(r2b + (int64_t)t * e_ldr2 + co + e);
          Elem<T>::st(yb + (int64_t)t * e_ldy + co + e, u);
        }
      }
    }
  };
  if constexpr (sizeof(T) == 2 && FN % 2 == 0) {
    // bf16: fragments (2h, 2h+1) of a lane are 8 consecutive channels (see wperm)
    const bool vec8 = (e_Cout & 7) == 0 && (e_ldy & 7) == 0 && e_al16 && (!rb || (e_ldr & 7) == 0) &&
                      (!r2b || (e_ldr2 & 7) == 0);
    if (vec8) {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int t = t0 + (wm * FM + fm) * 16 + lr;
        const bool keep = !(e_omask && t >= len);
#pragma unroll
        for (int h = 0; h < FN / 2; ++h) {
          const int co = n0 + wn * FN * 16 + h * 32 + lg * 8;
          if (t < e_T && co < e_Cout) {
            f32x4 v0 = finish4(acc[fm][2 * h], t, co, keep), v1 = finish4(acc[fm][2 * h + 1], t, co + 4, keep);
            if (rb) {
              const uint4 r = *reinterpret_cast<const uint4*>(rb + (int64_t)t * e_ldr + co);
              v0[0] += H2<T>::lo(r.x) * e_rscale; v0[1] += H2<T>::hi(r.x) * e_rscale;
              v0[2] += H2<T>::lo(r.y) * e_rscale; v0[3] += H2<T>::hi(r.y) * e_rscale;
              v1[0] += H2<T>::lo(r.z) * e_rscale; v1[1] += H2<T>::hi(r.z) * e_rscale;
              v1[2] += H2<T>::lo(r.w) * e_rscale; v1[3] += H2<T>::hi(r.w) * e_rscale;
            }
            if (r2b) {
              const uint4 r = *reinterpret_cast<const uint4*>(r2b + (int64_t)t * e_ldr2 + co);
              v0[0] += H2<T>::lo(r.x); v0[1] += H2<T>::hi(r.x);
              v0[2] += H2<T>::lo(r.y); v0[3] += H2<T>::hi(r.y);
              v1[0] += H2<T>::lo(r.z); v1[1] += H2<T>::hi(r.z);
              v1[2] += H2<T>::lo(r.w); v1[3] += H2<T>::hi(r.w);
            }
            if constexpr (ACT == PTPP_ACT_GATE) {
              // fused DiffNet gate: the 8 channels are [4 "gate" | their 4 "filter" partners] (weights
              // packed in that interleaved order); y has Cout / 2 channels
              uint2 o;
              float gte[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) gte[e] = keep ? gate_fast(v0[e], v1[e]) : 0.f;
              o.x = H2<T>::pack(gte[0], gte[1]);
              o.y = H2<T>::pack(gte[2], gte[3]);
              *reinterpret_cast<uint2*>(yb + (int64_t)t * e_ldy + (co >> 1)) = o;
              continue;
            }
            uint4 o;
            o.x = H2<T>::pack(v0[0], v0[1]);
            o.y = H2<T>::pack(v0[2], v0[3]);
            o.z = H2<T>::pack(v1[0], v1[1]);
            o.w = H2<T>::pack(v1[2], v1[3]);
            *reinterpret_cast<uint4*>(yb + (int64_t)t * e_ldy + co) = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int t = t0 + (wm * FM + fm) * 16 + lr;
        const bool keep = !(e_omask && t >= len);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          const int c4 = n0 + wn * FN * 16 + (fn >> 1) * 32 + lg * 8 + (fn & 1) * 4;
          if (t < e_T && c4 < e_Cout) out4(acc[fm][fn], t, c4, keep);
        }
      }
    }
  } else {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const int t = t0 + (wm * FM + fm) * 16 + lr;
      const bool keep = !(e_omask && t >= len);
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int co = n0 + (wn * FN + fn) * 16 + lg * 4;
        if (t < e_T && co < e_Cout) out4(acc[fm][fn], t, co, keep);
      }
    }
  }
}

// the epilogue specialised for the (wave-uniform) activation: see act_apply_c.  bf16 only: the f32 kernels serve the
// parity mode, where one generic copy keeps the library small and the build short.
template <typename T, int FM, int FN>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[FM][FN], int b, int t0, int n0, int wm, int wn,
                                              int lane, int len) {
  if constexpr (sizeof(T) == 2 && FN % 2 == 0) {
    act_dispatch(p.act, [&](auto tag) __attribute__((always_inline)) {
      conv_epilogue_act<T, FM, FN, decltype(tag)::value>(p, acc, b, t0, n0, wm, wn, lane, len);
    });
  } else {
    conv_epilogue_act<T, FM, FN, -1>(p, acc, b, t0, n0, wm, wn, lane, len);
  }
}

}  // namespace
