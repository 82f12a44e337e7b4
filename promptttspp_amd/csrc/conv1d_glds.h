// LDS-DMA pipelined variant of the channels-last bf16 Conv1d (included by conv1d_cl.hip).
//
// Same GEMM view, LDS images, swizzle, weight-row permutation and epilogue as conv1d_cl_kernel, but
// the operands travel global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write
// pass) one K step ahead of the MFMAs, and a block is 4 waves with 64 x 64 (large grids) or 32 x 64
// (small grids) wave tiles instead of 16 waves of 32 x 32: half the LDS read traffic per MFMA
// (MI355X_MICROARCH.md: the 32 x 32 tiles need 512 LDS cycles per 515 MFMA cycles of a K step).
// Measurements of the variants: profiles/r02_conv_glds.txt.
//
//   step s = (Cin chunk ci of 64 channels, tap j);  ring of D weight stages, two x windows
//     top of step s :  s_waitcnt vmcnt(..)   this wave's pieces of W(s) (and of the x window) have landed
//                      s_waitcnt lgkmcnt(0)  this wave's LDS reads of step s-1 have returned
//                      s_barrier             -> everyone's pieces landed, stage (s-1) % D is free
//     then          :  issue x window of chunk ci+1 (first tap of a chunk only), issue W(s + D - 1)
//                      MFMAs of step s from stage s % D and window ci & 1
//
// The LDS-DMA is issued from inline asm (the compiler neither counts nor drains it, so the plain
// C++ LDS reads keep their compiler-scheduled lgkmcnt ladders); an LDS-DMA writes 1 KiB
// lane-linear, so the XOR swizzle and the weight-row permutation are applied to the per-lane
// SOURCE address.  Rows outside the utterance come from a zero page.
#pragma once
#include "conv1d_common.h"
#include "lds_dma.h"

namespace {

// inverse of wperm<bf16, FN>: weight-tile row held by LDS row q
template <int FN>
__device__ __forceinline__ int wperm_inv(int q) {
  if constexpr (FN % 2 == 0) {
    const int u = q % (16 * FN);
    const int tile = u >> 4, lg = (u >> 2) & 3, r = u & 3;
    return (q - u) + (tile >> 1) * 32 + lg * 8 + (tile & 1) * 4 + r;
  } else {
    return q;
  }
}

// Epilogue through an LDS image of the output tile.  The MFMA layout gives a lane 16 bytes of ONE row and its
// neighbours in lane order other rows, so direct stores / residual loads reach memory as 64 separate 16-byte
// requests per wave-instruction (measured: 11 us of a 23 us block for 32 KiB in + 32 KiB out, 28 us with 4 waves).
// Here the residual rows are fetched row-contiguous (16 lanes = one 256-byte row), parked in the tile image,
// every lane updates its own 16-byte slots in place (same arithmetic and rounding as conv_epilogue), and the block
// stores the image row-contiguous.  Slot (row, q) = 8 channels n0 + 8q .. of row t0 + row, at chunk q ^ (row & 15).
template <int FN>
__device__ __forceinline__ bool tile_epilogue_ok(const ConvP& p) {
  const bool al16 = ((uintptr_t)p.y & 15) == 0 && (!p.res || ((uintptr_t)p.res & 15) == 0);
  return FN % 2 == 0 && (p.Cout & 7) == 0 && (p.ldy & 7) == 0 && al16 && (!p.res || (p.ldr & 7) == 0) && !p.res2;
}
// phase 1: this thread's residual vectors -> image (threads tid < NT of the tile's compute waves)
template <int BM, int BN, int NT, int NRV>
__device__ __forceinline__ void tile_epi_put_res(const uint4 (&resv)[NRV], uint4* O, int tid) {
  constexpr int QPR = BN / 8;
  static_assert(NRV == BM * QPR / NT, "residual vectors per thread");
#pragma unroll
  for (int i = 0; i < NRV; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / QPR, q = idx % QPR;
    O[row * QPR + (q ^ (row & 15))] = resv[i];
  }
}

// phase 2: every lane folds its accumulators into its own 16-byte slots of the image (same arithmetic and rounding as
// conv_epilogue)
template <int FM, int FN, int ACT>
__device__ __forceinline__ void tile_epi_update(const ConvP& p, f32x4 (&acc)[FM][FN], uint4* O, int QPR, bool has_res, int b, int t0,
                                                int n0, int wm, int wn, int lane, int len, uint4* O2 = nullptr) {
  const int lr = lane & 15, lg = lane >> 4;
  const float e_scale = p.out_scale, e_rscale = p.res_scale, e_dinv = p.drop_inv_keep;
  const unsigned e_dth = p.drop_thresh16;
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int row = (wm * FM + fm) * 16 + lr;
    const int t = t0 + row;
    const bool keep = !(p.out_mask && t >= len);
#pragma unroll
    for (int h = 0; h < FN / 2; ++h) {
      const int q = wn * FN * 2 + h * 4 + lg;
      const int co = n0 + q * 8;
      f32x4 v[2] = {acc[fm][2 * h], acc[fm][2 * h + 1]};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (p.bias && co < p.Cout) v[u] += *reinterpret_cast<const f32x4*>(p.bias + co + 4 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = keep ? act_apply_c<ACT>(v[u][e], p.act) * e_scale : 0.f;
        if (e_dth) v[u] *= drop_mask4(p.drop_seed, (uint64_t)(((int64_t)b * p.T + t) * p.Cout + co + 4 * u) >> 2, e_dth, e_dinv);
      }
      uint4* slot = O + row * QPR + (q ^ (row & 15));
      if (has_res) {
        const uint4 r = *slot;
        v[0][0] += __uint_as_float(r.x << 16) * e_rscale; v[0][1] += __uint_as_float(r.x & 0xffff0000u) * e_rscale;
        v[0][2] += __uint_as_float(r.y << 16) * e_rscale; v[0][3] += __uint_as_float(r.y & 0xffff0000u) * e_rscale;
        v[1][0] += __uint_as_float(r.z << 16) * e_rscale; v[1][1] += __uint_as_float(r.z & 0xffff0000u) * e_rscale;
        v[1][2] += __uint_as_float(r.w << 16) * e_rscale; v[1][3] += __uint_as_float(r.w & 0xffff0000u) * e_rscale;
      }
      if constexpr (ACT == PTPP_ACT_GATE) {
        // fused DiffNet gate: the 8 channels are [4 "gate" | their 4 "filter" partners] (weights packed in that
        // interleaved order); y has Cout / 2 channels -- the 8-byte result takes the first half of the slot
        if (O2) {
          // training: the pre-activation is kept (rounded to bf16, second image) and the gate is computed FROM THE ROUNDED
          // values with gate_fwd_kernel's expression, so a and g are bit for bit those of the conv + gate_fwd pair
          uint4 av;
          av.x = (uint32_t)f32_to_bf16(v[0][0]) | ((uint32_t)f32_to_bf16(v[0][1]) << 16);
          av.y = (uint32_t)f32_to_bf16(v[0][2]) | ((uint32_t)f32_to_bf16(v[0][3]) << 16);
          av.z = (uint32_t)f32_to_bf16(v[1][0]) | ((uint32_t)f32_to_bf16(v[1][1]) << 16);
          av.w = (uint32_t)f32_to_bf16(v[1][2]) | ((uint32_t)f32_to_bf16(v[1][3]) << 16);
          O2[row * QPR + (q ^ (row & 15))] = av;
          const float sr[4] = {__uint_as_float(av.x << 16), __uint_as_float(av.x & 0xffff0000u), __uint_as_float(av.y << 16),
                               __uint_as_float(av.y & 0xffff0000u)};
          const float fr[4] = {__uint_as_float(av.z << 16), __uint_as_float(av.z & 0xffff0000u), __uint_as_float(av.w << 16),
                               __uint_as_float(av.w & 0xffff0000u)};
          uint2 o;
          o.x = (uint32_t)f32_to_bf16(gate_fast(sr[0], fr[0])) | ((uint32_t)f32_to_bf16(gate_fast(sr[1], fr[1])) << 16);
          o.y = (uint32_t)f32_to_bf16(gate_fast(sr[2], fr[2])) | ((uint32_t)f32_to_bf16(gate_fast(sr[3], fr[3])) << 16);
          *reinterpret_cast<uint2*>(slot) = o;
          continue;
        }
        float gte[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gte[e] = keep ? gate_fast(v[0][e], v[1][e]) : 0.f;
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(gte[0]) | ((uint32_t)f32_to_bf16(gte[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(gte[2]) | ((uint32_t)f32_to_bf16(gte[3]) << 16);
        *reinterpret_cast<uint2*>(slot) = o;
      } else {
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(v[0][0]) | ((uint32_t)f32_to_bf16(v[0][1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[0][2]) | ((uint32_t)f32_to_bf16(v[0][3]) << 16);
        o.z = (uint32_t)f32_to_bf16(v[1][0]) | ((uint32_t)f32_to_bf16(v[1][1]) << 16);
        o.w = (uint32_t)f32_to_bf16(v[1][2]) | ((uint32_t)f32_to_bf16(v[1][3]) << 16);
        *slot = o;
      }
    }
  }
}

// phase 3: the image leaves row-contiguous; NV vectors per thread of the NT threads that store
template <int BM, int BN, int NT, int NV>
__device__ __forceinline__ void tile_epi_store(const ConvP& p, const uint4* O, int b, int t0, int n0, int tid, bool gate, int i0 = 0) {
  typedef bf16_raw T;
  constexpr int QPR = BN / 8;
  T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.T * p.ldy;
  uint4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + (i0 + i) * NT;
    const int row = idx / QPR, q = idx % QPR;
    v[i] = O[row * QPR + (q ^ (row & 15))];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + (i0 + i) * NT;
    const int row = idx / QPR, q = idx % QPR;
    const int t = t0 + row, co = n0 + q * 8;
    if (t < p.T && co < p.Cout) {
      if (gate) *reinterpret_cast<uint2*>(yb + (int64_t)t * p.ldy + (co >> 1)) = make_uint2(v[i].x, v[i].y);
      else *reinterpret_cast<uint4*>(yb + (int64_t)t * p.ldy + co) = v[i];
    }
  }
}

template <int FM, int FN, int WR, int WC, int ACT>
__device__ __forceinline__ void tile_epilogue(const ConvP& p, f32x4 (&acc)[FM][FN], uint4 (&resv)[WR * FM * WC * FN * 32 / (WR * WC * 64)],
                                              uint4* O, int b, int t0, int n0, int wm, int wn, int tid, int len) {
  constexpr int NT = WR * WC * 64, BM = WR * FM * 16, BN = WC * FN * 16;
  constexpr int QPR = BN / 8;  // 16-byte slots per tile row
  constexpr int NRV = BM * QPR / NT;
  static_assert(QPR >= 16, "tile rows of at least 256 bytes");
  const bool has_res = p.res != nullptr;
  if (has_res) {
    tile_epi_put_res<BM, BN, NT, NRV>(resv, O, tid);
    __syncthreads();
  }
  uint4* O2 = (ACT == PTPP_ACT_GATE && p.gate_save) ? O + BM * QPR : nullptr;  // second image: the kept pre-activation
  tile_epi_update<FM, FN, ACT>(p, acc, O, QPR, has_res, b, t0, n0, wm, wn, tid & 63, len, O2);
  __syncthreads();
  tile_epi_store<BM, BN, NT, NRV>(p, O, b, t0, n0, tid, ACT == PTPP_ACT_GATE);
  if constexpr (ACT == PTPP_ACT_GATE) {
    if (O2) {  // slot (row, q) = channels [c0 .. c0+3 | C + c0 .. C + c0+3] of the standard layout, c0 = (n0 + 8 q) / 2
      typedef bf16_raw T;
      T* ab = reinterpret_cast<T*>(p.gate_save) + (int64_t)b * p.T * p.gate_lds;
      const int Ch = p.Cout >> 1;
#pragma unroll
      for (int i = 0; i < NRV; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / QPR, q = idx % QPR;
        const int t = t0 + row, co = n0 + q * 8;
        if (t < p.T && co < p.Cout) {
          const uint4 v = O2[row * QPR + (q ^ (row & 15))];
          *reinterpret_cast<uint2*>(ab + (int64_t)t * p.gate_lds + (co >> 1)) = make_uint2(v.x, v.y);
          *reinterpret_cast<uint2*>(ab + (int64_t)t * p.gate_lds + Ch + (co >> 1)) = make_uint2(v.z, v.w);
        }
      }
    }
  }
}

// The DiffNet layer's tail fused into its 1 x 1 output projection (modules/denoiser.py:78-83; what diffnet_post_kernel
// does as a separate pass over o = conv(g)):  channels [0, C) -> xn = (x + o) / sqrt(2), yin = xn + dnext[b];  channels
// [C, 2C) -> skip (f32) = (init ? 0 : skip) + o.  C is a multiple of the tile width, so a tile is one or the other.
// o is rounded to bf16 first, exactly as the two-kernel path stores and re-reads it: results are bit-identical to it.
__device__ __forceinline__ float round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }
template <int FM, int FN, int WR, int WC>
__device__ __forceinline__ void tile_epilogue_post(const ConvP& p, f32x4 (&acc)[FM][FN], uint4 (&resv)[FM * FN / 2],
                                                   uint4 (&resv2)[FM * FN / 2], uint4* O, int b, int t0, int n0, int wm, int wn,
                                                   int tid, int len) {
  typedef bf16_raw T;
  constexpr int NT = WR * WC * 64, BM = WR * FM * 16, BN = WC * FN * 16;
  constexpr int QPR = BN / 8;
  constexpr int NRV = BM * QPR / NT;
  const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int C = p.post_C;
  const float r2 = 0.70710678118654752f;
  if (n0 < C) {
    uint4* O2 = O + BM * QPR;  // second image: yin
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / QPR, q = idx % QPR;
      O[row * QPR + (q ^ (row & 15))] = resv[i];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < FN / 2; ++h) {
      const int q = wn * FN * 2 + h * 4 + lg;
      const int co = n0 + q * 8;
      f32x4 bias[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dn[2] = {bias[0], bias[0]};
      if (co < C) {
        if (p.bias) { bias[0] = *reinterpret_cast<const f32x4*>(p.bias + co); bias[1] = *reinterpret_cast<const f32x4*>(p.bias + co + 4); }
        if (p.post_dnext) {
          dn[0] = *reinterpret_cast<const f32x4*>(p.post_dnext + (int64_t)b * C + co);
          dn[1] = *reinterpret_cast<const f32x4*>(p.post_dnext + (int64_t)b * C + co + 4);
        }
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int row = (wm * FM + fm) * 16 + lr;
        const bool keep = !(p.out_mask && t0 + row >= len);
        uint4* slot = O + row * QPR + (q ^ (row & 15));
        const uint4 r = *slot;
        const float xv[8] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                             __uint_as_float(r.y & 0xffff0000u), __uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
                             __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
        float xn[8], yi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float o = keep ? round_bf16((acc[fm][2 * h + (e >> 2)][e & 3] + bias[e >> 2][e & 3]) * p.out_scale) : 0.f;
          xn[e] = (xv[e] + o) * r2;
          yi[e] = xn[e] + dn[e >> 2][e & 3];
        }
        uint4 o1, o2;
        o1.x = (uint32_t)f32_to_bf16(xn[0]) | ((uint32_t)f32_to_bf16(xn[1]) << 16);
        o1.y = (uint32_t)f32_to_bf16(xn[2]) | ((uint32_t)f32_to_bf16(xn[3]) << 16);
        o1.z = (uint32_t)f32_to_bf16(xn[4]) | ((uint32_t)f32_to_bf16(xn[5]) << 16);
        o1.w = (uint32_t)f32_to_bf16(xn[6]) | ((uint32_t)f32_to_bf16(xn[7]) << 16);
        o2.x = (uint32_t)f32_to_bf16(yi[0]) | ((uint32_t)f32_to_bf16(yi[1]) << 16);
        o2.y = (uint32_t)f32_to_bf16(yi[2]) | ((uint32_t)f32_to_bf16(yi[3]) << 16);
        o2.z = (uint32_t)f32_to_bf16(yi[4]) | ((uint32_t)f32_to_bf16(yi[5]) << 16);
        o2.w = (uint32_t)f32_to_bf16(yi[6]) | ((uint32_t)f32_to_bf16(yi[7]) << 16);
        *slot = o1;
        O2[row * QPR + (q ^ (row & 15))] = o2;
      }
    }
    __syncthreads();
    T* xnb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.T * C;
    T* yib = p.post_yin ? reinterpret_cast<T*>(p.post_yin) + (int64_t)b * p.T * C : nullptr;
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / QPR, q = idx % QPR;
      const int t = t0 + row, co = n0 + q * 8;
      if (t < p.T && co < C) {
        *reinterpret_cast<uint4*>(xnb + (int64_t)t * C + co) = O[row * QPR + (q ^ (row & 15))];
        if (yib) *reinterpret_cast<uint4*>(yib + (int64_t)t * C + co) = O2[row * QPR + (q ^ (row & 15))];
      }
    }
  } else {
    // f32 image of the skip rows: slot pair (2q, 2q+1) of row = 8 channels; the XOR keeps a pair adjacent
    constexpr int SPR = 2 * QPR;
    if (!p.post_init) {
#pragma unroll
      for (int i = 0; i < NRV; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / QPR, q = idx % QPR;
        const int s0 = (2 * q) ^ ((row & 15) << 1);
        O[row * SPR + s0] = resv[i];
        O[row * SPR + s0 + 1] = resv2[i];
      }
      __syncthreads();
    }
#pragma unroll
    for (int h = 0; h < FN / 2; ++h) {
      const int q = wn * FN * 2 + h * 4 + lg;
      const int co = n0 + q * 8;
      f32x4 bias[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      if (p.bias && co < p.Cout) { bias[0] = *reinterpret_cast<const f32x4*>(p.bias + co); bias[1] = *reinterpret_cast<const f32x4*>(p.bias + co + 4); }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int row = (wm * FM + fm) * 16 + lr;
        const bool keep = !(p.out_mask && t0 + row >= len);
        f32x4* slot = reinterpret_cast<f32x4*>(O + row * SPR + ((2 * q) ^ ((row & 15) << 1)));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x4 s = p.post_init ? f32x4{0.f, 0.f, 0.f, 0.f} : slot[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] = (keep ? round_bf16((acc[fm][2 * h + u][e] + bias[u][e]) * p.out_scale) : 0.f) + s[e];
          slot[u] = s;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / QPR, q = idx % QPR;
      const int t = t0 + row, c = n0 - C + q * 8;
      if (t < p.T && c < C) {
        float* dst = p.post_skip + ((int64_t)b * p.T + t) * C + c;
        const int s0 = (2 * q) ^ ((row & 15) << 1);
        *reinterpret_cast<uint4*>(dst) = O[row * SPR + s0];
        *reinterpret_cast<uint4*>(dst + 4) = O[row * SPR + s0 + 1];
      }
    }
  }
}

// The DiffNet gate backward fused into the output projection's data gradient (what gate_bwd_kernel does as a pass over
// dg = conv(do, W_out^T)): with s = a[:, c], f = a[:, C + c] (saved pre-activation), sg = sigmoid(s), th = tanh(f):
//   da[:, c] = dg * th * sg * (1 - sg),   da[:, C + c] = dg * sg * (1 - th^2).
// dg is rounded to bf16 first, as the two-kernel path stores and re-reads it.
template <int FM, int FN, int WR, int WC>
__device__ __forceinline__ void tile_epilogue_gate_bwd(const ConvP& p, f32x4 (&acc)[FM][FN], uint4 (&resv)[FM * FN / 2],
                                                       uint4 (&resv2)[FM * FN / 2], uint4* O, int b, int t0, int n0, int wm, int wn,
                                                       int tid) {
  typedef bf16_raw T;
  constexpr int NT = WR * WC * 64, BM = WR * FM * 16, BN = WC * FN * 16;
  constexpr int QPR = BN / 8;
  constexpr int NRV = BM * QPR / NT;
  const int lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  uint4* O2 = O + BM * QPR;
#pragma unroll
  for (int i = 0; i < NRV; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / QPR, q = idx % QPR;
    O[row * QPR + (q ^ (row & 15))] = resv[i];
    O2[row * QPR + (q ^ (row & 15))] = resv2[i];
  }
  __syncthreads();
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int row = (wm * FM + fm) * 16 + lr;
#pragma unroll
    for (int h = 0; h < FN / 2; ++h) {
      const int q = wn * FN * 2 + h * 4 + lg;
      const int slot = row * QPR + (q ^ (row & 15));
      const uint4 rs = O[slot], rf = O2[slot];
      const float sv[8] = {__uint_as_float(rs.x << 16), __uint_as_float(rs.x & 0xffff0000u), __uint_as_float(rs.y << 16),
                           __uint_as_float(rs.y & 0xffff0000u), __uint_as_float(rs.z << 16), __uint_as_float(rs.z & 0xffff0000u),
                           __uint_as_float(rs.w << 16), __uint_as_float(rs.w & 0xffff0000u)};
      const float fv[8] = {__uint_as_float(rf.x << 16), __uint_as_float(rf.x & 0xffff0000u), __uint_as_float(rf.y << 16),
                           __uint_as_float(rf.y & 0xffff0000u), __uint_as_float(rf.z << 16), __uint_as_float(rf.z & 0xffff0000u),
                           __uint_as_float(rf.w << 16), __uint_as_float(rf.w & 0xffff0000u)};
      float ds[8], df[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = round_bf16(acc[fm][2 * h + (e >> 2)][e & 3] * p.out_scale);
        float sg, th;
        gate_fast_parts(sv[e], fv[e], sg, th);
        ds[e] = d * th * sg * (1.f - sg);
        df[e] = d * sg * (1.f - th * th);
      }
      uint4 o1, o2;
      o1.x = (uint32_t)f32_to_bf16(ds[0]) | ((uint32_t)f32_to_bf16(ds[1]) << 16);
      o1.y = (uint32_t)f32_to_bf16(ds[2]) | ((uint32_t)f32_to_bf16(ds[3]) << 16);
      o1.z = (uint32_t)f32_to_bf16(ds[4]) | ((uint32_t)f32_to_bf16(ds[5]) << 16);
      o1.w = (uint32_t)f32_to_bf16(ds[6]) | ((uint32_t)f32_to_bf16(ds[7]) << 16);
      o2.x = (uint32_t)f32_to_bf16(df[0]) | ((uint32_t)f32_to_bf16(df[1]) << 16);
      o2.y = (uint32_t)f32_to_bf16(df[2]) | ((uint32_t)f32_to_bf16(df[3]) << 16);
      o2.z = (uint32_t)f32_to_bf16(df[4]) | ((uint32_t)f32_to_bf16(df[5]) << 16);
      o2.w = (uint32_t)f32_to_bf16(df[6]) | ((uint32_t)f32_to_bf16(df[7]) << 16);
      O[slot] = o1;
      O2[slot] = o2;
    }
  }
  __syncthreads();
  T* dab = reinterpret_cast<T*>(p.gate_da) + (int64_t)b * p.T * p.gate_ldda;
#pragma unroll
  for (int i = 0; i < NRV; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / QPR, q = idx % QPR;
    const int t = t0 + row, co = n0 + q * 8;
    if (t < p.T && co < p.Cout) {
      *reinterpret_cast<uint4*>(dab + (int64_t)t * p.gate_ldda + co) = O[row * QPR + (q ^ (row & 15))];
      *reinterpret_cast<uint4*>(dab + (int64_t)t * p.gate_ldda + p.Cout + co) = O2[row * QPR + (q ^ (row & 15))];
    }
  }
}

__device__ unsigned long long* g_conv_stamps;  // experiments only (DBG = true): 4 clock stamps per block

// POST: the instantiation with the fused DiffNet tail (ptpp_conv1d_diffnet_post) -- separate, because its second set
// of prefetch registers and epilogue pushed the plain 128 x 128 kernel into scratch memory (576 B per lane; BigVGAN
// C = 128 k = 7 went from 707 to 1334 us).
// EPI: 0 = the ordinary epilogue, 1 = POST, 2 = the fused DiffNet gate backward (ptpp_conv1d_gate_bwd), 3 = split-K: block
// (b, mt, nt, split) walks its share of the Cin chunks and stores raw f32 partial sums for conv_splitk_finish_kernel (few
// output tiles, long K: the Conformer feed-forward k = 9 convs at phone level, 1024 -> 256 over ~150 rows per utterance)
template <int FM, int FN, int WR, int WC, int D, bool DBG = false, int EPI = 0>
__global__ __launch_bounds__(WR* WC * 64) void conv1d_glds_kernel(const ConvP p) {
  constexpr bool POST = EPI == 1;
  constexpr bool GBWD = EPI == 2;
  constexpr bool SK = EPI == 3;
  typedef bf16_raw T;
  constexpr int NW = WR * WC;
  constexpr int BM = WR * FM * 16, BN = WC * FN * 16;
  constexpr int LW = BN / 8 / NW;  // 1 KiB pieces (8 rows x 128 B) of a weight stage per wave
  static_assert(LW * NW * 8 == BN, "weight pieces must divide over the waves");
  constexpr int WSTAGE = BN * 8;  // uint4 per weight stage

  extern __shared__ __attribute__((aligned(16))) char smem[];  // the ONLY LDS object
  unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
  if constexpr (DBG) stamp0 = wall_clock64();
  const int xrows = (BM + (p.ks - 1) * p.dil + 7) & ~7;
  uint4* Ws = reinterpret_cast<uint4*>(smem);  // [D][BN][8]
  uint4* Xs = Ws + D * WSTAGE;                 // [2][xrows][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WC, wn = wave % WC;
  const int lr = lane & 15, lg = lane >> 4;

  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = lid % p.nNT;
  const int mt = (lid / p.nNT) % p.nMT;
  const int ball = lid / (p.nNT * p.nMT);
  const int b = SK ? ball % p.B : ball;
  const int split = SK ? ball / p.B : 0;
  const int t0 = mt * BM, n0 = nt * BN;

  const int len_raw = p.lengths ? p.lengths[b] : p.T;  // requested here, consumed after the first weight stage is on its way
  int len = p.T, Tin = p.T;
  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.T * p.ldx;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nC = p.cinp >> 6;
  int sbeg = 0, steps = nC * p.ks;  // the block's K steps are [sbeg, steps); a fully masked tile gets none (below)
  if constexpr (SK) {  // whole chunks per split, so a split starts at tap 0 of a chunk
    sbeg = (int)((int64_t)split * nC / p.nsplit) * p.ks;
    steps = (int)((int64_t)(split + 1) * nC / p.nsplit) * p.ks;
  }

  // per-lane source of the wave's weight pieces at (ci = 0, tap 0)
  const char* wsrc[LW];
#pragma unroll
  for (int q = 0; q < LW; ++q) {
    const int row = (wave * LW + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz<8>(row);
    const int n = min(n0 + wperm_inv<FN>(row), p.Cout - 1);  // rows past Cout: any finite data, never stored
    wsrc[q] = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.wp) + (int64_t)n * p.ks * p.cinp + c * 8);
  }
  const uint32_t ws_lds = lds_addr(Ws) + (uint32_t)wave * LW * 1024u;
  const uint32_t xs_lds = lds_addr(Xs);
  const char* zero = reinterpret_cast<const char*>(g_conv_zero_page) + lane * 16;

  // Weight stages are issued strictly in step order, so the issue side keeps its own (chunk, tap) pair incrementally: the
  // loop used to divide by ks twice per step (once here, once for the step it multiplies) -- ~60 of the ~120 scalar
  // instructions a step spent beside its 16 MFMAs (profiles/r05_kloop_instruction_mix.txt)
  int wci = sbeg / p.ks, wj = 0;  // (sbeg is a multiple of ks)
  auto issue_w = [&](int stage) {
    const int off = (wj * p.cinp + wci * 64) * 2;
#pragma unroll
    for (int q = 0; q < LW; ++q)
      glds16(wsrc[q] + off, __builtin_amdgcn_readfirstlane(ws_lds + (uint32_t)(stage * WSTAGE * 16 + q * 1024)));
    if (++wj == p.ks) {
      wj = 0;
      ++wci;
    }
  };
  auto issue_x = [&](int ci, int buf) {
    const int np = xrows >> 3;
    for (int piece = wave; piece < np; piece += NW) {
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ swz<8>(r);
      const int ts = t0 - p.pad + r;
      const char* src = (ts >= 0 && ts < Tin) ? reinterpret_cast<const char*>(xb + (int64_t)ts * p.ldx + ci * 64 + c * 8) : zero;
      glds16(src, __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)((buf * xrows + piece * 8) * 128)));
    }
  };

  // the residual rows of the tile, fetched row-contiguous (see tile_epilogue) while the first operands travel:
  // the K loop's first vmcnt(0) retires them together with the first stage
  constexpr bool post = POST;
  const bool tile_epi = !SK && (post || GBWD || tile_epilogue_ok<FN>(p));
  constexpr int NRV = BM * BN / 8 / (NW * 64);  // 16-byte vectors of the output tile per thread
  uint4 resv[NRV], resv2[(POST || GBWD) ? NRV : 1];
  if constexpr (GBWD) {  // the saved pre-activation rows: gate half [c] and filter half [C + c] of this tile's channels
    const T* ab = reinterpret_cast<const T*>(p.gate_a) + (int64_t)b * p.T * 2 * p.Cout;
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NW * 64;
      const int row = idx / (BN / 8), q = idx % (BN / 8);
      const int t = t0 + row, co = n0 + q * 8;
      const bool in = t < p.T && co < p.Cout;
      resv[i] = in ? *reinterpret_cast<const uint4*>(ab + (int64_t)t * 2 * p.Cout + co) : make_uint4(0, 0, 0, 0);
      resv2[i] = in ? *reinterpret_cast<const uint4*>(ab + (int64_t)t * 2 * p.Cout + p.Cout + co) : make_uint4(0, 0, 0, 0);
    }
  } else
  if (post && n0 >= p.post_C) {  // "skip" half of the DiffNet output projection: the f32 skip rows, two vectors per slot
    if constexpr (POST) {
      if (!p.post_init) {
#pragma unroll
        for (int i = 0; i < NRV; ++i) {
          const int idx = tid + i * NW * 64;
          const int row = idx / (BN / 8), q = idx % (BN / 8);
          const int t = t0 + row, c = n0 - p.post_C + q * 8;
          const float* src = p.post_skip + ((int64_t)b * p.T + t) * p.post_C + c;
          const bool in = t < p.T && c < p.post_C;
          resv[i] = in ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
          resv2[i] = in ? *reinterpret_cast<const uint4*>(src + 4) : make_uint4(0, 0, 0, 0);
        }
      }
    }
  } else if (tile_epi && p.res) {
    const T* rb = reinterpret_cast<const T*>(p.res) + (int64_t)b * p.T * p.ldr;
#pragma unroll
    for (int i = 0; i < NRV; ++i) {
      const int idx = tid + i * NW * 64;
      const int row = idx / (BN / 8), q = idx % (BN / 8);
      const int t = t0 + row, co = n0 + q * 8;
      resv[i] = (t < p.T && co < (post ? p.post_C : p.Cout)) ? *reinterpret_cast<const uint4*>(rb + (int64_t)t * p.ldr + co)
                                                             : make_uint4(0, 0, 0, 0);
    }
  }
  // The first weight stage does not depend on the utterance length: it leaves before the scalar load of lengths[b] is waited
  // for (that wait used to sit in front of every load of the block).
  const bool any = steps > sbeg;
  if (any) issue_w(0);
  len = min(len_raw, p.T);
  Tin = p.in_mask ? len : p.T;
  // No K loop for a tile whose output rows are all masked out, or whose whole input window lies past the utterance's end
  // with a masked input (every operand row is zero: the accumulators stay exactly zero) -- token-bucket batches are padded
  // to their longest utterance, a third of the row tiles of a frame-level launch.  (The stage in flight is drained.)
  if (!SK && ((p.out_mask && t0 >= len) || (p.in_mask && t0 - p.pad >= len))) steps = sbeg;
  if (steps > sbeg) {
    issue_x(sbeg / p.ks, (sbeg / p.ks) & 1);
    if (D > 2 && steps > sbeg + 1) issue_w(1);
  } else if (any) {
    glds_wait<0>();
  }
  int stage = 0;
  int ci = sbeg / p.ks, j = 0;  // chunk and tap of the step being multiplied
  for (int s = sbeg; s < steps; ++s) {
    if (D > 2 && s + 1 < steps) glds_wait<LW>();
    else glds_wait<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG) {
      if (s == 0) stamp1 = wall_clock64();
    }
    if (j == 0 && (ci + 1) * p.ks < steps) issue_x(ci + 1, (ci + 1) & 1);
    if (s + D - 1 < steps) issue_w(stage == 0 ? D - 1 : stage - 1);

    const uint4* Wb = Ws + stage * WSTAGE;
    const uint4* Xb = Xs + (ci & 1) * xrows * 8;
    const int rsh = j * p.dil;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + lg;
      uint4 wf[FN], xf[FM];
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = (wn * FN + fn) * 16 + lr;
        wf[fn] = Wb[n * 8 + (c ^ swz<8>(n))];
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int r = (wm * FM + fm) * 16 + lr + rsh;
        xf[fm] = Xb[r * 8 + (c ^ swz<8>(r))];
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
          acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[fn]),
                                                                __builtin_bit_cast(bf16x8_t, xf[fm]), acc[fm][fn], 0, 0, 0);
    }
    stage = stage + 1 == D ? 0 : stage + 1;
    if (++j == p.ks) {
      j = 0;
      ++ci;
    }
  }
  if constexpr (DBG) stamp2 = wall_clock64();
  if constexpr (SK) {
    float* wsb = p.ws + ((int64_t)split * p.B + b) * p.T * p.Cout;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const int t = t0 + (wm * FM + fm) * 16 + lr;
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int co = n0 + wn * FN * 16 + (fn >> 1) * 32 + lg * 8 + (fn & 1) * 4;  // (the channel map of conv_epilogue / wperm)
        if (t < p.T && co < p.Cout) *reinterpret_cast<f32x4*>(wsb + (int64_t)t * p.Cout + co) = acc[fm][fn];
      }
    }
    return;
  }
  if (tile_epi) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done with the operand stages: their memory becomes the tile
    if constexpr (POST) {
      tile_epilogue_post<FM, FN, WR, WC>(p, acc, resv, resv2, reinterpret_cast<uint4*>(smem), b, t0, n0, wm, wn, tid, len);
    } else if constexpr (GBWD) {
      tile_epilogue_gate_bwd<FM, FN, WR, WC>(p, acc, resv, resv2, reinterpret_cast<uint4*>(smem), b, t0, n0, wm, wn, tid);
    } else {
      act_dispatch(p.act, [&](auto tag) __attribute__((always_inline)) {
        tile_epilogue<FM, FN, WR, WC, decltype(tag)::value>(p, acc, resv, reinterpret_cast<uint4*>(smem), b, t0, n0, wm, wn, tid, len);
      });
    }
  } else {
    conv_epilogue<T, FM, FN>(p, acc, b, t0, n0, wm, wn, lane, len);
  }
  if constexpr (DBG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      unsigned long long* o = g_conv_stamps + (size_t)blockIdx.x * 4;
      o[0] = stamp0; o[1] = stamp1; o[2] = stamp2; o[3] = wall_clock64();
    }
  }
}

// bf16, whole 64-channel K chunks and 16-byte aligned rows (the caller has checked alignment of x)
inline bool glds_ok(const ConvP& p) { return p.cinp == p.Cin && (p.Cin & 63) == 0 && p.Cout >= 64; }

template <int FM, int FN, int WR, int WC, int D>
int launch_glds(ConvP& p, hipStream_t st) {
  constexpr int BM = WR * FM * 16, BN = WC * FN * 16;
  p.nMT = (p.T + BM - 1) / BM;
  p.nNT = (p.Cout + BN - 1) / BN;
  const int xrows = (BM + (p.ks - 1) * p.dil + 7) & ~7;
  const size_t smem = (size_t)(D * BN * 8 + 2 * xrows * 8) * 16;
  if (smem > 160 * 1024) return -1;
  auto kern = p.post_skip ? conv1d_glds_kernel<FM, FN, WR, WC, D, false, 1>
              : p.gate_a  ? conv1d_glds_kernel<FM, FN, WR, WC, D, false, 2>
                          : conv1d_glds_kernel<FM, FN, WR, WC, D, false, 0>;
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_fwd (lds-dma)")) return PTPP_ELAUNCH;
  const int64_t nblk = (int64_t)p.B * p.nMT * p.nNT;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(WR * WC * 64), smem, st, p);
  PTPP_CHECK_LAUNCH("conv1d_fwd (lds-dma)");
  return PTPP_OK;
}

// split-K launch of the LDS-DMA kernel + conv_splitk_finish_kernel (declared in conv1d_cl.hip): few output tiles, long K
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvP p);
template <int FM, int FN, int WR, int WC, int D>
bool launch_glds_splitk(ConvP& p, hipStream_t st, void* ws, size_t ws_bytes, int* status) {
  constexpr int BM = WR * FM * 16, BN = WC * FN * 16;
  static_assert(FN % 2 == 0, "the split-K store uses the paired-fragment channel map");
  const int nC = p.cinp >> 6;
  const int nMT = (p.T + BM - 1) / BM, nNT = (p.Cout + BN - 1) / BN;
  const int64_t blocks = (int64_t)p.B * nMT * nNT;
  // (thresholds 384 / 768 vs 512 / 1024 / 512: 18.29-18.37 ms per step, all within run-to-run noise)
  if (!ws || !glds_ok(p) || (p.Cout & 3) || p.act == PTPP_ACT_GATE || p.post_skip || p.gate_a || (int64_t)nC * p.ks < 32 || blocks >= 384)
    return false;
  int64_t ns = (768 + blocks - 1) / blocks;  // three resident blocks per CU
  if (ns > 8) ns = 8;
  if (ns > nC) ns = nC;
  const int64_t slab = (int64_t)p.B * p.T * p.Cout * (int64_t)sizeof(float);
  if (ns * slab > (int64_t)ws_bytes) ns = (int64_t)ws_bytes / slab;
  if (ns < 2) return false;
  const int xrows = (BM + (p.ks - 1) * p.dil + 7) & ~7;
  const size_t smem = (size_t)(D * BN * 8 + 2 * xrows * 8) * 16;
  if (smem > 160 * 1024) return false;
  p.ws = reinterpret_cast<float*>(ws);
  p.nsplit = (int)ns;
  p.nMT = nMT;
  p.nNT = nNT;
  auto kern = conv1d_glds_kernel<FM, FN, WR, WC, D, false, 3>;
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_fwd (lds-dma, split-K)")) {
    *status = PTPP_ELAUNCH;
    return true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(blocks * ns)), dim3(WR * WC * 64), smem, st, p);
  const int64_t nvec = (int64_t)p.B * p.T * (p.Cout >> 2);
  int64_t fb = (nvec + 255) / 256;
  if (fb > 4096) fb = 4096;
  hipLaunchKernelGGL(conv_splitk_finish_kernel<bf16_raw>, dim3((unsigned)fb), dim3(256), 0, st, p);
  *status = hipGetLastError() == hipSuccess ? PTPP_OK : PTPP_ELAUNCH;
  if (*status != PTPP_OK) ptpp_set_error("conv1d_fwd (LDS-DMA split-K): launch failed");
  return true;
}

}  // namespace
