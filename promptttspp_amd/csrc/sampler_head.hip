// Everything between two DiffNet stacks of the reverse-diffusion loop, as ONE launch (bf16, 256 residual channels, mel
// dimension a multiple of 16 up to 96):
//   h     = relu(W_s s + b_s)                    skip projection          modules/denoiser.py:147-150
//   eps   = W_o h + b_o                          output projection        modules/denoiser.py:151-152
//   x'    = mean(x, clamp(x0(x, eps))) + sigma_t * noise                  modules/diffusion.py:283-302  (ptpp_ddpm_step)
//   h0'   = relu(W_in bf16(x') + b_in)           input projection of the NEXT step   modules/denoiser.py:131
//   yin0' = h0' + dstep'[b]                      + its first layer's step projection modules/denoiser.py:76
// Every row is independent (1 x 1 convolutions and elementwise maths): the sampler issued seven launches for it per step (52 us
// back to back at 32 x 546 frames); this launch takes 40 us there (26 us at 270 rows: a block's life is a chain of ~8 L2 round
// trips, which prefetching a whole pass's operands did not shorten -- it cost the second resident block instead).  Same operand
// order and the same rounding points as the launches it replaces (h, eps, bf16(x'), h0' are rounded to bf16 exactly where those
// stored them); the conv kernels feed the MFMA's K slots in another order, so isolated eps elements (~1 in 10^4) round to the
// other bf16 neighbour.
//
// A block owns 64 rows; the three GEMMs run on the MFMAs as out^T = W x act^T: weights are the A operand straight from the
// packed (mode 0, K-contiguous) operands in L2, activations the B operand from LDS tiles (rows of 512 B with the 16-byte
// chunk index xor-ed by row & 15; the 80-channel x tile in rows of 208 B: both conflict-free for 16 rows x one chunk).
#include "ptpp_common.h"
#include "../../include/ptpp.h"

namespace {

constexpr int SH_C = 256;
constexpr int SH_BM = 64;

struct ShP {
  const bf16_raw* s;
  const bf16_raw *ws, *wo, *win;
  const float *bs, *bo, *bin;
  const float* x;
  const float* noise;
  const long long* t;
  const float *sra, *srm1, *c1, *c2, *logvar;
  float* x_out;
  const float* ds0;
  bf16_raw *h0, *yin0;
  int64_t rows;
  int T, M, MP;  // MP = M padded to a multiple of 32 (the packed input-projection operand's row length)
};

// F16 (round 6): IEEE-half storage and matrix instruction instead of bf16 (BASELINE config 5: "fp16 mel decoder"); the pointers of
// ShP address 2-byte elements either way
typedef __attribute__((ext_vector_type(8))) _Float16 sh_f16x8;
template <bool F16>
__device__ __forceinline__ uint32_t sh_pack2_(float a, float b) {
  if constexpr (F16) return H2<f16_raw>::pack(a, b);
  else return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
template <bool F16>
__device__ __forceinline__ float sh_round_(float v) {
  if constexpr (F16) return (float)(_Float16)v;
  else return bf16_to_f32(f32_to_bf16(v));
}
template <bool F16>
__device__ __forceinline__ f32x4 sh_mfma_(uint4 a, uint4 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sh_f16x8, a), __builtin_bit_cast(sh_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// (the f16 form holds ~60 more registers than the bf16 one -- the conversions do not fold into integer shifts -- and at M = 96 no
// longer fits two blocks per CU without spilling: that one instantiation runs one block per CU; config 5's M = 80 is unaffected)
template <int NT, bool F16 = false>  // n-tiles of 16 of the mel dimension (M = 16 NT)
__global__ __launch_bounds__(256, (F16 && NT == 6) ? 1 : 2) void sampler_head_kernel(const ShP p) {
  auto sh_pack2 = [](float a, float b) __attribute__((always_inline)) { return sh_pack2_<F16>(a, b); };
  auto sh_mfma = [](uint4 a, uint4 b, f32x4 c) __attribute__((always_inline)) { return sh_mfma_<F16>(a, b, c); };
  constexpr int XCH = 2 * NT + ((2 * NT) % 4 ? 4 - (2 * NT) % 4 : 0);  // 16-byte chunks of a padded x row (MP / 8)
  constexpr int XST = XCH + 1;                                         // row stride of the x tile in chunks (odd multiple: no conflicts)
  __shared__ uint4 S[SH_BM * 32];   // s, then h: [row][chunk ^ (row & 15)]
  __shared__ uint4 XS[SH_BM * XST];  // bf16(x') rows, zero padded to MP channels
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * SH_BM;

  // ---- s tile -> LDS; the x tile's padding zeroed; ALL weight fragments of the first GEMM requested meanwhile (the kernel is a
  // chain of L2 round trips: each pass's operands are requested a phase ahead, 32 + 8 NT + 12 fragments per lane in all) ----
  uint4 a1[4][4];  // (half a pass at a time: two resident blocks per CU need <= 128 registers)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) a1[ks][fn] = *reinterpret_cast<const uint4*>(p.ws + (int64_t)(w * 64 + fn * 16 + lq) * SH_C + ks * 32 + lg * 8);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 5, c = idx & 31;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + r < p.rows) v = *reinterpret_cast<const uint4*>(p.s + (row0 + r) * SH_C + c * 8);
    S[r * 32 + (c ^ (r & 15))] = v;
  }
  for (int idx = tid; idx < SH_BM * XST; idx += 256) XS[idx] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- GEMM 1: h = relu(W_s s + b_s); wave w owns output channels 64 w .. 64 w + 63, all 64 rows ----
  f32x4 acc[4][4];
#pragma unroll
  for (int fm = 0; fm < 4; ++fm)
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    uint4 cur[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) cur[ks][fn] = a1[ks][fn];
    if (half == 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
          a1[ks][fn] = *reinterpret_cast<const uint4*>(p.ws + (int64_t)(w * 64 + fn * 16 + lq) * SH_C + (4 + ks) * 32 + lg * 8);
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int ks = half * 4 + k4;
      uint4 b[4];
#pragma unroll
      for (int fm = 0; fm < 4; ++fm) {
        const int r = fm * 16 + lq;
        b[fm] = S[r * 32 + ((ks * 4 + lg) ^ (r & 15))];
      }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = sh_mfma(cur[k4][fn], b[fm], acc[fm][fn]);
    }
  }
  // the first half of the second GEMM's weights (all rows of W_o) is on its way while h is formed
  uint4 a2[4][NT];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a2[ks][nt] = *reinterpret_cast<const uint4*>(p.wo + (int64_t)(nt * 16 + lq) * SH_C + ks * 32 + lg * 8);
  __syncthreads();  // every wave is done reading s
  {
    uint2* S2 = reinterpret_cast<uint2*>(S);
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      const int n = w * 64 + fn * 16 + 4 * lg;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bs + n);
#pragma unroll
      for (int fm = 0; fm < 4; ++fm) {
        const int r = fm * 16 + lq;
        f32x4 v = acc[fm][fn] + bias;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        S2[(r * 32 + ((n >> 3) ^ (r & 15))) * 2 + ((n >> 2) & 1)] = make_uint2(sh_pack2(v[0], v[1]), sh_pack2(v[2], v[3]));
      }
    }
  }
  __syncthreads();

  // ---- GEMM 2: eps = W_o h + b_o for the wave's 16 rows, then the reverse-diffusion update ----
  uint4 a3[XCH / 4][4];  // the third GEMM's weights: requested before the update's own loads
  if (p.win) {
#pragma unroll
    for (int ks = 0; ks < XCH / 4; ++ks)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) a3[ks][fn] = *reinterpret_cast<const uint4*>(p.win + (int64_t)(w * 64 + fn * 16 + lq) * p.MP + ks * 32 + lg * 8);
  }
  {
    f32x4 e2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) e2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r = w * 16 + lq;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 cur[4][NT];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) cur[ks][nt] = a2[ks][nt];
      if (half == 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) a2[ks][nt] = *reinterpret_cast<const uint4*>(p.wo + (int64_t)(nt * 16 + lq) * SH_C + (4 + ks) * 32 + lg * 8);
      }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const uint4 b = S[r * 32 + (((half * 4 + k4) * 4 + lg) ^ (r & 15))];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) e2[nt] = sh_mfma(cur[k4][nt], b, e2[nt]);
      }
    }
    const int64_t grow = row0 + r;
    if (grow < p.rows) {
      const long long tb = p.t[grow / p.T];
      const float ca = p.sra[tb], cb = p.srm1[tb], k1 = p.c1[tb], k2 = p.c2[tb], sg = expf(__fmul_rn(0.5f, p.logvar[tb]));
      uint2* X2 = reinterpret_cast<uint2*>(XS);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + 4 * lg;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bo + n);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + grow * p.M + n);
        f32x4 nv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.noise) nv = *reinterpret_cast<const f32x4*>(p.noise + grow * p.M + n);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ev = sh_round_<F16>(e2[nt][e] + bias[e]);  // eps as the output projection stores it
          o[e] = ddpm_update(ca, cb, k1, k2, sg, xv[e], ev, nv[e]);
        }
        *reinterpret_cast<f32x4*>(p.x_out + grow * p.M + n) = o;
        X2[(r * XST + (n >> 3)) * 2 + ((n >> 2) & 1)] = make_uint2(sh_pack2(o[0], o[1]), sh_pack2(o[2], o[3]));
      }
    }
  }
  if (!p.win) return;
  __syncthreads();

  // ---- GEMM 3: h0' = relu(W_in bf16(x') + b_in), yin0' = h0' + dstep'[b]; wave w owns channels 64 w .. 64 w + 63 ----
#pragma unroll
  for (int fm = 0; fm < 4; ++fm)
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < XCH / 4; ++ks) {  // (MP / 32 steps)
    uint4 b[4];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) b[fm] = XS[(fm * 16 + lq) * XST + ks * 4 + lg];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) acc[fm][fn] = sh_mfma(a3[ks][fn], b[fm], acc[fm][fn]);
  }
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int64_t grow = row0 + fm * 16 + lq;
    if (grow >= p.rows) continue;
    const float* dsb = p.ds0 + (grow / p.T) * SH_C;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      const int n = w * 64 + fn * 16 + 4 * lg;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bin + n), dsv = *reinterpret_cast<const f32x4*>(dsb + n);
      f32x4 v = acc[fm][fn] + bias;
      float hr[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hr[e] = sh_round_<F16>(fmaxf(v[e], 0.f));
      *reinterpret_cast<uint2*>(p.h0 + grow * SH_C + n) = make_uint2(sh_pack2(hr[0], hr[1]), sh_pack2(hr[2], hr[3]));
      *reinterpret_cast<uint2*>(p.yin0 + grow * SH_C + n) =
          make_uint2(sh_pack2(hr[0] + dsv[0], hr[1] + dsv[1]), sh_pack2(hr[2] + dsv[2], hr[3] + dsv[3]));
    }
  }
}

}  // namespace

extern "C" int ptpp_sampler_head_supported(int C, int M, int dtype) {
  return (dtype == PTPP_BF16 || dtype == PTPP_F16) && C == SH_C && M > 0 && M % 16 == 0 && M <= 96;  // (wider mels: the W_o prefetch spills)
}

extern "C" int ptpp_sampler_head(const ptpp_sampler_head_args* a, void* stream) {
  PTPP_CHECK_ARG(a && a->s && a->ws_p && a->ws_b && a->wo_p && a->wo_b && a->x && a->t && a->sra && a->srm1 && a->c1 && a->c2 &&
                     a->logvar && a->x_out,
                 "sampler_head: null pointer");
  PTPP_CHECK_ARG(ptpp_sampler_head_supported(a->C, a->M, a->dtype), "sampler_head: bf16 / f16, C = 256, M %% 16 == 0, M <= 96 (C %d M %d dtype %d)",
                 a->C, a->M, a->dtype);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0, "sampler_head: bad shape");
  PTPP_CHECK_ARG(!a->win_p || (a->win_b && a->ds0 && a->h0 && a->yin0), "sampler_head: the next step's outputs need win_b, ds0, h0, yin0");
  const uintptr_t al = (uintptr_t)a->s | (uintptr_t)a->ws_p | (uintptr_t)a->wo_p | (uintptr_t)a->win_p | (uintptr_t)a->x | (uintptr_t)a->noise |
                       (uintptr_t)a->x_out | (uintptr_t)a->h0 | (uintptr_t)a->yin0 | (uintptr_t)a->ws_b | (uintptr_t)a->wo_b |
                       (uintptr_t)a->win_b | (uintptr_t)a->ds0;
  PTPP_CHECK_ARG((al & 15) == 0, "sampler_head: every tensor must be 16-byte aligned");
  ShP p;
  p.s = reinterpret_cast<const bf16_raw*>(a->s);
  p.ws = reinterpret_cast<const bf16_raw*>(a->ws_p); p.wo = reinterpret_cast<const bf16_raw*>(a->wo_p);
  p.win = reinterpret_cast<const bf16_raw*>(a->win_p);
  p.bs = a->ws_b; p.bo = a->wo_b; p.bin = a->win_b;
  p.x = a->x; p.noise = a->noise; p.t = reinterpret_cast<const long long*>(a->t);
  p.sra = a->sra; p.srm1 = a->srm1; p.c1 = a->c1; p.c2 = a->c2; p.logvar = a->logvar;
  p.x_out = a->x_out; p.ds0 = a->ds0;
  p.h0 = reinterpret_cast<bf16_raw*>(a->h0); p.yin0 = reinterpret_cast<bf16_raw*>(a->yin0);
  p.rows = (int64_t)a->B * a->T; p.T = a->T; p.M = a->M; p.MP = (a->M + 31) & ~31;
  const dim3 grid((unsigned)((p.rows + SH_BM - 1) / SH_BM)), blk(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == PTPP_F16) {
    switch (a->M / 16) {
      case 1: hipLaunchKernelGGL((sampler_head_kernel<1, true>), grid, blk, 0, st, p); break;
      case 2: hipLaunchKernelGGL((sampler_head_kernel<2, true>), grid, blk, 0, st, p); break;
      case 3: hipLaunchKernelGGL((sampler_head_kernel<3, true>), grid, blk, 0, st, p); break;
      case 4: hipLaunchKernelGGL((sampler_head_kernel<4, true>), grid, blk, 0, st, p); break;
      case 5: hipLaunchKernelGGL((sampler_head_kernel<5, true>), grid, blk, 0, st, p); break;
      default: hipLaunchKernelGGL((sampler_head_kernel<6, true>), grid, blk, 0, st, p); break;
    }
    PTPP_CHECK_LAUNCH("sampler_head");
    return PTPP_OK;
  }
  switch (a->M / 16) {
    case 1: hipLaunchKernelGGL(sampler_head_kernel<1>, grid, blk, 0, st, p); break;
    case 2: hipLaunchKernelGGL(sampler_head_kernel<2>, grid, blk, 0, st, p); break;
    case 3: hipLaunchKernelGGL(sampler_head_kernel<3>, grid, blk, 0, st, p); break;
    case 4: hipLaunchKernelGGL(sampler_head_kernel<4>, grid, blk, 0, st, p); break;
    case 5: hipLaunchKernelGGL(sampler_head_kernel<5>, grid, blk, 0, st, p); break;
    default: hipLaunchKernelGGL(sampler_head_kernel<6>, grid, blk, 0, st, p); break;
  }
  PTPP_CHECK_LAUNCH("sampler_head");
  return PTPP_OK;
}
