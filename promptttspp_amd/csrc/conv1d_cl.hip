// Channels-last Conv1d / Linear as an MFMA implicit GEMM for gfx950.
//
//   y[b,t,co] = res_scale*res[b,t,co] + res2[b,t,co]
//             + out_scale * mask_out(t) * act(bias[co] + sum_{j,ci} W[co,ci,j] * xm[b, t + j*dil - pad, ci])
//
// GEMM view per utterance b: M = T rows (time), N = Cout, K = ks*Cin.  The
// activations are channels-last, so for a fixed tap j the A operand of an
// M-tile is a contiguous row window of x shifted by j*dil - pad: the block
// stages ONE window of BM + (ks-1)*dil rows per Cin-chunk into LDS and all ks
// taps read it at shifted row offsets (ks-fold reuse of every staged byte).
// Weights are pre-packed [Cout][ks][CinP] (K-contiguous) by pack.hip.
//
// Tiles never span utterances (grid = B x ceil(T/BM) x ceil(Cout/BN)), which
// makes zero padding at sequence edges a pure row-range test and lets tiles
// that lie entirely in an utterance's padding skip the K loop.
//
// MFMA: the operands are consumed in 16-byte K-chunks, so the same staging /
// fragment code serves bf16 (one v_mfma_f32_16x16x32_bf16 per chunk pair) and
// exact f32 (four v_mfma_f32_16x16x4_f32).  The weight fragment is passed as
// the MFMA "A" operand and the activation fragment as "B", so each lane ends
// up with 4 CONSECUTIVE output channels of one output row: bias loads and the
// stores are 8/16-byte vectors.  For bf16 the weight rows of a wave's N span are
// additionally permuted when they are copied to LDS (wperm) so that a lane's
// fragments (2h, 2h+1) hold 8 consecutive channels: every store / residual load is a
// full 16-byte vector and the four lane groups of a row cover 64 contiguous bytes
// (the unpermuted layout wrote 8-byte pieces, 32 contiguous bytes per row).
#include "ptpp_common.h"

namespace {

template <typename T>
struct Mma;
template <>
struct Mma<float> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    const f32x4 af = __builtin_bit_cast(f32x4, a);
    const f32x4 bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc, 0, 0, 0);
  }
};
template <>
struct Mma<bf16_raw> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};

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
};

template <typename T, int NCH, int WM, int FM, int FN, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv1d_cl_kernel(const ConvP p) {
  constexpr int WN = NW / WM;
  constexpr int NT = NW * 64;  // threads per block
  constexpr int BM = WM * FM * 16;
  constexpr int BN = WN * FN * 16;
  constexpr int KC = 16 / (int)sizeof(T);
  constexpr int BKE = NCH * KC;
  constexpr int WCH = BN * NCH;                 // 16-B chunks in a W tile
  constexpr int WREG = (WCH + NT - 1) / NT;     // chunks per thread

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int BMW = BM + (p.ks - 1) * p.dil;
  uint4* Ws = reinterpret_cast<uint4*>(smem);   // [2][BN][NCH]
  uint4* Xs = Ws + 2 * WCH;                     // [2][BMW][NCH]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lg = lane >> 4;

  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = lid % p.nNT;
  const int mt = (lid / p.nNT) % p.nMT;
  const int b = lid / (p.nNT * p.nMT);
  const int t0 = mt * BM, n0 = nt * BN;

  const int len = p.lengths ? min(p.lengths[b], p.T) : p.T;
  const int Tin = p.in_mask ? len : p.T;

  const T* xb = reinterpret_cast<const T*>(p.x) + (int64_t)b * p.T * p.ldx;
  const T* wp = reinterpret_cast<const T*>(p.wp);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nC = p.cinp / BKE;
  // a tile whose rows are all masked out contributes nothing: skip the K loop
  const int steps = (p.out_mask && t0 >= len) ? 0 : nC * p.ks;

  uint4 wreg[WREG];
  auto load_w = [&](int s) {
    const int ci = s / p.ks, j = s - ci * p.ks;
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int idx = tid + i * NT;
      const int n = idx / NCH, c = idx % NCH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (idx < WCH && n0 + n < p.Cout)
        v = *reinterpret_cast<const uint4*>(wp + ((int64_t)(n0 + n) * p.ks + j) * p.cinp + ci * BKE + c * KC);
      wreg[i] = v;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int idx = tid + i * NT;
      const int n = idx / NCH, c = idx % NCH;
      const int q = wperm<T, FN>(n);
      if (idx < WCH) Ws[buf * WCH + q * NCH + (c ^ swz<NCH>(q))] = wreg[i];
    }
  };
  auto stage_x = [&](int ci, int buf) {
    uint4* dst = Xs + buf * BMW * NCH;
    for (int idx = tid; idx < BMW * NCH; idx += NT) {
      const int r = idx / NCH, c = idx % NCH;
      const int ts = t0 - p.pad + r;
      const int ch = ci * BKE + c * KC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ts >= 0 && ts < Tin && ch < p.Cin)
        v = *reinterpret_cast<const uint4*>(xb + (int64_t)ts * p.ldx + ch);
      dst[r * NCH + (c ^ swz<NCH>(r))] = v;
    }
  };

  if (steps > 0) load_w(0);
  int wbuf = 0, xbuf = 1;
  for (int s = 0; s < steps; ++s) {
    const int ci = s / p.ks, j = s - ci * p.ks;
    if (j == 0) {
      xbuf ^= 1;
      stage_x(ci, xbuf);
    }
    store_w(wbuf);
    __syncthreads();
    if (s + 1 < steps) load_w(s + 1);

    const uint4* Wb = Ws + wbuf * WCH;
    const uint4* Xb = Xs + xbuf * BMW * NCH;
    const int rsh = j * p.dil;
#pragma unroll
    for (int kk = 0; kk < NCH / 4; ++kk) {
      const int c = kk * 4 + lg;
      uint4 wf[FN], xf[FM];
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = (wn * FN + fn) * 16 + lr;
        wf[fn] = Wb[n * NCH + (c ^ swz<NCH>(n))];
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int r = (wm * FM + fm) * 16 + lr + rsh;
        xf[fm] = Xb[r * NCH + (c ^ swz<NCH>(r))];
      }
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) Mma<T>::run(acc[fm][fn], wf[fn], xf[fm]);
    }
    wbuf ^= 1;
  }

  // ---- epilogue: lane holds channels co..co+3 of row t for each fragment ----
  T* yb = reinterpret_cast<T*>(p.y) + (int64_t)b * p.T * p.ldy;
  const T* rb = p.res ? reinterpret_cast<const T*>(p.res) + (int64_t)b * p.T * p.ldr : nullptr;
  const T* r2b = p.res2 ? reinterpret_cast<const T*>(p.res2) + (int64_t)b * p.T * p.ldr2 : nullptr;
  // (epilogue parameters as local scalars: lambdas that capture the by-value argument block
  //  itself make the compiler keep a copy of it in scratch memory)
  const float* const e_bias = p.bias;
  const int e_act = p.act, e_T = p.T, e_Cout = p.Cout, e_ldy = p.ldy, e_ldr = p.ldr, e_ldr2 = p.ldr2;
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
    for (int e = 0; e < 4; ++e) v[e] = keep ? act_apply(v[e], e_act) * e_scale : 0.f;
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
          u = keep ? act_apply(u, e_act) * e_scale : 0.f;
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
              v0[0] += __uint_as_float(r.x << 16) * e_rscale; v0[1] += __uint_as_float(r.x & 0xffff0000u) * e_rscale;
              v0[2] += __uint_as_float(r.y << 16) * e_rscale; v0[3] += __uint_as_float(r.y & 0xffff0000u) * e_rscale;
              v1[0] += __uint_as_float(r.z << 16) * e_rscale; v1[1] += __uint_as_float(r.z & 0xffff0000u) * e_rscale;
              v1[2] += __uint_as_float(r.w << 16) * e_rscale; v1[3] += __uint_as_float(r.w & 0xffff0000u) * e_rscale;
            }
            if (r2b) {
              const uint4 r = *reinterpret_cast<const uint4*>(r2b + (int64_t)t * e_ldr2 + co);
              v0[0] += __uint_as_float(r.x << 16); v0[1] += __uint_as_float(r.x & 0xffff0000u);
              v0[2] += __uint_as_float(r.y << 16); v0[3] += __uint_as_float(r.y & 0xffff0000u);
              v1[0] += __uint_as_float(r.z << 16); v1[1] += __uint_as_float(r.z & 0xffff0000u);
              v1[2] += __uint_as_float(r.w << 16); v1[3] += __uint_as_float(r.w & 0xffff0000u);
            }
            uint4 o;
            o.x = (uint32_t)f32_to_bf16(v0[0]) | ((uint32_t)f32_to_bf16(v0[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(v0[2]) | ((uint32_t)f32_to_bf16(v0[3]) << 16);
            o.z = (uint32_t)f32_to_bf16(v1[0]) | ((uint32_t)f32_to_bf16(v1[1]) << 16);
            o.w = (uint32_t)f32_to_bf16(v1[2]) | ((uint32_t)f32_to_bf16(v1[3]) << 16);
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

template <typename T, int NCH, int WM, int FM, int FN, int NW = 4>
int launch_cfg(ConvP& p, hipStream_t st) {
  constexpr int WN = NW / WM;
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
  p.nMT = (p.T + BM - 1) / BM;
  p.nNT = (p.Cout + BN - 1) / BN;
  const int BMW = BM + (p.ks - 1) * p.dil;
  const size_t smem = (size_t)(2 * BN * NCH + 2 * BMW * NCH) * 16;
  if (smem > 160 * 1024) {
    ptpp_set_error("conv1d: LDS window too large (%zu B)", smem);
    return PTPP_EINVAL;
  }
  auto kern = conv1d_cl_kernel<T, NCH, WM, FM, FN, NW>;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t nblk = (int64_t)p.B * p.nMT * p.nNT;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NW * 64), smem, st, p);
  PTPP_CHECK_LAUNCH("conv1d_fwd");
  return PTPP_OK;
}

template <typename T, int NCH>
int launch_tiles(ConvP& p, hipStream_t st) {
  // Tile choice: BN follows Cout; BM follows the per-utterance length so short
  // (phone-level) sequences do not waste MFMA work on padding rows.
  // Many small waves per block: measured (tools/bench_wgrad.py) 1.3-2.6x faster than 4 waves of
  // 64 x 64 on every shape of the training step -- with K = ks*Cin of a few hundred the K loop is
  // dominated by global->LDS latency and barriers, which 4-8 waves per SIMD hide and 2 do not.
  if (p.Cout <= 32) return launch_cfg<T, NCH, 8, 2, 2, 8>(p, st);      // 256 x 32, 8 waves of 32 x 32
  if (p.Cout <= 64) return launch_cfg<T, NCH, 4, 2, 2, 8>(p, st);      // 128 x 64, 8 waves of 32 x 32
  if (p.T <= 48) return launch_cfg<T, NCH, 2, 1, 2, 8>(p, st);         //  32 x 128, 8 waves of 16 x 32
  if (p.T <= 96 || (p.T % 128 != 0 && p.T % 128 <= 64 && p.T < 512))
    return launch_cfg<T, NCH, 2, 2, 2, 8>(p, st);                      //  64 x 128, 8 waves of 32 x 32
  return launch_cfg<T, NCH, 4, 2, 2, 16>(p, st);                       // 128 x 128, 16 waves of 32 x 32
}

}  // namespace

extern "C" int ptpp_conv_cin_padded(int cin, int dtype) {
  const int kc = dtype == PTPP_BF16 ? 8 : 4;
  if (cin % (8 * kc) == 0) return cin;
  const int q = 4 * kc;
  return (cin + q - 1) / q * q;
}

// Extended argument block (adds the second residual of the AMP-block mean).
extern "C" int ptpp_conv1d_fwd_ex(const ptpp_conv1d_args* a, const void* res2, int ldr2, float res_scale,
                                  float drop_p, uint64_t drop_seed, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->wp && a->y, "conv1d: null pointer");
  PTPP_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16, "conv1d: bad dtype %d", a->dtype);
  const int kc = a->dtype == PTPP_BF16 ? 8 : 4;
  const int es = a->dtype == PTPP_BF16 ? 2 : 4;
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->Cin > 0 && a->Cout > 0 && a->ks > 0 && a->dil > 0,
                 "conv1d: bad shape B=%d T=%d Cin=%d Cout=%d ks=%d dil=%d", a->B, a->T, a->Cin, a->Cout, a->ks, a->dil);
  PTPP_CHECK_ARG(a->Cin % kc == 0, "conv1d: Cin=%d must be a multiple of %d for this dtype", a->Cin, kc);
  PTPP_CHECK_ARG(a->ldx % kc == 0 && ((uintptr_t)a->x % 16) == 0, "conv1d: x rows must be 16-byte aligned");
  PTPP_CHECK_ARG(((uintptr_t)a->wp % 16) == 0, "conv1d: packed weight must be 16-byte aligned");
  if ((a->Cout & 3) == 0) {
    PTPP_CHECK_ARG(a->ldy % 4 == 0 && ((uintptr_t)a->y % (4 * es)) == 0, "conv1d: y rows must be vector aligned");
    if (a->res) PTPP_CHECK_ARG(a->ldr % 4 == 0 && ((uintptr_t)a->res % (4 * es)) == 0, "conv1d: res misaligned");
    if (res2) PTPP_CHECK_ARG(ldr2 % 4 == 0 && ((uintptr_t)res2 % (4 * es)) == 0, "conv1d: res2 misaligned");
    if (a->bias) PTPP_CHECK_ARG(((uintptr_t)a->bias % 16) == 0, "conv1d: bias misaligned");
  }
  PTPP_CHECK_ARG(!(a->in_mask || a->out_mask) || a->lengths, "conv1d: masks need lengths");
  ConvP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.res = a->res; p.res2 = res2; p.y = a->y;
  p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.Cout = a->Cout; p.ks = a->ks; p.dil = a->dil; p.pad = a->pad;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldr = a->ldr; p.ldr2 = ldr2;
  p.cinp = ptpp_conv_cin_padded(a->Cin, a->dtype);
  p.act = a->act; p.in_mask = a->in_mask; p.out_mask = a->out_mask;
  p.out_scale = a->out_scale; p.res_scale = res_scale;
  PTPP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "conv1d: bad dropout p");
  PTPP_CHECK_ARG(drop_p == 0.f || (a->Cout & 3) == 0, "conv1d: dropout needs Cout %% 4 == 0");
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = (p.cinp % (8 * kc)) == 0;
  if (a->dtype == PTPP_F32)
    return wide ? launch_tiles<float, 8>(p, st) : launch_tiles<float, 4>(p, st);
  return wide ? launch_tiles<bf16_raw, 8>(p, st) : launch_tiles<bf16_raw, 4>(p, st);
}

extern "C" int ptpp_conv1d_fwd(const ptpp_conv1d_args* a, void* stream) {
  return ptpp_conv1d_fwd_ex(a, nullptr, 0, 1.0f, 0.f, 0, stream);
}
