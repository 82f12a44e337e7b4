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
#include <stdlib.h>

#include "conv1d_common.h"
#include "conv1d_glds.h"

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

template <>
struct Mma<f16_raw> {
  static __device__ __forceinline__ void run(f32x4& acc, const uint4& a, const uint4& b) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
  }
};

// SK: split-K variant -- block (b, mt, nt, split) walks only its share of the Cin chunks and stores raw
// f32 partial sums; conv_splitk_finish_kernel adds the splits and applies the epilogue.  For layers with
// few output tiles and a long K (Conformer FFN k = 9: 144 K steps, BERT FFN: 48) every K step costs a
// global -> LDS round trip that one or two resident blocks per CU cannot hide; more blocks can.
template <typename T, int NCH, int WM, int FM, int FN, int NW = 4, bool SK = false>
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
  const int ball = lid / (p.nNT * p.nMT);
  const int b = SK ? ball % p.B : ball;
  const int split = SK ? ball / p.B : 0;
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
  int sbeg = 0, steps = ((p.out_mask && t0 >= len) || (p.in_mask && t0 - p.pad >= len)) ? 0 : nC * p.ks;
  if constexpr (SK) {  // whole chunks per split, so a split starts at tap 0 of a chunk
    sbeg = (int)((int64_t)split * nC / p.nsplit) * p.ks;
    steps = (int)((int64_t)(split + 1) * nC / p.nsplit) * p.ks;
  }

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

  if (steps > sbeg) load_w(sbeg);
  int wbuf = 0, xbuf = 1;
  for (int s = sbeg; s < steps; ++s) {
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

  if constexpr (SK) {
    float* wsb = p.ws + ((int64_t)split * p.B + b) * p.T * p.Cout;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const int t = t0 + (wm * FM + fm) * 16 + lr;
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        int co;  // the channel a lane's fragment holds: same maps as conv_epilogue
        if constexpr (sizeof(T) == 2 && FN % 2 == 0) co = n0 + wn * FN * 16 + (fn >> 1) * 32 + lg * 8 + (fn & 1) * 4;
        else co = n0 + (wn * FN + fn) * 16 + lg * 4;
        if (t < p.T && co < p.Cout) *reinterpret_cast<f32x4*>(wsb + (int64_t)t * p.Cout + co) = acc[fm][fn];
      }
    }
  } else {
    conv_epilogue<T, FM, FN>(p, acc, b, t0, n0, wm, wn, lane, len);
  }
}

// sum of the split-K partials + the epilogue of conv_epilogue (bias, activation, mask, scale, dropout,
// residuals), 4 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvP p) {
  const int cv = p.Cout >> 2;
  const int64_t nvec = (int64_t)p.B * p.T * cv;
  const int64_t slab = (int64_t)p.B * p.T * p.Cout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / cv;
    const int co = (int)(i - row * cv) * 4;
    const int b = (int)(row / p.T), t = (int)(row - (int64_t)b * p.T);
    f32x4 v = *reinterpret_cast<const f32x4*>(p.ws + row * p.Cout + co);
    for (int s = 1; s < p.nsplit; ++s) v += *reinterpret_cast<const f32x4*>(p.ws + s * slab + row * p.Cout + co);
    const bool keep = !(p.out_mask && t >= min(p.lengths[b], p.T));
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + co);
    act_dispatch(p.act, [&](auto tag) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? act_apply_c<decltype(tag)::value>(v[e], p.act) * p.out_scale : 0.f;
    });
    if (p.drop_thresh16) v *= drop_mask4(p.drop_seed, (uint64_t)(row * p.Cout + co) >> 2, p.drop_thresh16, p.drop_inv_keep);
    if (p.res) v += Elem<T>::ld4(reinterpret_cast<const T*>(p.res) + row * p.ldr + co) * p.res_scale;
    if (p.res2) v += Elem<T>::ld4(reinterpret_cast<const T*>(p.res2) + row * p.ldr2 + co);
    Elem<T>::st4(reinterpret_cast<T*>(p.y) + row * p.ldy + co, v);
  }
}

// split-K launch of a small-tile configuration (see conv1d_cl_kernel): nsplit chosen so that about four
// 8-wave blocks per CU are in flight; needs the caller's workspace
template <typename T, int NCH, int WM, int FM, int FN, int NW>
bool launch_splitk(ConvP& p, hipStream_t st, void* ws, size_t ws_bytes, int* status) {
  constexpr int WN = NW / WM;
  constexpr int BM = WM * FM * 16, BN = WN * FN * 16;
  constexpr int KC = 16 / (int)sizeof(T);
  const int nC = p.cinp / (NCH * KC);
  const int nMT = (p.T + BM - 1) / BM, nNT = (p.Cout + BN - 1) / BN;
  const int64_t blocks = (int64_t)p.B * nMT * nNT;
  if (!ws || (p.Cout & 3) || p.act == PTPP_ACT_GATE || (int64_t)nC * p.ks < 32 || blocks >= 512) return false;
  int64_t ns = (1024 + blocks - 1) / blocks;
  if (ns > 8) ns = 8;
  if (ns > nC) ns = nC;
  const int64_t slab = (int64_t)p.B * p.T * p.Cout * (int64_t)sizeof(float);
  if (ns * slab > (int64_t)ws_bytes) ns = (int64_t)ws_bytes / slab;
  if (ns < 2) return false;
  p.ws = reinterpret_cast<float*>(ws);
  p.nsplit = (int)ns;
  p.nMT = nMT;
  p.nNT = nNT;
  const int BMW = BM + (p.ks - 1) * p.dil;
  const size_t smem = (size_t)(2 * BN * NCH + 2 * BMW * NCH) * 16;
  if (smem > 160 * 1024) return false;
  auto kern = conv1d_cl_kernel<T, NCH, WM, FM, FN, NW, true>;
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_fwd (split-K)")) {
    *status = PTPP_ELAUNCH;
    return true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(blocks * ns)), dim3(NW * 64), smem, st, p);
  const int64_t nvec = (int64_t)p.B * p.T * (p.Cout >> 2);
  int64_t fb = (nvec + 255) / 256;
  if (fb > 4096) fb = 4096;
  hipLaunchKernelGGL(conv_splitk_finish_kernel<T>, dim3((unsigned)fb), dim3(256), 0, st, p);
  *status = hipGetLastError() == hipSuccess ? PTPP_OK : PTPP_ELAUNCH;
  if (*status != PTPP_OK) ptpp_set_error("conv1d_fwd (split-K): launch failed");
  return true;
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
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_fwd")) return PTPP_ELAUNCH;
  const int64_t nblk = (int64_t)p.B * p.nMT * p.nNT;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NW * 64), smem, st, p);
  PTPP_CHECK_LAUNCH("conv1d_fwd");
  return PTPP_OK;
}

// ---- skinny GEMM: a handful of rows (one per utterance: step-embedding MLP, prompt adaptor, GRU,
// style heads) against a long K.  The tile kernel gives such a launch 2-8 blocks that walk K serially
// (33-70 us for a 32 x 1024 x 256 product, all of it load -> LDS -> barrier latency).  Here a block owns
// one 16 x 16 output tile and its 4 waves split K: fragments come straight from global memory
// (16 bytes per lane, no LDS, no barrier in the loop), partial tiles meet in LDS, wave 0 runs the
// shared epilogue.  Rows are addressed linearly (row r = b*T + t; batches are T consecutive rows).
template <typename T>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const ConvP p) {
  constexpr int KC = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int nt = blockIdx.x % p.nNT, mt = blockIdx.x / p.nNT;
  const int n0 = nt * 16, t0 = mt * 16;
  const int rows = p.T;  // the launcher flattened (B, T) to (1, B*T)
  const T* xrow = reinterpret_cast<const T*>(p.x) + (int64_t)min(t0 + lr, rows - 1) * p.ldx;
  const T* wrow = reinterpret_cast<const T*>(p.wp) + (int64_t)min(n0 + lr, p.Cout - 1) * p.cinp;
  const int xch = p.Cin / KC;  // the zero-padded tail chunks of the packed weight rows contribute nothing
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  // loop bounds are WAVE-UNIFORM (an MFMA needs every lane): a lane whose chunk lies past the row
  // contributes zeros instead of skipping
  const uint4 zero = uint4{0u, 0u, 0u, 0u};
  int cb = wave * 4;
  for (; cb + 48 + 3 < xch; cb += 64) {  // four independent 16-byte fragment pairs in flight per lane
    uint4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const uint4*>(wrow + (int64_t)(cb + lg + 16 * u) * KC);
      b[u] = *reinterpret_cast<const uint4*>(xrow + (int64_t)(cb + lg + 16 * u) * KC);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) Mma<T>::run(acc, a[u], b[u]);
  }
  for (; cb < xch; cb += 16) {
    const int c = cb + lg;
    const bool in = c < xch;
    const uint4 a = in ? *reinterpret_cast<const uint4*>(wrow + (int64_t)c * KC) : zero;
    const uint4 b = in ? *reinterpret_cast<const uint4*>(xrow + (int64_t)c * KC) : zero;
    Mma<T>::run(acc, a, b);
  }
  __shared__ f32x4 part[3][64];
  if (wave) part[wave - 1][lane] = acc;
  __syncthreads();
  if (wave) return;
  f32x4 tot[1][1];
  tot[0][0] = acc + part[0][lane] + part[1][lane] + part[2][lane];
  conv_epilogue<T, 1, 1>(p, tot, 0, t0, n0, 0, 0, lane, rows);
}

inline bool skinny_ok(const ConvP& p) {
  return p.ks == 1 && (int64_t)p.B * p.T <= 128 && p.cinp >= 256 && !p.in_mask && !p.out_mask && !p.res2 &&
         p.drop_thresh16 == 0 && p.act != PTPP_ACT_GATE;
}

template <typename T>
int launch_skinny(ConvP& p, hipStream_t st) {
  p.T = p.B * p.T;
  p.B = 1;
  p.nMT = (p.T + 15) / 16;
  p.nNT = (p.Cout + 15) / 16;
  hipLaunchKernelGGL(gemm_skinny_kernel<T>, dim3((unsigned)(p.nMT * p.nNT)), dim3(256), 0, st, p);
  PTPP_CHECK_LAUNCH("conv1d_fwd (skinny)");
  return PTPP_OK;
}

template <typename T, int NCH>
int launch_tiles(ConvP& p, hipStream_t st, void* ws, size_t ws_bytes) {
  if (skinny_ok(p)) return launch_skinny<T>(p, st);
  int sk_status = PTPP_OK;
  // Tile choice: BN follows Cout; BM follows the per-utterance length so short
  // (phone-level) sequences do not waste MFMA work on padding rows.
  // Many small waves per block: measured (tools/bench_wgrad.py) 1.3-2.6x faster than 4 waves of
  // 64 x 64 on every shape of the training step -- with K = ks*Cin of a few hundred the K loop is
  // dominated by global->LDS latency and barriers, which 4-8 waves per SIMD hide and 2 do not.
  if (p.Cout <= 32) return launch_cfg<T, NCH, 8, 2, 2, 8>(p, st);      // 256 x 32, 8 waves of 32 x 32
  if (p.Cout <= 64) return launch_cfg<T, NCH, 4, 2, 2, 8>(p, st);      // 128 x 64, 8 waves of 32 x 32
  // The small-tile configurations first try split-K (long K, few tiles: launch_splitk decides).
  if (p.T <= 48) {
    if (p.T > 32 && launch_splitk<T, NCH, 2, 2, 2, 8>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
    if (launch_splitk<T, NCH, 2, 1, 2, 8>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
    return launch_cfg<T, NCH, 2, 1, 2, 8>(p, st);                      //  32 x 128, 8 waves of 16 x 32
  }
  if constexpr (IsBf16<T>::value && NCH == 8) {  // (the LDS-DMA kernels are bf16 only)
    // few output tiles and a long K (the Conformer feed-forward k = 9 convs at phone level: 1024 -> 256 over ~150 rows per
    // utterance = 114 tiles, 144 K steps): split-K on the LDS-DMA kernel -- a K step there costs ~0.5 us against ~3 us in
    // the register-staged 8-wave kernel below.  PTPP_CONV_GLDS_SPLITK=0 keeps the old route.
    static const char* gsk = getenv("PTPP_CONV_GLDS_SPLITK");
    // (128 x 128 tiles for the split: 46.3 vs 44.3 us on the 1024 -> 256 k = 9 shape)
    if (!(gsk && gsk[0] == '0') && launch_glds_splitk<2, 4, 2, 2, 2>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
    // enough 64-row tiles for every CU and a K loop of >= 32 steps (256 -> 1024, k = 9 at phone level: 456 tiles, 36 steps):
    // the LDS-DMA kernel without a split instead of the 8-wave split-K kernel (18.67 -> 18.52 ms per step;
    // PTPP_CONV_GLDS_SHORT=0 keeps the old route)
    static const char* gsh = getenv("PTPP_CONV_GLDS_SHORT");
    if (!(gsh && gsh[0] == '0') && glds_ok(p) && (p.T <= 96 || (p.T % 128 != 0 && p.T % 128 <= 64 && p.T < 512)) &&
        (long long)p.B * ((p.T + 63) / 64) * ((p.Cout + 127) / 128) >= 384 && (long long)(p.cinp >> 6) * p.ks >= 32) {
      const int rc = launch_glds<2, 4, 2, 2, 2>(p, st);
      if (rc >= 0) return rc;
    }
  }
  if (p.T <= 96 || (p.T % 128 != 0 && p.T % 128 <= 64 && p.T < 512)) {
    if (launch_splitk<T, NCH, 2, 2, 2, 8>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
    return launch_cfg<T, NCH, 2, 2, 2, 8>(p, st);                      //  64 x 128, 8 waves of 32 x 32
  }
  // a grid that cannot fill the 256 CUs with 128-row tiles (the frozen BERT's (B*tokens) x 768 GEMMs:
  // 60-72 blocks, 59 us each) gets smaller tiles: its K loop is latency-bound, more blocks = more overlap
  const long long nt = (p.Cout + 127) / 128;
  if ((long long)p.B * ((p.T + 127) / 128) * nt < 192) {
    // long K: keep the 128-row tiles (every weight tile fetched from L2 serves 128 rows, not 32) and get the
    // parallelism from split-K instead
    if (launch_splitk<T, NCH, 4, 2, 2, 16>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
    if ((long long)p.B * ((p.T + 63) / 64) * nt < 192) {
      if (launch_splitk<T, NCH, 2, 1, 2, 8>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
      return launch_cfg<T, NCH, 2, 1, 2, 8>(p, st);
    }
    if (launch_splitk<T, NCH, 2, 2, 2, 8>(p, st, ws, ws_bytes, &sk_status)) return sk_status;
    return launch_cfg<T, NCH, 2, 2, 2, 8>(p, st);
  }
  if constexpr (IsBf16<T>::value && NCH == 8) {  // (the LDS-DMA kernels are bf16 only)
    // LDS-DMA pipeline (conv1d_glds.h), 4 waves per block.  Large grids (the vocoder: thousands of 128 x 128 tiles,
    // many rounds over the 256 CUs) take 128 x 128 tiles with 64 x 64 wave tiles -- the highest MFMA : L2-traffic
    // ratio; the frame-level shapes of the acoustic model (B x 576: about two rounds) take 64 x 128 tiles, three
    // blocks per CU, finer tail (measured: profiles/r02_conv_glds.txt).  PTPP_CONV_GLDS=0 keeps the register-staged
    // kernel, A / E force one of the two.
    static const char* gl = getenv("PTPP_CONV_GLDS");
    if (glds_ok(p) && !(gl && gl[0] == '0')) {
      const long long tiles128 = (long long)p.B * ((p.T + 127) / 128) * ((p.Cout + 127) / 128);
      static const char* thr = getenv("PTPP_CONV_GLDS_TILES");
      const bool big = gl && gl[0] == 'A' ? true : gl && gl[0] == 'E' ? false : tiles128 >= (thr ? atoll(thr) : 1536);
      const int rc = big ? launch_glds<4, 4, 2, 2, 2>(p, st) : launch_glds<2, 4, 2, 2, 2>(p, st);
      if (rc >= 0) return rc;
    }
  }
  // (64 x 128 and 256 x 64 tiles measured 10-45 % slower on the frame-level shapes)
  return launch_cfg<T, NCH, 4, 2, 2, 16>(p, st);                       // 128 x 128, 16 waves of 32 x 32
}

}  // namespace

extern "C" int ptpp_conv_cin_padded(int cin, int dtype) {
  const int kc = dtype == PTPP_F32 ? 4 : 8;
  if (cin % (8 * kc) == 0) return cin;
  const int q = 4 * kc;
  return (cin + q - 1) / q * q;
}

// Full form: second residual of the AMP-block mean, fused dropout, optional split-K workspace.
extern "C" int ptpp_conv1d_fwd_ws(const ptpp_conv1d_args* a, const void* res2, int ldr2, float res_scale,
                                  float drop_p, uint64_t drop_seed, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->wp && a->y, "conv1d: null pointer");
  PTPP_CHECK_ARG(a->dtype == PTPP_F32 || a->dtype == PTPP_BF16 || a->dtype == PTPP_F16, "conv1d: bad dtype %d", a->dtype);
  const int kc = a->dtype == PTPP_F32 ? 4 : 8;
  const int es = a->dtype == PTPP_F32 ? 4 : 2;
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
  if (a->act == PTPP_ACT_GATE)
    PTPP_CHECK_ARG(a->dtype == PTPP_BF16 && (a->Cout & 7) == 0 && (a->ldy & 7) == 0 && ((uintptr_t)a->y & 15) == 0 &&
                       (!a->res || ((a->ldr & 7) == 0 && ((uintptr_t)a->res & 15) == 0)) && !res2 && drop_p == 0.f,
                   "conv1d: the fused gate epilogue needs bf16, Cout %% 8 == 0 and 16-byte aligned rows");
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
  p.ws = nullptr;
  p.nsplit = 1;
  p.post_skip = nullptr; p.post_dnext = nullptr; p.post_yin = nullptr; p.post_C = 0; p.post_init = 0;
  p.gate_a = nullptr; p.gate_da = nullptr; p.gate_ldda = 0; p.gate_save = nullptr; p.gate_lds = 0;
  if (workspace && ((uintptr_t)workspace & 15)) workspace = nullptr;
  // A linear layer without sequence masks does not care where one utterance ends: treat the (B, T) rows
  // as ONE sequence (rows are linear in memory: batches are T consecutive rows) so that short utterances
  // (BERT: ~25 tokens, phone level: ~100) fill whole 128-row tiles instead of padding each to a tile
  if (p.ks == 1 && !p.in_mask && !p.out_mask && p.B > 1) {
    p.T = p.B * p.T;
    p.B = 1;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = (p.cinp % (8 * kc)) == 0;
  if (a->dtype == PTPP_F32)
    return wide ? launch_tiles<float, 8>(p, st, workspace, workspace_bytes)
                : launch_tiles<float, 4>(p, st, workspace, workspace_bytes);
  if (a->dtype == PTPP_F16)
    return wide ? launch_tiles<f16_raw, 8>(p, st, workspace, workspace_bytes)
                : launch_tiles<f16_raw, 4>(p, st, workspace, workspace_bytes);
  return wide ? launch_tiles<bf16_raw, 8>(p, st, workspace, workspace_bytes)
              : launch_tiles<bf16_raw, 4>(p, st, workspace, workspace_bytes);
}

// The epilogue pass over f32 partial sums [nsplit][B][T][Cout] that ANOTHER kernel produced (the row-tile conv's split over Cin,
// conv1d_rt.hip): conv_splitk_finish_kernel with the epilogue of ptpp_conv1d_fwd_ex on these arguments (bf16).
int ptpp_conv_splitk_finish_bf16(const ptpp_conv1d_args* a, float res_scale, float drop_p, uint64_t drop_seed, float* ws, int nsplit,
                                 hipStream_t st) {
  ConvP p;
  p.x = a->x; p.wp = nullptr; p.bias = a->bias; p.res = a->res; p.res2 = nullptr; p.y = a->y;
  p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.Cout = a->Cout; p.ks = a->ks; p.dil = a->dil; p.pad = a->pad;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldr = a->ldr; p.ldr2 = 0;
  p.cinp = a->Cin;
  p.act = a->act; p.in_mask = a->in_mask; p.out_mask = a->out_mask;
  p.out_scale = a->out_scale; p.res_scale = res_scale;
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  p.ws = ws;
  p.nsplit = nsplit;
  p.nMT = p.nNT = 0;
  p.post_skip = nullptr; p.post_dnext = nullptr; p.post_yin = nullptr; p.post_C = 0; p.post_init = 0;
  p.gate_a = nullptr; p.gate_da = nullptr; p.gate_ldda = 0; p.gate_save = nullptr; p.gate_lds = 0;
  const int64_t nvec = (int64_t)p.B * p.T * (p.Cout >> 2);
  int64_t fb = (nvec + 255) / 256;
  if (fb > 4096) fb = 4096;
  hipLaunchKernelGGL(conv_splitk_finish_kernel<bf16_raw>, dim3((unsigned)fb), dim3(256), 0, st, p);
  PTPP_CHECK_LAUNCH("conv1d (split finish)");
  return PTPP_OK;
}

extern "C" int ptpp_conv1d_fwd_ex(const ptpp_conv1d_args* a, const void* res2, int ldr2, float res_scale,
                                  float drop_p, uint64_t drop_seed, void* stream) {
  return ptpp_conv1d_fwd_ws(a, res2, ldr2, res_scale, drop_p, drop_seed, nullptr, 0, stream);
}

extern "C" int ptpp_conv1d_fwd(const ptpp_conv1d_args* a, void* stream) {
  return ptpp_conv1d_fwd_ex(a, nullptr, 0, 1.0f, 0.f, 0, stream);
}

// ---- DiffNet output projection with the layer's tail fused into the epilogue (conv1d_glds.h::tile_epilogue_post) ----
extern "C" int ptpp_conv1d_diffnet_post_supported(int C, int cin, int dtype) {
  return dtype == PTPP_BF16 && C > 0 && C % 128 == 0 && cin > 0 && cin % 64 == 0;
}

extern "C" int ptpp_conv1d_diffnet_post(const ptpp_conv1d_args* a, const void* x, float* skip, const float* dnext, void* xn,
                                        void* yin, int init, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->wp && x && skip && xn, "conv1d_diffnet_post: null pointer");
  const int C = a->Cout / 2;
  PTPP_CHECK_ARG(a->ks == 1 && a->pad == 0 && a->Cout == 2 * C && ptpp_conv1d_diffnet_post_supported(C, a->Cin, a->dtype),
                 "conv1d_diffnet_post: needs bf16, a 1 x 1 projection to 2C channels, C %% 128 == 0, Cin %% 64 == 0 (C=%d Cin=%d)", C,
                 a->Cin);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->ldx % 8 == 0 && ((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->wp % 16) == 0 &&
                     ((uintptr_t)x % 16) == 0 && ((uintptr_t)xn % 16) == 0 && ((uintptr_t)yin % 16) == 0 &&
                     ((uintptr_t)skip % 16) == 0 && ((uintptr_t)dnext % 16) == 0 && (!a->bias || ((uintptr_t)a->bias % 16) == 0),
                 "conv1d_diffnet_post: operands must be 16-byte aligned");
  PTPP_CHECK_ARG(!a->out_mask || a->lengths, "conv1d_diffnet_post: masks need lengths");
  ConvP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.res = x; p.res2 = nullptr; p.y = xn; p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.Cout = a->Cout; p.ks = 1; p.dil = 1; p.pad = 0;
  p.ldx = a->ldx; p.ldy = C; p.ldr = C; p.ldr2 = 0;
  p.cinp = a->Cin;
  p.act = PTPP_ACT_NONE; p.in_mask = a->in_mask; p.out_mask = a->out_mask;
  p.out_scale = a->out_scale; p.res_scale = 1.f;
  p.drop_thresh16 = 0; p.drop_inv_keep = 1.f; p.drop_seed = 0;
  p.ws = nullptr; p.nsplit = 1;
  p.post_skip = skip; p.post_dnext = dnext; p.post_yin = yin; p.post_C = C; p.post_init = init;
  p.gate_a = nullptr; p.gate_da = nullptr; p.gate_ldda = 0; p.gate_save = nullptr; p.gate_lds = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long tiles128 = (long long)p.B * ((p.T + 127) / 128) * ((p.Cout + 127) / 128);
  const int rc = tiles128 >= 1536 ? launch_glds<4, 4, 2, 2, 2>(p, st) : launch_glds<2, 4, 2, 2, 2>(p, st);
  if (rc < 0) {
    ptpp_set_error("conv1d_diffnet_post: tile does not fit LDS");
    return PTPP_EINVAL;
  }
  return rc;
}

// ---- DiffNet output projection's data gradient with the gate backward fused into the epilogue ----
extern "C" int ptpp_conv1d_gate_bwd_supported(int C, int cin, int dtype) {
  return dtype == PTPP_BF16 && C > 0 && C % 8 == 0 && C >= 64 && cin > 0 && cin % 64 == 0;
}

extern "C" int ptpp_conv1d_gate_bwd(const ptpp_conv1d_args* a, const void* act, void* da, int ldda, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->wp && act && da, "conv1d_gate_bwd: null pointer");
  const int C = a->Cout;
  PTPP_CHECK_ARG(a->ks == 1 && a->pad == 0 && ptpp_conv1d_gate_bwd_supported(C, a->Cin, a->dtype),
                 "conv1d_gate_bwd: needs bf16, a 1 x 1 projection, C %% 8 == 0, Cin %% 64 == 0 (C=%d Cin=%d)", C, a->Cin);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->ldx % 8 == 0 && ldda % 8 == 0 && ldda >= 2 * C && ((uintptr_t)a->x % 16) == 0 &&
                     ((uintptr_t)a->wp % 16) == 0 && ((uintptr_t)act % 16) == 0 && ((uintptr_t)da % 16) == 0 && !a->bias &&
                     !a->out_mask && (!a->in_mask || a->lengths),
                 "conv1d_gate_bwd: operands must be 16-byte aligned, no bias / output mask; an input mask needs lengths");
  ConvP p;
  p.x = a->x; p.wp = a->wp; p.bias = nullptr; p.res = nullptr; p.res2 = nullptr; p.y = da; p.lengths = a->in_mask ? a->lengths : nullptr;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.Cout = C; p.ks = 1; p.dil = 1; p.pad = 0;
  p.ldx = a->ldx; p.ldy = ldda; p.ldr = 0; p.ldr2 = 0;
  p.cinp = a->Cin;
  p.act = PTPP_ACT_NONE; p.in_mask = a->in_mask; p.out_mask = 0;
  p.out_scale = a->out_scale; p.res_scale = 1.f;
  p.drop_thresh16 = 0; p.drop_inv_keep = 1.f; p.drop_seed = 0;
  p.ws = nullptr; p.nsplit = 1;
  p.post_skip = nullptr; p.post_dnext = nullptr; p.post_yin = nullptr; p.post_C = 0; p.post_init = 0;
  p.gate_a = act; p.gate_da = da; p.gate_ldda = ldda; p.gate_save = nullptr; p.gate_lds = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long tiles128 = (long long)p.B * ((p.T + 127) / 128) * ((p.Cout + 127) / 128);
  const int rc = tiles128 >= 1536 ? launch_glds<4, 4, 2, 2, 2>(p, st) : launch_glds<2, 4, 2, 2, 2>(p, st);
  if (rc < 0) {
    ptpp_set_error("conv1d_gate_bwd: tile does not fit LDS");
    return PTPP_EINVAL;
  }
  return rc;
}


// ---- DiffNet dilated conv (+ conditioner slice) with the gate fused into the epilogue AND the pre-activation kept ----
// Training forward (modules/denoiser.py:76-79): a = conv(yin) + bias + cond slice, g = sigmoid(a[:, :C]) * tanh(a[:, C:]).
// The weights (and bias, and the cond slice) are in the gate-interleaved row order of PTPP_ACT_GATE (ptpp_pack_conv_weight
// mode 2); g (B, T, C) and a (B, T, 2C, STANDARD [gate | filter] order, for the backward) leave in one launch, bit for bit
// what ptpp_conv1d_fwd + ptpp_gate_fwd produce.
extern "C" int ptpp_conv1d_gate_fwd_save_supported(int C, int cin, int dtype) {
  return dtype == PTPP_BF16 && C > 0 && C % 64 == 0 && cin > 0 && cin % 64 == 0;
}

extern "C" int ptpp_conv1d_gate_fwd_save(const ptpp_conv1d_args* a, void* a_out, int lda, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->wp && a->y && a_out, "conv1d_gate_fwd_save: null pointer");
  const int C = a->Cout / 2;
  PTPP_CHECK_ARG(a->Cout == 2 * C && ptpp_conv1d_gate_fwd_save_supported(C, a->Cin, a->dtype),
                 "conv1d_gate_fwd_save: needs bf16, 2C output channels with C %% 64 == 0, Cin %% 64 == 0 (C=%d Cin=%d)", C, a->Cin);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->ks > 0 && a->dil > 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0 && lda % 8 == 0 && lda >= 2 * C &&
                     (!a->res || a->ldr % 8 == 0) && ((uintptr_t)a->x % 16) == 0 && ((uintptr_t)a->wp % 16) == 0 &&
                     ((uintptr_t)a->y % 16) == 0 && ((uintptr_t)a_out % 16) == 0 && ((uintptr_t)a->res % 16) == 0 &&
                     (!a->bias || ((uintptr_t)a->bias % 16) == 0),
                 "conv1d_gate_fwd_save: operands must be 16-byte aligned with row strides in multiples of 8");
  PTPP_CHECK_ARG(!(a->in_mask || a->out_mask) || a->lengths, "conv1d_gate_fwd_save: masks need lengths");
  ConvP p;
  p.x = a->x; p.wp = a->wp; p.bias = a->bias; p.res = a->res; p.res2 = nullptr; p.y = a->y; p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.Cout = a->Cout; p.ks = a->ks; p.dil = a->dil; p.pad = a->pad;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldr = a->ldr; p.ldr2 = 0;
  p.cinp = a->Cin;
  p.act = PTPP_ACT_GATE; p.in_mask = a->in_mask; p.out_mask = a->out_mask;
  p.out_scale = a->out_scale; p.res_scale = 1.f;
  p.drop_thresh16 = 0; p.drop_inv_keep = 1.f; p.drop_seed = 0;
  p.ws = nullptr; p.nsplit = 1;
  p.post_skip = nullptr; p.post_dnext = nullptr; p.post_yin = nullptr; p.post_C = 0; p.post_init = 0;
  p.gate_a = nullptr; p.gate_da = nullptr; p.gate_ldda = 0;
  p.gate_save = a_out; p.gate_lds = lda;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long tiles128 = (long long)p.B * ((p.T + 127) / 128) * ((p.Cout + 127) / 128);
  const int rc = tiles128 >= 1536 ? launch_glds<4, 4, 2, 2, 2>(p, st) : launch_glds<2, 4, 2, 2, 2>(p, st);
  if (rc < 0) {
    ptpp_set_error("conv1d_gate_fwd_save: tile does not fit LDS");
    return PTPP_EINVAL;
  }
  return rc;
}
