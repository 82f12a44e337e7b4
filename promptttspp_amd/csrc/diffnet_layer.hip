// One DiffNet residual layer as ONE kernel (reference modules/denoiser.py:69-83):
//
//   a   = dilated_conv_k3(yin) + cond_slice            (yin = x + diffusion_step projection, written by the previous layer)
//   g   = sigmoid(a[:, :C]) * tanh(a[:, C:])
//   o   = output_projection(g)                          (1 x 1, C -> 2C)
//   xn  = (x + o[:, :C]) / sqrt(2);   skip += o[:, C:];   yin' = xn + dnext[b]
//
// C = 256, bf16 tensors, f32 accumulation.  Bit for bit the results of the two-launch path
// (ptpp_conv1d_gate_fwd_save / ptpp_conv1d_fwd with PTPP_ACT_GATE, then ptpp_conv1d_diffnet_post): same K order of the
// accumulation (64-channel chunk, tap, 32-channel MFMA step), same rounding points (a, g, o, xn, yin' to bf16).
//
// Structure: a block of 8 waves owns a 128-row tile of one utterance with ALL channels.
//   * ONE continuous weight stream: the layer's two weights are re-packed once per weight version
//     (ptpp_diffnet_pack_wstream) into 64 stages of 16 KiB that are already the LDS images the MFMA fragments are read from
//     (row permutation and XOR swizzle applied), in consumption order; a stage = [256 output channels][32 k] bf16.
//     The stages travel global -> LDS by LDS-DMA into a ring of NS stages, NS - 2 stages (32-48 KiB) in flight under the
//     MFMAs of the current one; the stream never drains between the passes (the gate epilogue runs while the first
//     stages of the output projection land).
//   * pass A, 48 steps (chunk ci, tap, k-half kh, channel half nh): the wave tile is 64 rows x 64 channels of each
//     256-channel half; both halves accumulate at once (2 x 16 MFMA tiles = 128 accumulator registers), so the x window
//     of a 64-channel chunk (128 + 2 dil rows, double-buffered LDS-DMA) is fetched ONCE and each x fragment serves two steps.
//   * gate epilogue: + bias + conditioner slice, a rounded and stored (training), g -> LDS as the B operand of pass B
//     ([128 rows][256 channels], XOR-swizzled 16-byte chunks; the region the x windows used) -- g never leaves the CU
//     (training stores a copy for the weight gradient).
//   * pass B, 16 steps (k32 step kc, channel half nh): nh = 0 is the residual half, nh = 1 the skip half.
//   * tail: xn / yin' / skip straight from the accumulators.
// LDS: NS x 16 KiB ring + 64 KiB = 144 KiB (NS = 5): one block per CU; a 30 000-frame batch is 235 blocks on 256 CUs.
#include <stdlib.h>

#include "conv1d_common.h"
#include "lds_dma.h"
#include <type_traits>

namespace {

constexpr int DN_C = 256;        // residual channels
constexpr int DN_STEPS = 64;     // 48 (dilated conv) + 16 (output projection)
constexpr int DN_STAGE_U4 = 1024;  // uint4 per stage (16 KiB)

struct DnLayerP {
  const bf16_raw* yin;
  const bf16_raw* x;
  const bf16_raw* cond;
  const uint4* wstream;
  const float* dil_b;
  const float* out_b;
  const float* dnext;
  float* skip;
  bf16_raw* xn;
  bf16_raw* yin_next;
  bf16_raw* a_out;
  bf16_raw* g_out;
  bf16_raw* skip_scaled;  // optional: bf16(skip * skip_scale), what the skip projection reads (last layer)
  const bf16_raw* condx;  // COND instantiation: the conditioner INPUT (B, T, 256), row stride ldcx -- its projection is 16 more
  int ldcx;               // stages of pass A instead of a precomputed (B, T, 2C) slice added in the gate epilogue
  const int* lengths;
  float skip_scale;
  int B, T, dil, ldc, init, nMT;
  unsigned long long* stamps;  // diagnostics only
};

__device__ __forceinline__ void lds_barrier() {
  // LDS reads / writes of this wave have completed, then the workgroup barrier; never waits for the LDS-DMA queue
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// two f32 -> packed bf16 (round to nearest even) in ONE v_cvt_pk_bf16_f32; equal to f32_to_bf16 on every non-NaN input
typedef __attribute__((ext_vector_type(2))) float dn_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 dn_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  const dn_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dn_bf16x2));
}
__device__ __forceinline__ float lo_bf16(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi_bf16(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ float round_bf16_(float v) { return lo_bf16(pack_bf16x2(v, 0.f)); }

// 16 MFMAs of one step: weights as the A operand, acc[fm][fn] += W[fn] * X[fm]
// element traits of the two 16-bit storage types the layer is built for (round 6: IEEE half for the sampler, BASELINE config 5
// "fp16 mel decoder"): conversions and the matrix instruction; everything else in the kernel moves raw 16-byte vectors
typedef __attribute__((ext_vector_type(8))) _Float16 dn_f16x8;
template <bool F16>
struct DnCv {
  static __device__ __forceinline__ float lo(uint32_t v) { return lo_bf16(v); }
  static __device__ __forceinline__ float hi(uint32_t v) { return hi_bf16(v); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return pack_bf16x2(a, b); }
  static __device__ __forceinline__ float round(float v) { return round_bf16_(v); }
  template <typename A, typename B>
  static __device__ __forceinline__ f32x4 mfma(const A& w, const B& x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
  }
};
template <>
struct DnCv<true> {
  static __device__ __forceinline__ float lo(uint32_t v) { return H2<f16_raw>::lo(v); }
  static __device__ __forceinline__ float hi(uint32_t v) { return H2<f16_raw>::hi(v); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return H2<f16_raw>::pack(a, b); }
  static __device__ __forceinline__ float round(float v) { return (float)(_Float16)v; }
  template <typename A, typename B>
  static __device__ __forceinline__ f32x4 mfma(const A& w, const B& x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dn_f16x8, w), __builtin_bit_cast(dn_f16x8, x), c, 0, 0, 0);
  }
};

template <bool NOMFMA = false, int FM = 4, int FN = 4, bool F16 = false>
__device__ __forceinline__ void dn_mfma_step(f32x4 (&acc)[FM][FN], const uint4 (&wf)[FN], const uint4 (&xf)[FM]) {
  if constexpr (NOMFMA) {  // keep the LDS reads alive without the matrix work
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      asm volatile("" ::"v"(wf[i].x), "v"(wf[i].w), "v"(xf[i % FM].x), "v"(xf[i % FM].w));
    }
    return;
  }
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
      acc[fm][fn] = DnCv<F16>::mfma(wf[fn], xf[fm], acc[fm][fn]);
}

template <int NS, int NSTEPS = DN_STEPS>
__device__ __forceinline__ void dn_wait_stage(int s) {
  // top of step s: this wave's pieces of stage s + 1 must have landed.  Issued so far: stages up to min(63, s + NS - 2), two
  // pieces per stage and wave, so min(NS - 3, 62 - s) younger stages may stay in flight (the x-window pieces issued in
  // between only make the wait stricter)
  const int younger = min(NS - 3, NSTEPS - 2 - s);
  if (younger >= 3) glds_wait<6>();
  else if (younger == 2) glds_wait<4>();
  else if (younger == 1) glds_wait<2>();
  else glds_wait<0>();
}

// DBG (tools only, ptpp_diffnet_layer_fwd_dbg): bit 0 = clock stamps per block into p.stamps, bit 1 = no MFMAs, bit 2 = no
// weight stream (the waits find nothing outstanding), bit 3 = no epilogue loads / stores
// GW (round 5): the weight fragments come STRAIGHT FROM GLOBAL MEMORY into registers, two steps ahead -- no LDS ring, no LDS-DMA
// of weights, no barrier inside a 64-channel chunk.  A fragment of the stream's stage image is 1 KiB contiguous (16 rows of 64
// bytes; the swizzle only permutes inside it), so the SAME stream serves both forms.  Why: the ring loop spends 700-900 cycles
// per step against 256 of MFMA issue -- a barrier, a counted vmcnt wait and ~40 scalar / address instructions per step for the
// DMA bookkeeping (profiles/r04_diffnet_layer_phases.txt; the same fragment reads ALONE run at 332 cycles per step with a
// barrier per step and 192 without, tools/experiments/lds_read_mimic.hip).  Here a chunk is 12 straight-line steps with
// compile-time tap / half / register-set indices; barriers stay at the chunk boundaries (x windows) and around the epilogues.
// FN (round 5): MFMA tiles per wave along the output channels.  4 = the 2 x 4 wave grid (a wave owns FM row tiles x 64 channels
// per half); 2 = a 1 x 8 grid (every wave owns ALL rows of the block x 32 channels): no two waves read the same weight fragment,
// so the GW form's L1 traffic halves (16 KiB per step and CU) -- the x fragments are what every wave reads, from LDS.
template <int NS, bool SAVE, int DBG = 0, int FM = 4, bool COND = false, bool GW = false, int FN = 4, bool F16 = false>
__global__ __launch_bounds__(512, 2) void diffnet_layer_kernel(const DnLayerP p) {
  constexpr int NWN = 16 / FN, NWM = 8 / NWN;  // the wave grid
  constexpr int HN = FN / 2;                   // 32-channel groups [16 gate | 16 filter] per wave and half
  constexpr int BM = 16 * FM * NWM;            // rows per block: NWM wave rows x FM MFMA tiles of 16
  static_assert(FN == 2 || FN == 4, "wave grid");
  // COND: the conditioner projection (1 x 1, 256 -> 2C) joins pass A as four more 64-channel chunks with one tap each
  // (8 more pairs of steps); the gate epilogue then has no conditioner slice to read
  constexpr int NA = COND ? 32 : 24;        // pairs of steps of pass A
  constexpr int SB = 2 * NA;                // first stage of pass B
  constexpr int NSTEPS = SB + 16;
  constexpr int LASTC = COND ? 7 : 3;       // last x chunk
  static_assert(NS >= 3 && NS <= 6, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the ONLY LDS object
  uint4* Ring = reinterpret_cast<uint4*>(smem);                // [NS][1024]  (GW: no ring)
  uint4* G = Ring + (GW ? 0 : NS) * DN_STAGE_U4;               // g [BM][32 chunks of 16 bytes]; first the x windows [2][xrows][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.nMT, mt = blockIdx.x - b * p.nMT;
  const int t0 = mt * BM;
  const int dil = p.dil, T = p.T;
  const int xrows = (BM + 2 * dil + 7) & ~7;
  const int np = xrows >> 3;
  const int len_raw = p.lengths ? p.lengths[b] : T;

  const bf16_raw* yb = p.yin + (int64_t)b * T * DN_C;
  const bf16_raw* cxb = COND ? p.condx + (int64_t)b * T * p.ldcx : nullptr;
  const char* wsrc = reinterpret_cast<const char*>(p.wstream) + wave * 2048 + lane * 16;
  const uint32_t ring_lds = lds_addr(Ring) + (uint32_t)wave * 2048u;
  const uint32_t xs_lds = lds_addr(G);
  const char* zero = reinterpret_cast<const char*>(g_conv_zero_page) + lane * 16;

  auto issue_w = [&](int s, int slot) {
    if constexpr (DBG & 4) return;
    const char* src = wsrc + (size_t)s * (DN_STAGE_U4 * 16);
    const uint32_t dst = ring_lds + (uint32_t)slot * (DN_STAGE_U4 * 16);
    glds16(src, __builtin_amdgcn_readfirstlane(dst));
    glds16(src + 1024, __builtin_amdgcn_readfirstlane(dst + 1024u));
  };
  auto issue_x_piece = [&](int ci, int piece) {  // 8 rows x 128 bytes of the window of chunk ci
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz<8>(r);
    const int ts = t0 - dil + r;
    const bf16_raw* row = (COND && ci >= 4) ? cxb + (int64_t)ts * p.ldcx + (ci - 4) * 64 : yb + (int64_t)ts * DN_C + ci * 64;
    const char* src = (ts >= 0 && ts < T) ? reinterpret_cast<const char*>(row + c * 8) : zero;
    glds16(src, __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)(((ci & 1) * xrows + piece * 8) * 128)));
  };

  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (DBG & 1) stamp[0] = wall_clock64();
  f32x4 acc[2][FM][FN];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: the ring fills while the scalar load of lengths[b] is on its way
  if constexpr (!GW) issue_w(0, 0);
  const int len = min(len_raw, T);
  const bool masked = p.lengths != nullptr;
  // a tile past the utterance's end (token-bucket batches are padded): every output is masked, no MFMA work
  const bool active = !(masked && t0 >= len);

  // Software pipeline (one step deep, through registers): at the top of step s the wave waits for ITS pieces of stage s + 1
  // and passes the barrier (-> everyone's pieces of stage s + 1 have landed, everyone is done with stage s - 1), issues the
  // DMA of stage s + NS - 1 into the slot of stage s - 1, requests the fragments of step s + 1 from LDS and only then
  // runs the 16 MFMAs of step s on the fragments requested a step earlier: the LDS round trip (all 8 waves read at once
  // after a barrier: ~0.2 us, as long as the MFMAs themselves) hides under the matrix work instead of preceding it.
  uint4 wf0[FN] = {}, wf1[FN] = {}, xa[FM] = {}, xb_[FM] = {};
  auto ld_w = [&](uint4 (&wf)[FN], int slot) {
    if constexpr ((DBG & 64) != 0) return;  // (timing experiment: no fragment reads)
    const uint4* Wst = Ring + slot * DN_STAGE_U4;
    if constexpr ((DBG & 128) != 0) {  // (timing experiment: the x window's 128-byte-row pattern on the ring memory)
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int q = (wn & 1) * 64 + fn * 16 + lr;
        wf[fn] = Wst[q * 8 + (lg ^ swz<8>(q))];
      }
      return;
    }
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      const int q = wn * (16 * FN) + fn * 16 + lr;
      wf[fn] = Wst[q * 4 + (lg ^ swz<4>(q))];
    }
  };
  // one LDS address per request: the tiles of a wave are 16 rows apart, which leaves both swizzles unchanged, so tile fm
  // is a compile-time offset (the address arithmetic per step was as long as the MFMA issue itself)
  const int xrow0 = wm * (16 * FM) + lr;
  auto ld_x = [&](uint4 (&xf)[FM], int pair) {  // pair = (ci * 3 + tap) * 2 + kh; conditioner chunks: 24 + (ci - 4) * 2 + kh, centre tap
    int ci = pair / 6, tk = pair - ci * 6;
    int tap = tk >> 1, kh = tk & 1;
    if (COND && pair >= 24) {
      ci = 4 + ((pair - 24) >> 1);
      tap = 1;
      kh = pair & 1;
    }
    if constexpr ((DBG & 64) != 0) return;
    const int r = xrow0 + tap * dil;
    const uint4* src = G + (ci & 1) * xrows * 8 + r * 8 + ((kh * 4 + lg) ^ swz<8>(r));
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) xf[fm] = src[fm * 128];
  };
  const uint4* gsrc0 = G + xrow0 * 32;
  auto ld_g = [&](uint4 (&gf)[FM], int kc) {
    if constexpr ((DBG & 64) != 0) return;
    const uint4* src = gsrc0 + ((kc * 4 + lg) ^ (xrow0 & 15));
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) gf[fm] = src[fm * 512];
  };
  // DBG bit 4: cycle stamps (s_memtime) inside steps 16..19 of waves 0 and 5, parked in LDS behind the g region
  unsigned long long* ST = reinterpret_cast<unsigned long long*>(G + BM * 32);
  auto fine_stamp = [&](int s, int k) {
    if constexpr ((DBG & 16) != 0) {
      if (s >= 16 && s < 20 && (wave == 0 || wave == 5) && lane == 0) ST[((wave != 0) * 4 + (s - 16)) * 8 + k] = __builtin_readcyclecounter();
    }
  };

  // top of step s: stage s + 1 has landed for everyone, stage s - 1 is free; then the next DMA
  auto step_top = [&](int s, int slot) {
    fine_stamp(s, 0);
    if (s + 1 < NSTEPS) dn_wait_stage<NS, NSTEPS>(s);
    fine_stamp(s, 1);
    if constexpr ((DBG & 32) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (timing experiment: no barrier, results invalid)
    else lds_barrier();
    fine_stamp(s, 2);
    if (s + NS - 1 < NSTEPS) issue_w(s + NS - 1, slot == 0 ? NS - 1 : slot - 1);
    fine_stamp(s, 3);
  };
  auto next_slot = [&](int slot) { return slot + 1 == NS ? 0 : slot + 1; };
  int slot = 0;  // slot of the CURRENT step's stage
  // ---- GW: weight fragments from global memory.  Lane (lr, lg) of wave (wm, wn) reads, for fragment fn of stage s, the 16 bytes
  // at stage + ((wn * 64 + fn * 16 + lr) * 4 + (lg ^ swz<4>(row))) * 16 -- swz<4> does not change with fn * 16 or wn * 64, so the
  // four fragments are ONE per-lane offset plus 1 KiB each.
  typedef __attribute__((ext_vector_type(4))) uint32_t dn_u32x4;
  constexpr bool XPF = (FN == 2 && !(COND && FM == 8)) || FM < 4;  // a second x-fragment register set (prefetch one pair ahead) where it fits
  dn_u32x4 W[3][FN];
  const uint32_t gw_off = (uint32_t)(((wn * (16 * FN) + lr) * 4 + (lg ^ swz<4>(wn * (16 * FN) + lr))) * 16);
  const char* gw_base = reinterpret_cast<const char*>(p.wstream);
  auto ldg = [&](dn_u32x4 (&w)[FN], int s) __attribute__((always_inline)) {
    const char* sb = gw_base + (size_t)s * (DN_STAGE_U4 * 16);
    if constexpr (FN == 4) {
      asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                   "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                   "global_load_dwordx4 %3, %4, %5 offset:3072"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                   : "v"(gw_off), "s"(sb)
                   : "memory");
    } else {
      asm volatile("global_load_dwordx4 %0, %2, %3\n\t"
                   "global_load_dwordx4 %1, %2, %3 offset:1024"
                   : "=&v"(w[0]), "=&v"(w[1])
                   : "v"(gw_off), "s"(sb)
                   : "memory");
    }
  };
  auto gw_mfma = [&](f32x4 (&a)[FM][FN], const dn_u32x4 (&w)[FN], const uint4 (&xf)[FM]) __attribute__((always_inline)) {
    if constexpr ((DBG & 2) != 0) {  // (timing experiment: the loads stay, the matrix work goes)
#pragma unroll
      for (int i = 0; i < FN; ++i) asm volatile("" ::"v"(w[i]), "v"(xf[i % FM].x), "v"(xf[i % FM].w));
      return;
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
        a[fm][fn] = DnCv<F16>::mfma(w[fn], xf[fm], a[fm][fn]);
  };
  // wait until the weights of step s have landed: everything issued after them may stay in flight -- the two younger weight
  // groups (if they exist) and the NPW x-window pieces issued in between (loads retire in order).  EVERY wave issues exactly
  // NPW pieces per window (a wave without a piece of its own repeats the window's last piece: same bytes to the same place),
  // so the counts are compile-time constants.
  constexpr int NPW = (BM + 2 * 8 + 7) / 8 > 16 ? 3 : 2;  // 8 waves x NPW >= the pieces of the widest window
  auto gw_wait = [&](int younger_w, bool pieces) __attribute__((always_inline)) {
    const int n = younger_w * FN + (pieces ? NPW : 0);
    if (n >= 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // x-window pieces with one lane offset: piece q holds rows 8 q .. 8 q + 7 of the window; lane (l3 = lane / 8, l7 = lane % 8)
  // moves the 16 bytes of row l3, chunk column l7 ^ swz<8>(8 q + l3) = l7 ^ (l3 / 2) ^ 4 (q & 1)
  const uint32_t gx_voff = (uint32_t)((lane >> 3) * (DN_C * 2) + (((lane & 7) ^ (lane >> 4)) << 4));
  auto gw_piece = [&](int ci, int k) __attribute__((always_inline)) {
    const int q = min(wave + 8 * k, np - 1);
    const int ts0 = t0 - dil + q * 8;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)(((ci & 1) * xrows + q * 8) * 128));
    const bf16_raw* base = (COND && ci >= 4) ? cxb + (int64_t)ts0 * p.ldcx + (ci - 4) * 64 : yb + (int64_t)ts0 * DN_C + ci * 64;
    const uint32_t rowb = (COND && ci >= 4) ? (uint32_t)p.ldcx * 2u : (uint32_t)(DN_C * 2);
    uint32_t voff = gx_voff ^ (uint32_t)((q & 1) << 6);
    if (COND && ci >= 4) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // (opaque: the conditioner windows' lane offsets are not kept across the trip)
      voff = (uint32_t)(lane_o >> 3) * rowb + (uint32_t)((((lane_o & 7) ^ (lane_o >> 4)) << 4) ^ ((q & 1) << 6));
    }
    if (ts0 >= 0 && ts0 + 7 < T) {
      glds16_s(base, voff, dst);
    } else {  // a window edge: rows outside the utterance come from the zero page (still exactly one DMA)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int ts = ts0 + (lane_o >> 3);
      const char* zero_o = reinterpret_cast<const char*>(g_conv_zero_page) + lane_o * 16;
      const char* src = (ts >= 0 && ts < T) ? reinterpret_cast<const char*>(base) + voff : zero_o;
      glds16(src, dst);
    }
  };
  // x fragments with three lane addresses (one per tap): kh flips bit 2 of the swizzled chunk column = 64 bytes
  int gx_idx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int r = xrow0 + k * dil;
    gx_idx[k] = r * 8 + (lg ^ swz<8>(r));
  }
  auto gw_ld_x = [&](uint4 (&xf)[FM], int par, int tap, int kh) __attribute__((always_inline)) {
    int idx = gx_idx[tap];
    if constexpr (COND && FM == 8) {  // (no register to spare for the three tap addresses: recomputed where they are used)
      int lr_o = xrow0;
      asm volatile("" : "+v"(lr_o));
      const int r = lr_o + tap * dil;
      idx = r * 8 + (lg ^ swz<8>(r));
    }
    if (kh) asm volatile("v_xor_b32 %0, 4, %1" : "=v"(idx) : "v"(idx));  // (volatile: not hoisted into six loop invariants)
    const uint4* src = G + par * xrows * 8 + idx;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) xf[fm] = src[fm * 128];
  };
  const int gg_idx = xrow0 * 32 + (lg ^ (xrow0 & 15));
  auto gw_ld_g = [&](uint4 (&gf)[FM], int kc) __attribute__((always_inline)) {
    int idx = gg_idx;
    if (kc) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(idx) : "s"(kc * 4), "v"(gg_idx));
    const uint4* src = G + idx;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) gf[fm] = src[fm * 512];
  };
  if constexpr (GW) {
    if (active) {
#pragma unroll
      for (int k = 0; k < NPW; ++k) gw_piece(0, k);
      ldg(W[0], 0);
      ldg(W[1], 1);
      gw_wait(2, false);  // the window pieces are older than the two weight groups
      lds_barrier();
      if constexpr (DBG & 1) stamp[1] = wall_clock64();
      gw_ld_x(xa, 0, 0, 0);
      // One trip = one 64-channel chunk: 12 straight-line steps i = ((tap * 2 + kh) * 2 + nh); the register set of step s is
      // s % 3 (a trip starts at a multiple of 3).  The next chunk's window pieces go out at the trip's first step, between the
      // weight groups of s0 + 1 and s0 + 2: they are younger than the weights steps s0 and s0 + 1 wait for, older afterwards.
      auto chunk = [&](int ci, int s0, auto more_c) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_c)::value;  // another window follows this chunk
        const int par = ci & 1;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int s = s0 + i;
          const int tk = i >> 1;  // tap * 2 + kh
          if (i == 0 && MORE) {
#pragma unroll
            for (int k = 0; k < NPW; ++k) gw_piece(ci + 1, k);
          }
          if (s + 2 < SB) ldg(W[(i + 2) % 3], s + 2);
          if constexpr (XPF) {  // the next pair's x fragments, one pair ahead (a second register set)
            if (!(i & 1) && i + 2 < 12) {
              if (tk & 1) gw_ld_x(xa, par, (tk + 1) >> 1, (tk + 1) & 1);
              else gw_ld_x(xb_, par, (tk + 1) >> 1, (tk + 1) & 1);
            }
          } else {              // 128-row tiles: no registers for a second set -- the pair's fragments right before its first step
            if (!(i & 1) && i > 0) gw_ld_x(xa, par, tk >> 1, tk & 1);
          }
          const int yw = (s + 2 < SB ? 1 : 0) + (s + 1 < SB ? 1 : 0);
          gw_wait(yw, i <= 1 && MORE);
          __builtin_amdgcn_sched_barrier(0);
          if (XPF && (tk & 1)) gw_mfma(acc[i & 1], W[i % 3], xb_);
          else gw_mfma(acc[i & 1], W[i % 3], xa);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
#pragma unroll 1
      for (int ci = 0; ci < (COND ? 4 : 3); ++ci) {
        chunk(ci, ci * 12, std::true_type{});
        // the next window: this wave's pieces landed steps ago (the in-order waits above), now everyone's
        lds_barrier();
        gw_ld_x(xa, (ci + 1) & 1, COND && ci == 3 ? 1 : 0, 0);
      }
      if constexpr (!COND) chunk(3, 36, std::false_type{});
      if constexpr (COND) {
        // the four conditioner chunks (centre tap only: 4 steps each) as ONE straight-line trip of 16 steps, s0 = 48 = 0 mod 3
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int par = cc & 1;  // chunk 4 + cc
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = cc * 4 + j;
            const int s = 48 + i;
            if (j == 0 && cc < 3) {
#pragma unroll
              for (int k = 0; k < NPW; ++k) gw_piece(4 + cc + 1, k);
            }
            if (s + 2 < SB) ldg(W[(i + 2) % 3], s + 2);
            if (XPF && j == 0) gw_ld_x(xb_, par, 1, 1);
            if (!XPF && j == 2) gw_ld_x(xa, par, 1, 1);
            const int yw = (s + 2 < SB ? 1 : 0) + (s + 1 < SB ? 1 : 0);
            gw_wait(yw, j <= 1 && cc < 3);
            __builtin_amdgcn_sched_barrier(0);
            if (XPF && (j >> 1)) gw_mfma(acc[j & 1], W[i % 3], xb_);
            else gw_mfma(acc[j & 1], W[i % 3], xa);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (cc < 3) {
            lds_barrier();
            gw_ld_x(xa, (cc + 1) & 1, 1, 0);
          }
        }
      }
    } else {
      glds_wait<0>();
    }
  } else
  if (active) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (wave + 8 * k < np) issue_x_piece(0, wave + 8 * k);
#pragma unroll
    for (int s = 1; s <= NS - 3; ++s) issue_w(s, s);
    // pseudo-step -1: stage 0 and the first x window have landed; the first fragments
    dn_wait_stage<NS, NSTEPS>(-1);
    lds_barrier();
    issue_w(NS - 2, NS - 2);
    if constexpr (DBG & 1) stamp[1] = wall_clock64();
    ld_w(wf0, 0);
    ld_x(xa, 0);

    // ---- pass A: dilated conv, both channel halves (24 pairs of steps; two pairs per trip so that the x fragments
    // alternate between two register sets without copies)
    auto pair_steps = [&](int pr, uint4 (&xcur)[FM], uint4 (&xnext)[FM]) __attribute__((always_inline)) {
      const int s = 2 * pr;
      int ci = pr / 6, tk = pr - ci * 6;
      if (COND && pr >= 24) {
        ci = 4 + ((pr - 24) >> 1);
        tk = pr & 1;
      }
      // step s: channel half 0
      step_top(s, slot);
      if (ci < LASTC && tk < 2) {  // the next chunk's window: pieces wave, wave + 8, wave + 16 over three steps
        const int piece = wave + 16 * tk;
        if (piece < np) issue_x_piece(ci + 1, piece);
      }
      ld_w(wf1, next_slot(slot));
      fine_stamp(s, 4);
      __builtin_amdgcn_sched_barrier(0);  // the LDS requests above are issued BEFORE the matrix work ...
      dn_mfma_step<(DBG & 2) != 0, FM, FN, F16>(acc[0], wf0, xcur);
      __builtin_amdgcn_sched_barrier(0);  // ... which stays on this side of the next barrier (register-only code moves across asm)
      fine_stamp(s, 5);
      slot = next_slot(slot);
      // step s + 1: channel half 1 (same x fragments)
      step_top(s + 1, slot);
      if (ci < LASTC && tk == 0) {
        const int piece = wave + 8;
        if (piece < np) issue_x_piece(ci + 1, piece);
      }
      if (pr < NA - 1) {
        ld_w(wf0, next_slot(slot));
        ld_x(xnext, pr + 1);
      }
      fine_stamp(s + 1, 4);
      __builtin_amdgcn_sched_barrier(0);
      dn_mfma_step<(DBG & 2) != 0, FM, FN, F16>(acc[1], wf1, xcur);
      __builtin_amdgcn_sched_barrier(0);
      fine_stamp(s + 1, 5);
      slot = next_slot(slot);
    };
#pragma unroll 1
    for (int pr = 0; pr < NA; pr += 2) {
      pair_steps(pr, xa, xb_);
      pair_steps(pr + 1, xb_, xa);
    }
  } else {
    glds_wait<0>();
  }

  // ---- gate epilogue: every wave is done with the x windows, their memory becomes g
  lds_barrier();
  if constexpr (DBG & 1) stamp[2] = wall_clock64();
  {
    const bf16_raw* cb = p.cond + (int64_t)b * T * p.ldc;
    bf16_raw* ab = SAVE ? p.a_out + (int64_t)b * T * (2 * DN_C) : nullptr;
    uint2* G2 = reinterpret_cast<uint2*>(G);
    // (the lane coordinates pass through an opaque copy: the epilogue's addresses are then computed HERE, not hoisted above
    // pass A where they would sit in registers the K loop needs)
    int lr_e = lr, lg_e = lg;
    asm volatile("" : "+v"(lr_e), "+v"(lg_e));
    // the conditioner vectors are requested two phases (nh, h) ahead of their use: two memory round trips in flight at any time,
    // not sixteen serial ones -- and not all four phases at once either (64 registers the K loops need at 128-row tiles)
    uint4 cv[2][HN][FM];
    auto cv_load = [&](int nh, int h) __attribute__((always_inline)) {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const int t = t0 + wm * (16 * FM) + fm * 16 + lr_e;
        cv[nh][h][fm] = make_uint4(0, 0, 0, 0);
        if (!COND && t < T && !(DBG & 8)) cv[nh][h][fm] = *reinterpret_cast<const uint4*>(cb + (int64_t)t * p.ldc + nh * 256 + wn * (16 * FN) + h * 32 + lg_e * 8);
      }
    };
#pragma unroll
    for (int h = 0; h < HN; ++h) cv_load(0, h);
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
      for (int h = 0; h < HN; ++h) {
        const int pch = nh * 256 + wn * (16 * FN) + h * 32 + lg_e * 8;  // 8 packed channels: [4 gate | their 4 filter partners]
        const int gch = pch >> 1;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.dil_b + pch), b1 = *reinterpret_cast<const f32x4*>(p.dil_b + pch + 4);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          const int row = wm * (16 * FM) + fm * 16 + lr_e;
          const int t = t0 + row;
          const bool valid = t < T;
          const bool keep = !(masked && t >= len);
          f32x4 v0 = acc[nh][fm][2 * h], v1 = acc[nh][fm][2 * h + 1];
          v0 += b0;
          v1 += b1;
          if (!keep) {
            v0 = f32x4{0.f, 0.f, 0.f, 0.f};
            v1 = v0;
          }
          const uint4 r = cv[nh][h][fm];
          v0[0] += DnCv<F16>::lo(r.x); v0[1] += DnCv<F16>::hi(r.x); v0[2] += DnCv<F16>::lo(r.y); v0[3] += DnCv<F16>::hi(r.y);
          v1[0] += DnCv<F16>::lo(r.z); v1[1] += DnCv<F16>::hi(r.z); v1[2] += DnCv<F16>::lo(r.w); v1[3] += DnCv<F16>::hi(r.w);
          uint2 o;
          if constexpr (SAVE) {
            // training: the pre-activation is kept (bf16) and the gate is computed FROM THE ROUNDED values, as gate_fwd does
            const uint2 as = make_uint2(DnCv<F16>::pack(v0[0], v0[1]), DnCv<F16>::pack(v0[2], v0[3]));
            const uint2 af = make_uint2(DnCv<F16>::pack(v1[0], v1[1]), DnCv<F16>::pack(v1[2], v1[3]));
            if (valid && !(DBG & 8)) {
              *reinterpret_cast<uint2*>(ab + (int64_t)t * (2 * DN_C) + gch) = as;
              *reinterpret_cast<uint2*>(ab + (int64_t)t * (2 * DN_C) + DN_C + gch) = af;
            }
            const float sr[4] = {DnCv<F16>::lo(as.x), DnCv<F16>::hi(as.x), DnCv<F16>::lo(as.y), DnCv<F16>::hi(as.y)};
            const float fr[4] = {DnCv<F16>::lo(af.x), DnCv<F16>::hi(af.x), DnCv<F16>::lo(af.y), DnCv<F16>::hi(af.y)};
            o.x = DnCv<F16>::pack(gate_fast(sr[0], fr[0]), gate_fast(sr[1], fr[1]));
            o.y = DnCv<F16>::pack(gate_fast(sr[2], fr[2]), gate_fast(sr[3], fr[3]));
          } else {
            float gte[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gte[e] = keep ? gate_fast(v0[e], v1[e]) : 0.f;
            o.x = DnCv<F16>::pack(gte[0], gte[1]);
            o.y = DnCv<F16>::pack(gte[2], gte[3]);
          }
          G2[(row * 32 + ((gch >> 3) ^ (row & 15))) * 2 + ((gch >> 2) & 1)] = o;
        }
        if (nh == 0) {
          if constexpr (!COND) __builtin_amdgcn_sched_barrier(0);
          cv_load(1, h);
          if constexpr (!COND) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  lds_barrier();
  if constexpr (DBG & 1) stamp[3] = wall_clock64();

  // ---- pass B: output projection (k32 step kc, channel half nh), accumulators reused
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (GW) {
    if (active) {  // 16 straight-line steps i = kc * 2 + nh; no barrier: g is read-only now
      constexpr int PB = SB % 3;
      ldg(W[PB], SB);
      ldg(W[(PB + 1) % 3], SB + 1);
      gw_ld_g(xa, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + 2 < 16) ldg(W[(PB + i + 2) % 3], SB + i + 2);
        if constexpr (XPF) {
          if (!(i & 1) && i + 2 < 16) {
            if ((i >> 1) & 1) gw_ld_g(xa, (i >> 1) + 1);
            else gw_ld_g(xb_, (i >> 1) + 1);
          }
        } else {
          if (!(i & 1) && i > 0) gw_ld_g(xa, i >> 1);
        }
        gw_wait((i + 2 < 16 ? 1 : 0) + (i + 1 < 16 ? 1 : 0), false);
        __builtin_amdgcn_sched_barrier(0);
        if (XPF && ((i >> 1) & 1)) gw_mfma(acc[i & 1], W[(PB + i) % 3], xb_);
        else gw_mfma(acc[i & 1], W[(PB + i) % 3], xa);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else
  if (active) {
    // (stage 48 landed and became visible at the top of step 47; g became visible at the barrier above)
    ld_w(wf0, slot);
    ld_g(xa, 0);
    auto proj_steps = [&](int kc, uint4 (&gcur)[FM], uint4 (&gnext)[FM]) __attribute__((always_inline)) {
      const int s = SB + 2 * kc;
      step_top(s, slot);
      ld_w(wf1, next_slot(slot));
      __builtin_amdgcn_sched_barrier(0);
      dn_mfma_step<(DBG & 2) != 0, FM, FN, F16>(acc[0], wf0, gcur);
      __builtin_amdgcn_sched_barrier(0);
      slot = next_slot(slot);
      step_top(s + 1, slot);
      if (kc < 7) {
        ld_w(wf0, next_slot(slot));
        ld_g(gnext, kc + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      dn_mfma_step<(DBG & 2) != 0, FM, FN, F16>(acc[1], wf1, gcur);
      __builtin_amdgcn_sched_barrier(0);
      slot = next_slot(slot);
    };
#pragma unroll 1
    for (int kc = 0; kc < 8; kc += 2) {
      proj_steps(kc, xa, xb_);
      proj_steps(kc + 1, xb_, xa);
    }
  }

  if constexpr (DBG & 1) stamp[4] = wall_clock64();
  // ---- tail: residual half -> xn, yin'; skip half -> skip (f32); o is rounded to bf16 first, as the two-kernel path stores it
  {
    const float r2 = 0.70710678118654752f;
    // (opaque lane coordinates, as in the gate epilogue: the bias / next-step vectors and the row addresses are computed and
    //  loaded HERE, not hoisted above the K loops where they would hold ~30 registers)
    int lr_t = lr, lg_t = lg;
    asm volatile("" : "+v"(lr_t), "+v"(lg_t));
    const bf16_raw* xb = p.x + (int64_t)b * T * DN_C;
    bf16_raw* xnb = p.xn + (int64_t)b * T * DN_C;
    bf16_raw* yib = p.yin_next ? p.yin_next + (int64_t)b * T * DN_C : nullptr;
    float* skb = p.skip + (int64_t)b * T * DN_C;
    // biases / next step projection of the wave's 2 x 8 channels per lane
    f32x4 bo[HN][2], bs[HN][2], dn[HN][2];
#pragma unroll
    for (int h = 0; h < HN; ++h) {
      const int ch = wn * (16 * FN) + h * 32 + lg_t * 8;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bo[h][u] = *reinterpret_cast<const f32x4*>(p.out_b + ch + 4 * u);
        bs[h][u] = *reinterpret_cast<const f32x4*>(p.out_b + DN_C + ch + 4 * u);
        dn[h][u] = p.dnext ? *reinterpret_cast<const f32x4*>(p.dnext + (int64_t)b * DN_C + ch + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // Two row tiles (fm) at a time, BOTH channel halves of a row together: the wave's 64 channels of a row are one 128-byte
    // line of x / xn / yin' (two 16-byte pieces per lane, h = 0 and 1) and two lines of skip -- issued back to back they
    // meet in L1 / the L2 write combiner instead of being half-line accesses microseconds apart.  Twelve vectors per lane
    // are in flight per round (all sixteen + the accumulators spilled).
#pragma unroll
    for (int f0 = 0; f0 < FM; f0 += 2) {
      uint4 xr[2][HN];
      f32x4 sk[2][HN][2];
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        if (f0 + df >= FM) continue;  // (FM = 3: the last round has one tile)
        const int t = t0 + wm * (16 * FM) + (f0 + df) * 16 + lr_t;
        const bool in = t < T && !((DBG & 8) && t > 0);
#pragma unroll
        for (int h = 0; h < HN; ++h) {
          const int ch = wn * (16 * FN) + h * 32 + lg_t * 8;
          xr[df][h] = make_uint4(0, 0, 0, 0);
          sk[df][h][0] = sk[df][h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (in) {
            xr[df][h] = *reinterpret_cast<const uint4*>(xb + (int64_t)t * DN_C + ch);
            if (!p.init) {
              sk[df][h][0] = *reinterpret_cast<const f32x4*>(skb + (int64_t)t * DN_C + ch);
              sk[df][h][1] = *reinterpret_cast<const f32x4*>(skb + (int64_t)t * DN_C + ch + 4);
            }
          }
        }
      }
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const int fm = f0 + df < FM ? f0 + df : FM - 1;
        const int t = t0 + wm * (16 * FM) + fm * 16 + lr_t;
        if (f0 + df >= FM || t >= T || ((DBG & 8) && t > 0)) continue;
        const bool keep = !(masked && t >= len);
#pragma unroll
        for (int h = 0; h < HN; ++h) {
          const int ch = wn * (16 * FN) + h * 32 + lg_t * 8;
          const uint4 xv4 = xr[df][h];
          const float xv[8] = {DnCv<F16>::lo(xv4.x), DnCv<F16>::hi(xv4.x), DnCv<F16>::lo(xv4.y), DnCv<F16>::hi(xv4.y), DnCv<F16>::lo(xv4.z), DnCv<F16>::hi(xv4.z), DnCv<F16>::lo(xv4.w), DnCv<F16>::hi(xv4.w)};
          float xn[8], yi[8];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {  // (o rounded to bf16 in pairs: one conversion instruction per two values)
            const uint32_t ob = DnCv<F16>::pack(acc[0][fm][2 * h + (e >> 2)][e & 3] + bo[h][e >> 2][e & 3],
                                            acc[0][fm][2 * h + (e >> 2)][(e & 3) + 1] + bo[h][e >> 2][(e & 3) + 1]);
            const float o0 = keep ? DnCv<F16>::lo(ob) : 0.f, o1 = keep ? DnCv<F16>::hi(ob) : 0.f;
            xn[e] = (xv[e] + o0) * r2;
            xn[e + 1] = (xv[e + 1] + o1) * r2;
            yi[e] = xn[e] + dn[h][e >> 2][e & 3];
            yi[e + 1] = xn[e + 1] + dn[h][e >> 2][(e & 3) + 1];
          }
          *reinterpret_cast<uint4*>(xnb + (int64_t)t * DN_C + ch) =
              make_uint4(DnCv<F16>::pack(xn[0], xn[1]), DnCv<F16>::pack(xn[2], xn[3]), DnCv<F16>::pack(xn[4], xn[5]), DnCv<F16>::pack(xn[6], xn[7]));
          if (yib)
            *reinterpret_cast<uint4*>(yib + (int64_t)t * DN_C + ch) =
                make_uint4(DnCv<F16>::pack(yi[0], yi[1]), DnCv<F16>::pack(yi[2], yi[3]), DnCv<F16>::pack(yi[4], yi[5]), DnCv<F16>::pack(yi[6], yi[7]));
        }
#pragma unroll
        for (int h = 0; h < HN; ++h) {
          float* sp = skb + (int64_t)t * DN_C + wn * (16 * FN) + h * 32 + lg_t * 8;
          uint32_t sb[4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            f32x4 sv = sk[df][h][u];
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const uint32_t ob = DnCv<F16>::pack(acc[1][fm][2 * h + u][e] + bs[h][u][e], acc[1][fm][2 * h + u][e + 1] + bs[h][u][e + 1]);
              sv[e] = (keep ? DnCv<F16>::lo(ob) : 0.f) + sv[e];
              sv[e + 1] = (keep ? DnCv<F16>::hi(ob) : 0.f) + sv[e + 1];
            }
            *reinterpret_cast<f32x4*>(sp + 4 * u) = sv;
            sb[2 * u] = DnCv<F16>::pack(sv[0] * p.skip_scale, sv[1] * p.skip_scale);
            sb[2 * u + 1] = DnCv<F16>::pack(sv[2] * p.skip_scale, sv[3] * p.skip_scale);
          }
          if (p.skip_scaled)
            *reinterpret_cast<uint4*>(p.skip_scaled + ((int64_t)b * T + t) * DN_C + wn * (16 * FN) + h * 32 + lg_t * 8) = make_uint4(sb[0], sb[1], sb[2], sb[3]);
        }
      }
    }
    if constexpr (SAVE) {  // g for the output projection's weight gradient: the LDS image leaves row-contiguous
      bf16_raw* gb = p.g_out + (int64_t)b * T * DN_C;
#pragma unroll
      for (int i = 0; i < BM / 16; ++i) {
        const int idx = tid + i * 512;
        const int row = idx >> 5, pos = idx & 31;
        const int t = t0 + row;
        if (t < T && !((DBG & 8) && t > 0)) *reinterpret_cast<uint4*>(gb + (int64_t)t * DN_C + (pos ^ (row & 15)) * 8) = G[idx];
      }
    }
  }
  if constexpr (DBG & 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && p.stamps) {
      stamp[5] = wall_clock64();
#pragma unroll
      for (int i = 0; i < 6; ++i) p.stamps[(size_t)blockIdx.x * 6 + i] = stamp[i];
    }
    if constexpr ((DBG & 16) != 0) {
      if (tid < 64 && p.stamps) p.stamps[(size_t)gridDim.x * 6 + (size_t)blockIdx.x * 64 + tid] = ST[tid];
    }
  }
}

// ---- the weight stream ---------------------------------------------------------------------------------------------------
// stage s of layer l = 1024 x 16 bytes; LDS row q (0..255), chunk position c' (0..3) holds k-chunk c = c' ^ swz<4>(q) of
// weight row n(q) (the row permutation of the MFMA tiles, conv1d_common.h wperm):
//   s < 48:  s = ((ci * 3 + tap) * 2 + kh) * 2 + nh  -> dil_wp[nh * 256 + n(q)][tap][ci * 64 + kh * 32 + c * 8 ..]   (mode-2 operand)
//   s >= 48: s - 48 = kc * 2 + nh                    -> out_wp[nh * 256 + n(q)][0][kc * 32 + c * 8 ..]
struct DnPackTab {
  const uint4* dil[32];
  const uint4* out[32];
  const uint4* cond[32];  // the with-conditioner stream (80 stages): [48, 64) = (cc * 2 + kh) * 2 + nh -> cond_wp[nh * 256 + n(q)][cc * 64 + kh * 32 + c * 8 ..]
};
__device__ __forceinline__ int dn_row_of(int q) {
  const int u = q & 63;
  const int tile = u >> 4, lgq = (u >> 2) & 3, r = u & 3;
  return (q - u) + (tile >> 1) * 32 + lgq * 8 + (tile & 1) * 4 + r;
}
__global__ __launch_bounds__(256) void diffnet_pack_wstream_kernel(const DnPackTab tab, uint4* __restrict__ ws, int nsteps) {
  const int l = blockIdx.x / nsteps, s = blockIdx.x - l * nsteps;
  const int sb = nsteps - 16;  // first stage of the output projection
  uint4* dst = ws + ((size_t)l * nsteps + s) * DN_STAGE_U4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int q = idx >> 2, cp = idx & 3;
    const int c = cp ^ swz<4>(q);
    const int n = dn_row_of(q);
    uint4 v;
    if (s < 48) {
      const int nh = s & 1, kh = (s >> 1) & 1, ct = s >> 2;  // ct = ci * 3 + tap
      const int ci = ct / 3, tap = ct - ci * 3;
      // operand [512][3][256] bf16: 32 uint4 per (row, tap)
      v = tab.dil[l][((size_t)(nh * 256 + n) * 3 + tap) * 32 + ci * 8 + kh * 4 + c];
    } else if (s < sb) {
      const int nh = s & 1, kh = (s >> 1) & 1, cc = (s - 48) >> 2;
      v = tab.cond[l][(size_t)(nh * 256 + n) * 32 + cc * 8 + kh * 4 + c];  // operand [512][1][256] bf16, gate-interleaved rows
    } else {
      const int nh = s & 1, kc = (s - sb) >> 1;
      v = tab.out[l][(size_t)(nh * 256 + n) * 32 + kc * 4 + c];
    }
    dst[idx] = v;
  }
}

}  // namespace

// bf16: every form; IEEE half (PTPP_F16): the inference forms on the 1 x 8 wave grid (the sampler; no kept pre-activation, no
// folded conditioner projection)
extern "C" int ptpp_diffnet_layer_supported(int C, int dtype) { return C == DN_C && (dtype == PTPP_BF16 || dtype == PTPP_F16); }

extern "C" int64_t ptpp_diffnet_wstream_bytes(int C) { return C == DN_C ? (int64_t)DN_STEPS * DN_STAGE_U4 * 16 : 0; }
extern "C" int64_t ptpp_diffnet_wstream_bytes_cond(int C) { return C == DN_C ? (int64_t)(DN_STEPS + 16) * DN_STAGE_U4 * 16 : 0; }

static int dn_pack(const void* const* dil_wp, const void* const* cond_wp, const void* const* out_wp, void* wstream, int L, int C, void* stream) {
  PTPP_CHECK_ARG(dil_wp && out_wp && wstream && L > 0, "diffnet_pack_wstream: null pointer / bad layer count");
  PTPP_CHECK_ARG(C == DN_C, "diffnet_pack_wstream: C = %d is not supported (256)", C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nsteps = cond_wp ? DN_STEPS + 16 : DN_STEPS;
  for (int l0 = 0; l0 < L; l0 += 32) {
    DnPackTab tab;
    const int n = L - l0 < 32 ? L - l0 : 32;
    for (int i = 0; i < 32; ++i) {
      tab.dil[i] = reinterpret_cast<const uint4*>(dil_wp[l0 + (i < n ? i : 0)]);
      tab.out[i] = reinterpret_cast<const uint4*>(out_wp[l0 + (i < n ? i : 0)]);
      tab.cond[i] = cond_wp ? reinterpret_cast<const uint4*>(cond_wp[l0 + (i < n ? i : 0)]) : nullptr;
      PTPP_CHECK_ARG(tab.dil[i] && tab.out[i] && (!cond_wp || tab.cond[i]), "diffnet_pack_wstream: null operand");
    }
    uint4* dst = reinterpret_cast<uint4*>(wstream) + (size_t)l0 * nsteps * DN_STAGE_U4;
    hipLaunchKernelGGL(diffnet_pack_wstream_kernel, dim3((unsigned)(n * nsteps), 1), dim3(256), 0, st, tab, dst, nsteps);
  }
  PTPP_CHECK_LAUNCH("diffnet_pack_wstream");
  return PTPP_OK;
}
extern "C" int ptpp_diffnet_pack_wstream(const void* const* dil_wp, const void* const* out_wp, void* wstream, int L, int C, void* stream) {
  return dn_pack(dil_wp, nullptr, out_wp, wstream, L, C, stream);
}
extern "C" int ptpp_diffnet_pack_wstream_cond(const void* const* dil_wp, const void* const* cond_wp, const void* const* out_wp, void* wstream,
                                             int L, int C, void* stream) {
  PTPP_CHECK_ARG(cond_wp, "diffnet_pack_wstream_cond: null pointer");
  return dn_pack(dil_wp, cond_wp, out_wp, wstream, L, C, stream);
}

static int diffnet_layer_launch(const ptpp_diffnet_layer_args* a, int dbg, unsigned long long* stamps, void* stream);

extern "C" int ptpp_diffnet_layer_fwd(const ptpp_diffnet_layer_args* a, void* stream) { return diffnet_layer_launch(a, 0, nullptr, stream); }
// diagnostics (tools/bench_diffnet_layer.py): dbg bit 0 = per-block clock stamps (6 x u64 per block, 100 MHz), bit 1 = no
// MFMAs, bit 2 = no weight stream, bit 3 = no epilogue loads / stores.  Results are only valid for dbg <= 1.
extern "C" int ptpp_diffnet_layer_fwd_dbg(const ptpp_diffnet_layer_args* a, int dbg, void* stamps, void* stream) {
  return diffnet_layer_launch(a, dbg | 1, reinterpret_cast<unsigned long long*>(stamps), stream);
}

static int diffnet_layer_launch(const ptpp_diffnet_layer_args* a, int dbg, unsigned long long* stamps, void* stream) {
  PTPP_CHECK_ARG(a && a->yin && a->x && (a->cond || a->condx) && a->wstream && a->dil_b && a->out_b && a->skip && a->xn,
                 "diffnet_layer_fwd: null pointer");
  PTPP_CHECK_ARG(!a->condx || (a->ldcx >= DN_C && (a->ldcx & 7) == 0 && ((uintptr_t)a->condx & 15) == 0 && !dbg),
                 "diffnet_layer_fwd: bad conditioner input (ldcx %d; 256 channels, 16-byte aligned, no diagnostics build)", a->ldcx);
  PTPP_CHECK_ARG(a->C == DN_C && (a->dtype == PTPP_BF16 || a->dtype == PTPP_F16), "diffnet_layer_fwd: C = 256 and bf16 / f16 only (C %d, dtype %d)",
                 a->C, a->dtype);
  const bool f16 = a->dtype == PTPP_F16;
  PTPP_CHECK_ARG(!f16 || (!a->a_out && !a->condx && !dbg), "diffnet_layer_fwd: f16 is built for the inference form (no kept pre-activation, "
                 "no conditioner input, no diagnostics)");
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && (a->dil == 1 || a->dil == 2 || a->dil == 4 || a->dil == 8) &&
                     (a->condx || (a->ldc >= 2 * DN_C && (a->ldc & 7) == 0)),
                 "diffnet_layer_fwd: bad shape (B %d T %d dil %d ldc %d)", a->B, a->T, a->dil, a->ldc);
  PTPP_CHECK_ARG((a->a_out != nullptr) == (a->g_out != nullptr), "diffnet_layer_fwd: a_out and g_out go together (training) or are both NULL");
  const uintptr_t al = (uintptr_t)a->yin | (uintptr_t)a->x | (uintptr_t)a->cond | (uintptr_t)a->wstream | (uintptr_t)a->skip | (uintptr_t)a->xn |
                       (uintptr_t)a->yin_next | (uintptr_t)a->a_out | (uintptr_t)a->g_out | (uintptr_t)a->dil_b | (uintptr_t)a->out_b |
                       (uintptr_t)a->dnext | (uintptr_t)a->skip_scaled;
  PTPP_CHECK_ARG((al & 15) == 0, "diffnet_layer_fwd: every tensor must be 16-byte aligned");
  DnLayerP p;
  p.yin = reinterpret_cast<const bf16_raw*>(a->yin);
  p.x = reinterpret_cast<const bf16_raw*>(a->x);
  p.cond = reinterpret_cast<const bf16_raw*>(a->cond);
  p.wstream = reinterpret_cast<const uint4*>(a->wstream);
  p.dil_b = a->dil_b;
  p.out_b = a->out_b;
  p.dnext = a->dnext;
  p.skip = a->skip;
  p.xn = reinterpret_cast<bf16_raw*>(a->xn);
  p.yin_next = reinterpret_cast<bf16_raw*>(a->yin_next);
  p.a_out = reinterpret_cast<bf16_raw*>(a->a_out);
  p.g_out = reinterpret_cast<bf16_raw*>(a->g_out);
  p.lengths = a->lengths;
  p.skip_scaled = reinterpret_cast<bf16_raw*>(a->skip_scaled);
  p.skip_scale = a->skip_scale;
  p.condx = reinterpret_cast<const bf16_raw*>(a->condx);
  p.ldcx = a->ldcx;
  p.B = a->B; p.T = a->T; p.dil = a->dil; p.ldc = a->ldc; p.init = a->init;
  // A block holds 96-144 KiB of LDS: one block per CU, a launch runs in rounds of 256 blocks.  Rows per block (128 / 96 / 64):
  // the choice with the least (rounds x time of one block) -- block times measured on the inference form, 57 / 46 / 35 us
  // (tools/bench_diffnet_layer.py); a 30 000-frame training bucket takes 128 (235 blocks), the sampler's 32 x 540 frames 96
  // (192 blocks: 64-row tiles would need two rounds, 128-row tiles leave 96 CUs idle).
  // PTPP_DIFFNET_GW: 0 = LDS ring; 1 = weights straight from global memory, 2 x 4 wave grid; 2 (default) = the same on the
  // 1 x 8 wave grid (FN = 2).  Diagnostics builds exist for the ring form and for dbg 200 + m (the 1 x 8 form, 128 rows).
  const char* gwe = getenv("PTPP_DIFFNET_GW");
  const int gwm = f16 ? 2 : dbg >= 200 ? 2 : dbg ? 0 : (gwe ? atoi(gwe) : 2);
  int bm = 128;
  {
    // the 1 x 8 form also has 80- and 112-row blocks for the inference forms (a wave owns bm / 16 row tiles): block times of the
    // inference form 41.7 / 36.2 / 34.3 us at 128 / 96 / 80 rows and 27.2 us per round at 64 (tools/bench_diffnet_layer.py 32 546), 112
    // rows interpolated -- the sampler's 32 x 546 frames take 80 rows (224 blocks, one round) instead of 96
    const int cand[5] = {128, 112, 96, 80, 64};
    const float tblk_ring[5] = {57.f, 1e9f, 46.f, 1e9f, 35.f};
    const float tblk_gw[5] = {41.7f, 39.5f, 36.2f, 34.3f, 27.2f};
    const bool fine = gwm == 2 && !a->a_out && !dbg;  // (80 / 112 rows: inference instantiations of the 1 x 8 form only)
    float best = 1e30f;
    for (int i = 0; i < 5; ++i) {
      if ((i & 1) && !fine) continue;
      const int64_t nb = (int64_t)a->B * ((a->T + cand[i] - 1) / cand[i]);
      const float cost = (float)((nb + 255) / 256) * (gwm == 2 ? tblk_gw[i] : tblk_ring[i]);
      if (cost < best - 0.5f) { best = cost; bm = cand[i]; }
    }
    if ((dbg & 16) || dbg >= 200) bm = 128;
    const char* force = getenv("PTPP_DIFFNET_BM");  // (experiments / tests)
    if (force) {
      const int f = atoi(force);
      if (f == 64 || f == 96 || f == 128 || (fine && (f == 80 || f == 112))) bm = f;
    }
  }
  const bool small = bm == 64;
  p.nMT = (a->T + bm - 1) / bm;
  constexpr int NS = 5;
  const bool gw = gwm == 1 || gwm == 2;
  const size_t smem = (size_t)((gw ? 0 : NS * DN_STAGE_U4) + bm * 32) * 16 + ((dbg < 200 && (dbg & 16)) ? 1024 : 0);
  p.stamps = stamps;
  const bool sv = a->a_out != nullptr;
  auto kern = sv ? diffnet_layer_kernel<NS, true> : diffnet_layer_kernel<NS, false>;
  if (small) kern = sv ? diffnet_layer_kernel<NS, true, 0, 2> : diffnet_layer_kernel<NS, false, 0, 2>;
  if (bm == 96) kern = sv ? diffnet_layer_kernel<NS, true, 0, 3> : diffnet_layer_kernel<NS, false, 0, 3>;
  if (a->condx) {  // the conditioner projection inside pass A (80-stage operand stream)
    kern = sv ? diffnet_layer_kernel<NS, true, 0, 4, true> : diffnet_layer_kernel<NS, false, 0, 4, true>;
    if (small) kern = sv ? diffnet_layer_kernel<NS, true, 0, 2, true> : diffnet_layer_kernel<NS, false, 0, 2, true>;
    if (bm == 96) kern = sv ? diffnet_layer_kernel<NS, true, 0, 3, true> : diffnet_layer_kernel<NS, false, 0, 3, true>;
  }
  if (gwm == 1) {
    if (a->condx) {
      kern = sv ? diffnet_layer_kernel<NS, true, 0, 4, true, true> : diffnet_layer_kernel<NS, false, 0, 4, true, true>;
      if (small) kern = sv ? diffnet_layer_kernel<NS, true, 0, 2, true, true> : diffnet_layer_kernel<NS, false, 0, 2, true, true>;
      if (bm == 96) kern = sv ? diffnet_layer_kernel<NS, true, 0, 3, true, true> : diffnet_layer_kernel<NS, false, 0, 3, true, true>;
    } else {
      kern = sv ? diffnet_layer_kernel<NS, true, 0, 4, false, true> : diffnet_layer_kernel<NS, false, 0, 4, false, true>;
      if (small) kern = sv ? diffnet_layer_kernel<NS, true, 0, 2, false, true> : diffnet_layer_kernel<NS, false, 0, 2, false, true>;
      if (bm == 96) kern = sv ? diffnet_layer_kernel<NS, true, 0, 3, false, true> : diffnet_layer_kernel<NS, false, 0, 3, false, true>;
    }
  }
  if (gwm == 2) {
    if (a->condx) {
      kern = sv ? diffnet_layer_kernel<NS, true, 0, 8, true, true, 2> : diffnet_layer_kernel<NS, false, 0, 8, true, true, 2>;
      if (small) kern = sv ? diffnet_layer_kernel<NS, true, 0, 4, true, true, 2> : diffnet_layer_kernel<NS, false, 0, 4, true, true, 2>;
      if (bm == 96) kern = sv ? diffnet_layer_kernel<NS, true, 0, 6, true, true, 2> : diffnet_layer_kernel<NS, false, 0, 6, true, true, 2>;
      if (bm == 80) kern = diffnet_layer_kernel<NS, false, 0, 5, true, true, 2>;
      if (bm == 112) kern = diffnet_layer_kernel<NS, false, 0, 7, true, true, 2>;
    } else {
      kern = sv ? diffnet_layer_kernel<NS, true, 0, 8, false, true, 2> : diffnet_layer_kernel<NS, false, 0, 8, false, true, 2>;
      if (small) kern = sv ? diffnet_layer_kernel<NS, true, 0, 4, false, true, 2> : diffnet_layer_kernel<NS, false, 0, 4, false, true, 2>;
      if (bm == 96) kern = sv ? diffnet_layer_kernel<NS, true, 0, 6, false, true, 2> : diffnet_layer_kernel<NS, false, 0, 6, false, true, 2>;
      if (bm == 80) kern = diffnet_layer_kernel<NS, false, 0, 5, false, true, 2>;
      if (bm == 112) kern = diffnet_layer_kernel<NS, false, 0, 7, false, true, 2>;
    }
  }
  if (f16) {
    kern = diffnet_layer_kernel<NS, false, 0, 8, false, true, 2, true>;
    if (small) kern = diffnet_layer_kernel<NS, false, 0, 4, false, true, 2, true>;
    if (bm == 96) kern = diffnet_layer_kernel<NS, false, 0, 6, false, true, 2, true>;
    if (bm == 80) kern = diffnet_layer_kernel<NS, false, 0, 5, false, true, 2, true>;
    if (bm == 112) kern = diffnet_layer_kernel<NS, false, 0, 7, false, true, 2, true>;
  }
  if (bm == 96 && dbg) { ptpp_set_error("diffnet_layer_fwd_dbg: the 96-row instantiation has no diagnostics build"); return PTPP_EINVAL; }
  if (dbg) {
    switch (dbg + (small ? 100 : 0)) {
      case 1: kern = sv ? diffnet_layer_kernel<NS, true, 1> : diffnet_layer_kernel<NS, false, 1>; break;
      case 3: kern = sv ? diffnet_layer_kernel<NS, true, 3> : diffnet_layer_kernel<NS, false, 3>; break;
      case 5: kern = sv ? diffnet_layer_kernel<NS, true, 5> : diffnet_layer_kernel<NS, false, 5>; break;
      case 7: kern = sv ? diffnet_layer_kernel<NS, true, 7> : diffnet_layer_kernel<NS, false, 7>; break;
      case 9: kern = sv ? diffnet_layer_kernel<NS, true, 9> : diffnet_layer_kernel<NS, false, 9>; break;
      case 15: kern = sv ? diffnet_layer_kernel<NS, true, 15> : diffnet_layer_kernel<NS, false, 15>; break;
      case 17: kern = sv ? diffnet_layer_kernel<NS, true, 17> : diffnet_layer_kernel<NS, false, 17>; break;
      case 23: kern = sv ? diffnet_layer_kernel<NS, true, 23> : diffnet_layer_kernel<NS, false, 23>; break;
      case 33: kern = sv ? diffnet_layer_kernel<NS, true, 33> : diffnet_layer_kernel<NS, false, 33>; break;
      case 65: kern = sv ? diffnet_layer_kernel<NS, true, 65> : diffnet_layer_kernel<NS, false, 65>; break;
      case 47: kern = sv ? diffnet_layer_kernel<NS, true, 47> : diffnet_layer_kernel<NS, false, 47>; break;
      case 79: kern = sv ? diffnet_layer_kernel<NS, true, 79> : diffnet_layer_kernel<NS, false, 79>; break;
      case 143: kern = sv ? diffnet_layer_kernel<NS, true, 143> : diffnet_layer_kernel<NS, false, 143>; break;
      case 129: kern = sv ? diffnet_layer_kernel<NS, true, 129> : diffnet_layer_kernel<NS, false, 129>; break;
      case 111: kern = sv ? diffnet_layer_kernel<NS, true, 111> : diffnet_layer_kernel<NS, false, 111>; break;
      case 201: kern = sv ? diffnet_layer_kernel<NS, true, 1, 8, false, true, 2> : diffnet_layer_kernel<NS, false, 1, 8, false, true, 2>; break;
      case 203: kern = sv ? diffnet_layer_kernel<NS, true, 3, 8, false, true, 2> : diffnet_layer_kernel<NS, false, 3, 8, false, true, 2>; break;
      case 209: kern = sv ? diffnet_layer_kernel<NS, true, 9, 8, false, true, 2> : diffnet_layer_kernel<NS, false, 9, 8, false, true, 2>; break;
      case 211: kern = sv ? diffnet_layer_kernel<NS, true, 11, 8, false, true, 2> : diffnet_layer_kernel<NS, false, 11, 8, false, true, 2>; break;
      case 101: kern = sv ? diffnet_layer_kernel<NS, true, 1, 2> : diffnet_layer_kernel<NS, false, 1, 2>; break;
      case 109: kern = sv ? diffnet_layer_kernel<NS, true, 9, 2> : diffnet_layer_kernel<NS, false, 9, 2>; break;
      case 115: kern = sv ? diffnet_layer_kernel<NS, true, 15, 2> : diffnet_layer_kernel<NS, false, 15, 2>; break;
      default: ptpp_set_error("diffnet_layer_fwd_dbg: mode %d is not built", dbg); return PTPP_EINVAL;
    }
  }
  {  // once per (device, kernel): the dynamic LDS size is above the 64 KiB default
    const void* kp = reinterpret_cast<const void*>(kern);
    if (!lds_limit_raised(kp)) {
      const hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        ptpp_set_error("diffnet_layer_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        return PTPP_ELAUNCH;
      }
      lds_limit_mark(kp);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT)), dim3(512), smem, reinterpret_cast<hipStream_t>(stream), p);
  PTPP_CHECK_LAUNCH("diffnet_layer_fwd");
  return PTPP_OK;
}
