// Row-tile Conv1d for the frame-level layers with 256 output channels (bf16): the structure of the one-launch DiffNet layer
// (diffnet_layer.hip) as a general forward / data-gradient convolution
//
//   y = res_scale * res + out_scale * mask_out(act(conv(x_masked) + bias))        act in {none, ReLU}
//
// for the shapes that dominate the training step's conv time: frame prior network (256 -> 256, k = 17; reference
// modules/frame_prior.py:85-89), pitch predictor (k = 5; modules/variance_adaptor.py:31-36), the DiffNet dilated conv's data
// gradient (512 -> 256, k = 3, dilated; modules/denoiser.py:58-64) -- forward operands in pack mode 3, data-gradient operands
// in mode 4.  Bit-identical to ptpp_conv1d_fwd_ex on the same arguments: same K order (64-channel chunk, tap, 32-channel
// MFMA step), same epilogue arithmetic.
//
// A block of 8 waves owns BM = 128 / 96 / 64 rows of one utterance and ALL 256 output channels:
//   * the x window of a 64-channel chunk (BM + (ks - 1) dil rows) is fetched ONCE per block (the tile kernel fetched it once
//     per 128-channel tile) by LDS-DMA, double-buffered;
//   * the weights arrive as ONE contiguous stream of 16 KiB stages ([256 channels][32 k], already the LDS image: pack modes
//     3 / 4) through a ring of NS stages, NS - 3 stages in flight under the MFMAs;
//   * wave tile 64 x 64 (BM = 128): 8 fragment reads per 16 MFMAs, requested one step ahead so the round trip hides under the
//     previous step's MFMAs.  (Round 4 blamed an "LDS -> register return path of ~64 B/clk/CU" for this loop's 33-50 % of peak;
//     round 5's micro-benchmark measures 256 B/clk -- the loop is bound by the ~100 scalar instructions per step of its ring
//     bookkeeping, DESIGN.md section 5f.  conv1d_rt_gw_kernel below is the form built on that finding; this one stays for the
//     tap counts and channel counts it does not cover and as the bit-identity reference.)
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "conv1d_common.h"
#include "lds_dma.h"

namespace {

constexpr int RT_N = 256;
constexpr int RT_STAGE_U4 = 1024;

struct RtP {
  const bf16_raw* x;
  const uint4* wstream;
  const float* bias;
  const bf16_raw* res;
  bf16_raw* y;
  const int* lengths;
  int B, T, Cin, ks, dil, pad, ldx, ldy, ldr;
  int act, in_mask, out_mask;
  float out_scale, res_scale;
  int nMT;
  bf16_raw* aux;  // optional second output: bf16(y * aux_scale), row stride ldaux, rows past an utterance's end zero
  int ldaux;
  float aux_scale;
  // gate-backward epilogue (conv1d_rt_gw_kernel<.., EPI = 1>): saved pre-activation (B, T, 2 N) and the gradient it yields
  const bf16_raw* gate_a;
  bf16_raw* gate_da;
  int ldda;
  // round 6 (ptpp_conv1d_rt_fwd_ex, conv1d_rt_gw_kernel only): output channels in groups of 256 (blockIdx.y; the operand stream of
  // group g starts grp_stride bytes further, bias / y / res / dropout indices move by 256 g), dropout on the conv term, and a
  // split of the Cin trips over blockIdx.z whose raw f32 partial sums go to `ws` (EPI = 2) for conv_splitk_finish_kernel
  int cout_total;
  int64_t grp_stride;
  unsigned drop_thresh16;
  float drop_inv_keep;
  unsigned long long drop_seed;
  float* ws;
  int split_trips;  // trips (two 64-channel chunks) per split; Cin / 128 when there is no split
  // round 6 (conv1d_rt_gw_kernel, plain epilogue): per-tile column sums of the ROUNDED output, (B, ceil(T / 32), 256) f32 -- a tile
  // writes its sum to slot t0 / 32 and zeros to its other slots, so the caller needs no memset and sums the slots in a fixed order
  float* colpart;
  // round 6 (plain epilogue): the ReLU / dropout backward of the NEXT layer towards the input as this launch's output --
  // relu_dz[row][ch] = relu_src[row][ch] > 0 && row inside its utterance ? bf16(y) * relu_inv : 0 (ptpp_epilogue_bwd's arithmetic on
  // the ROUNDED y, which is not stored): the Conformer feed-forward backward's  conv -> mask -> conv  chain without the middle pass
  const bf16_raw* relu_src;
  bf16_raw* relu_dz;
  int ldrs, lddz;
  float relu_inv;
};
inline void rtp_plain(RtP& p) {  // the fields of the forms that came before round 6
  p.cout_total = RT_N; p.grp_stride = 0; p.drop_thresh16 = 0; p.drop_inv_keep = 1.f; p.drop_seed = 0; p.ws = nullptr;
  p.split_trips = p.Cin >> 7; p.colpart = nullptr; p.relu_src = nullptr; p.relu_dz = nullptr; p.ldrs = p.lddz = 0; p.relu_inv = 1.f;
}

__device__ __forceinline__ void rt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int FM>
__device__ __forceinline__ void rt_mfma_step(f32x4 (&acc)[FM][4], const uint4 (&wf)[4], const uint4 (&xf)[FM]) {
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)
      acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[fn]), __builtin_bit_cast(bf16x8_t, xf[fm]),
                                                            acc[fm][fn], 0, 0, 0);
}

template <int NS>
__device__ __forceinline__ void rt_wait_stage(int s, int S) {
  // top of step s: this wave's pieces of stage s + 1 must have landed; stages up to min(S - 1, s + NS - 2) are issued
  const int younger = min(NS - 3, S - 2 - s);
  if (younger >= 3) glds_wait<6>();
  else if (younger == 2) glds_wait<4>();
  else if (younger == 1) glds_wait<2>();
  else glds_wait<0>();
}

template <int NS, int FM, int ACT>
__global__ __launch_bounds__(512, 2) void conv1d_rt_kernel(const RtP p) {
  constexpr int BM = 32 * FM;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the ONLY LDS object
  uint4* Ring = reinterpret_cast<uint4*>(smem);                // [NS][1024]
  uint4* Xw = Ring + NS * RT_STAGE_U4;                         // x windows [2][xrows][8]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.nMT, mt = blockIdx.x - b * p.nMT;
  const int t0 = mt * BM;
  const int T = p.T, ks = p.ks, dil = p.dil;
  const int xrows = (BM + (ks - 1) * dil + 7) & ~7;
  const int np = xrows >> 3;
  const int nC = p.Cin >> 6;
  const int S = nC * ks * 2;      // stages = steps
  const int cs = ks * 2;          // steps per 64-channel chunk
  const int len_raw = p.lengths ? p.lengths[b] : T;

  const bf16_raw* xb = p.x + (int64_t)b * T * p.ldx;
  const char* wsrc = reinterpret_cast<const char*>(p.wstream) + wave * 2048 + lane * 16;
  const uint32_t ring_lds = lds_addr(Ring) + (uint32_t)wave * 2048u;
  const uint32_t xs_lds = lds_addr(Xw);
  const char* zero = reinterpret_cast<const char*>(g_conv_zero_page) + lane * 16;

  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue_w = [&](int s, int slot) {
    const char* src = wsrc + (size_t)s * (RT_STAGE_U4 * 16);
    const uint32_t dst = ring_lds + (uint32_t)slot * (RT_STAGE_U4 * 16);
    glds16(src, __builtin_amdgcn_readfirstlane(dst));
    glds16(src + 1024, __builtin_amdgcn_readfirstlane(dst + 1024u));
  };
  issue_w(0, 0);  // (does not depend on the utterance length: leaves before the scalar load is waited for)
  const int len = min(len_raw, T);
  const int Tin = p.in_mask ? len : T;
  auto issue_x_piece = [&](int ci, int piece) {
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz<8>(r);
    const int ts = t0 - p.pad + r;
    const char* src = (ts >= 0 && ts < Tin) ? reinterpret_cast<const char*>(xb + (int64_t)ts * p.ldx + ci * 64 + c * 8) : zero;
    glds16(src, __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)(((ci & 1) * xrows + piece * 8) * 128)));
  };
  // no K loop for a tile whose output rows are all masked out, or whose whole input window lies past the utterance's end
  // with a masked input (the accumulators stay exactly zero, as in the tile kernel)
  const bool active = !((p.out_mask && t0 >= len) || (p.in_mask && t0 - p.pad >= len));

  uint4 wa[4], wb[4], xa[FM], xb_[FM];
  auto ld_w = [&](uint4 (&wf)[4], int slot) {
    const uint4* Wst = Ring + slot * RT_STAGE_U4;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      const int q = wn * 64 + fn * 16 + lr;
      wf[fn] = Wst[q * 4 + (lg ^ swz<4>(q))];
    }
  };
  const int xrow0 = wm * (16 * FM) + lr;
  auto ld_x = [&](uint4 (&xf)[FM], int ci, int tk) {  // step (ci * ks + tap) * 2 + kh, tk = tap * 2 + kh
    const int tap = tk >> 1, kh = tk & 1;
    const int r = xrow0 + tap * dil;
    const uint4* src = Xw + (ci & 1) * xrows * 8 + r * 8 + ((kh * 4 + lg) ^ swz<8>(r));
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) xf[fm] = src[fm * 128];
  };
  auto next_slot = [&](int slot) { return slot + 1 == NS ? 0 : slot + 1; };

  if (active) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (wave + 8 * k < np) issue_x_piece(0, wave + 8 * k);
#pragma unroll
    for (int s = 1; s <= NS - 3; ++s)
      if (s < S) issue_w(s, s);
    // pseudo-step -1: stage 0 and the first window have landed; the first fragments
    rt_wait_stage<NS>(-1, S);
    rt_barrier();
    if (NS - 2 < S) issue_w(NS - 2, NS - 2);
    ld_w(wa, 0);
    ld_x(xa, 0, 0);
    int slot = 0;
    int ci = 0, j = 0;  // chunk and step-in-chunk of the CURRENT step (kept incrementally: no division in the loop)
    // one step: stage s + 1 landed for everyone (wait + barrier), stage s - 1's slot refilled, the fragments of step s + 1
    // requested, then the 16 * FM / 4 MFMAs of step s on the fragments requested a step earlier
    auto step = [&](int s, uint4 (&wc)[4], uint4 (&wn_)[4], uint4 (&xc)[FM], uint4 (&xn_)[FM]) __attribute__((always_inline)) {
      if (s + 1 < S) rt_wait_stage<NS>(s, S);
      rt_barrier();
      if (s + NS - 1 < S) issue_w(s + NS - 1, slot == 0 ? NS - 1 : slot - 1);
      {  // the next chunk's window: pieces wave, wave + 8 at the chunk's first step, wave + 16 at its second (ks >= 3: they
         // are retired and published two steps before the first read)
        if (ci + 1 < nC && j < 2) {
          if (j == 0) {
            if (wave < np) issue_x_piece(ci + 1, wave);
            if (wave + 8 < np) issue_x_piece(ci + 1, wave + 8);
          } else if (wave + 16 < np) {
            issue_x_piece(ci + 1, wave + 16);
          }
        }
      }
      const int nj = j + 1 == cs ? 0 : j + 1, nci = j + 1 == cs ? ci + 1 : ci;
      if (s + 1 < S) {
        ld_w(wn_, next_slot(slot));
        ld_x(xn_, nci, nj);
      }
      __builtin_amdgcn_sched_barrier(0);
      rt_mfma_step<FM>(acc, wc, xc);
      __builtin_amdgcn_sched_barrier(0);
      slot = next_slot(slot);
      ci = nci;
      j = nj;
    };
#pragma unroll 1
    for (int s = 0; s < S; s += 2) {  // (S is even: two k-halves per tap)
      step(s, wa, wb, xa, xb_);
      step(s + 1, wb, wa, xb_, xa);
    }
  } else {
    glds_wait<0>();
  }

  // ---- epilogue, straight from the accumulators (a lane: 8 consecutive channels of one row per fragment pair); the arithmetic
  // of conv1d_common.h conv_epilogue_act.  Both channel halves of a row together, two row tiles per round.
  {
    bf16_raw* yb = p.y + (int64_t)b * T * p.ldy;
    const bf16_raw* rb = p.res ? p.res + (int64_t)b * T * p.ldr : nullptr;
    const float e_scale = p.out_scale, e_rscale = p.res_scale;
    f32x4 bia[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        bia[h][u] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + wn * 64 + h * 32 + lg * 8 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f0 = 0; f0 < FM; f0 += 2) {
      uint4 rr[2][2];
#pragma unroll
      for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          rr[df][h] = make_uint4(0, 0, 0, 0);
          const int t = t0 + wm * (16 * FM) + (f0 + df) * 16 + lr;
          if (rb && f0 + df < FM && t < T) rr[df][h] = *reinterpret_cast<const uint4*>(rb + (int64_t)t * p.ldr + wn * 64 + h * 32 + lg * 8);
        }
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const int fm = f0 + df < FM ? f0 + df : FM - 1;
        const int t = t0 + wm * (16 * FM) + fm * 16 + lr;
        if (f0 + df >= FM || t >= T) continue;
        const bool keep = !(p.out_mask && t >= len);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 v[2] = {acc[fm][2 * h], acc[fm][2 * h + 1]};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (p.bias) v[u] += bia[h][u];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = keep ? act_apply_c<ACT>(v[u][e]) * e_scale : 0.f;
          }
          if (rb) {
            const uint4 r = rr[df][h];
            v[0][0] += __uint_as_float(r.x << 16) * e_rscale; v[0][1] += __uint_as_float(r.x & 0xffff0000u) * e_rscale;
            v[0][2] += __uint_as_float(r.y << 16) * e_rscale; v[0][3] += __uint_as_float(r.y & 0xffff0000u) * e_rscale;
            v[1][0] += __uint_as_float(r.z << 16) * e_rscale; v[1][1] += __uint_as_float(r.z & 0xffff0000u) * e_rscale;
            v[1][2] += __uint_as_float(r.w << 16) * e_rscale; v[1][3] += __uint_as_float(r.w & 0xffff0000u) * e_rscale;
          }
          uint4 o;
          o.x = (uint32_t)f32_to_bf16(v[0][0]) | ((uint32_t)f32_to_bf16(v[0][1]) << 16);
          o.y = (uint32_t)f32_to_bf16(v[0][2]) | ((uint32_t)f32_to_bf16(v[0][3]) << 16);
          o.z = (uint32_t)f32_to_bf16(v[1][0]) | ((uint32_t)f32_to_bf16(v[1][1]) << 16);
          o.w = (uint32_t)f32_to_bf16(v[1][2]) | ((uint32_t)f32_to_bf16(v[1][3]) << 16);
          *reinterpret_cast<uint4*>(yb + (int64_t)t * p.ldy + wn * 64 + h * 32 + lg * 8) = o;
          if (p.aux) {  // from the ROUNDED output, like a separate pass over y would compute it
            const float sc = t < len ? p.aux_scale : 0.f;
            uint4 q;
            q.x = (uint32_t)f32_to_bf16(__uint_as_float(o.x << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.x & 0xffff0000u) * sc) << 16);
            q.y = (uint32_t)f32_to_bf16(__uint_as_float(o.y << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.y & 0xffff0000u) * sc) << 16);
            q.z = (uint32_t)f32_to_bf16(__uint_as_float(o.z << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.z & 0xffff0000u) * sc) << 16);
            q.w = (uint32_t)f32_to_bf16(__uint_as_float(o.w << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.w & 0xffff0000u) * sc) << 16);
            *reinterpret_cast<uint4*>(p.aux + ((int64_t)b * T + t) * p.ldaux + wn * 64 + h * 32 + lg * 8) = q;
          }
        }
      }
    }
  }
}


// ---- the same convolution with the weight fragments STRAIGHT FROM GLOBAL MEMORY on a 1 x 8 wave grid (round 5) -----------------
// The ring loop above spends ~100 scalar instructions per step (two LDS-DMA issues with their M0 traffic, a counted wait picked
// by comparisons, slot / chunk / tap counters, a barrier) beside 16 MFMAs -- two waves per SIMD then need ~2x the matrix time
// (profiles/r05_kloop_instruction_mix.txt).  Here a wave owns ALL rows of the block x 32 output channels, so no two waves read
// the same weight fragment: the fragments (1 KiB contiguous each in the stream's stage image, two per step) come with plain
// global loads three steps ahead into four rotating register sets, the x window stays in LDS (double-buffered per 64-channel
// chunk, LDS-DMA), and the only barriers are the window swaps.  KS is a template parameter and a trip covers two chunks
// (4 KS straight-line steps): taps, k-halves, register sets, wait counts and the window pieces' position are compile-time.
// Same stream, same K order, same epilogue arithmetic as the ring kernel: bit-identical outputs.
template <typename F, int... I>
__device__ __forceinline__ void rt_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void rt_static_for(F&& f) {
  rt_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// EPI = 1: the DiffNet gate backward as the epilogue (tile_epilogue_gate_bwd of conv1d_glds.h; reference modules/denoiser.py:76-77
// differentiated): the accumulator is dg, never stored; with s = a[:, c], f = a[:, N + c]: da[:, c] = dg th sg (1 - sg),
// da[:, N + c] = dg sg (1 - th^2), dg rounded to bf16 first as the two-kernel path stores it.
template <int FM, int KS, int ACT, int EPI = 0>
__global__ __launch_bounds__(512, 2) void conv1d_rt_gw_kernel(const RtP p) {
  constexpr int BM = 16 * FM;
  constexpr int CS = 2 * KS;       // steps per chunk
  constexpr int NSTEP = 2 * CS;    // steps per trip (two chunks)
  constexpr int NPW = (BM + 7) / 8 + 6 > 16 ? 3 : 2;  // window pieces per wave (runtime np <= 8 NPW is checked by the launcher)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Xw = reinterpret_cast<uint4*>(smem);  // x windows [2][xrows][8]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.nMT, mt = blockIdx.x - b * p.nMT;
  const int t0 = mt * BM;
  const int T = p.T, dil = p.dil;
  const int xrows = (BM + (KS - 1) * dil + 7) & ~7;
  const int np = xrows >> 3;
  const int grp = blockIdx.y, split = blockIdx.z;  // output-channel group of 256; share of the Cin trips (0 / 0 for the plain forms)
  const int trip0 = split * p.split_trips;
  const int ntrips = min(p.split_trips, (p.Cin >> 7) - trip0);  // two 64-channel chunks per trip
  const int c0 = 2 * trip0;                                      // first 64-channel chunk of this block
  const int len_raw = p.lengths ? p.lengths[b] : T;

  const bf16_raw* xb = p.x + (int64_t)b * T * p.ldx;
  const uint32_t xs_lds = lds_addr(Xw);
  const char* zero = reinterpret_cast<const char*>(g_conv_zero_page) + lane * 16;

  f32x4 acc[FM][2];
#pragma unroll
  for (int i = 0; i < FM; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 W[4][2];
  const uint32_t gw_off = (uint32_t)(((wave * 32 + lr) * 4 + (lg ^ swz<4>(wave * 32 + lr))) * 16);
  // stage of the next group to request: a trip is 4 KS stages of 16 KiB
  const char* wptr = reinterpret_cast<const char*>(p.wstream) + (int64_t)grp * p.grp_stride + (int64_t)trip0 * (4 * KS * RT_STAGE_U4 * 16);
  auto ldg = [&](u32x4 (&w)[2]) __attribute__((always_inline)) {
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(w[0]), "=&v"(w[1]) : "v"(gw_off), "s"(wptr) : "memory");
    wptr += RT_STAGE_U4 * 16;
  };
  ldg(W[0]);  // (independent of the utterance length)
  const int len = min(len_raw, T);
  const int Tin = p.in_mask ? len : T;
  const bool active = !((p.out_mask && t0 >= len) || (p.in_mask && t0 - p.pad >= len));

  // window pieces: piece q = rows 8 q .. 8 q + 7; lane (l3 = lane / 8, l7 = lane % 8) moves 16 bytes of row l3, chunk column
  // l7 ^ swz<8>(8 q + l3) = l7 ^ (l3 / 2) ^ 4 (q & 1).  Every wave issues exactly NPW pieces per window (a wave without a
  // piece of its own repeats the last one: same bytes, same place) so the waits below count compile-time constants.
  const uint32_t gx_voff = (uint32_t)((lane >> 3) * (p.ldx * 2) + (((lane & 7) ^ (lane >> 4)) << 4));
  auto piece = [&](int ci, int par, int k) __attribute__((always_inline)) {
    const int q = min(wave + 8 * k, np - 1);
    const int ts0 = t0 - p.pad + q * 8;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(xs_lds + (uint32_t)((par * xrows + q * 8) * 128));
    const bf16_raw* base = xb + (int64_t)ts0 * p.ldx + (c0 + ci) * 64;
    const uint32_t voff = gx_voff ^ (uint32_t)((q & 1) << 6);
    if (ts0 >= 0 && ts0 + 7 < Tin) {
      glds16_s(base, voff, dst);
    } else {
      const int ts = ts0 + (lane >> 3);
      const char* src = (ts >= 0 && ts < Tin) ? reinterpret_cast<const char*>(base) + voff : zero;
      glds16(src, dst);
    }
  };
  // x fragments of a step: tile fm is 16 rows = 2 KiB further (both swizzles unchanged)
  auto ld_x = [&](uint4 (&xf)[FM], int par, int tap, int kh) __attribute__((always_inline)) {
    int lr_o = lr;
    asm volatile("" : "+v"(lr_o));  // (opaque: the 2 KS addresses of a chunk are computed where they are used -- five VALU
                                    //  instructions per step -- instead of living in 2 KS registers across the trip loop)
    const int r = lr_o + tap * dil;
    const uint4* src = Xw + par * xrows * 8 + r * 8 + ((kh * 4 + lg) ^ swz<8>(r));
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) xf[fm] = src[fm * 128];
  };
  auto wait_n = [&](int n) __attribute__((always_inline)) {  // (n is a compile-time constant at every call)
    if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto mfma = [&](const u32x4 (&w)[2], const uint4 (&xf)[FM]) __attribute__((always_inline)) {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int fn = 0; fn < 2; ++fn)
        acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[fn]), __builtin_bit_cast(bf16x8_t, xf[fm]),
                                                              acc[fm][fn], 0, 0, 0);
  };

  if (active) {
    uint4 xa[FM], xb_[FM];
#pragma unroll
    for (int k = 0; k < NPW; ++k) piece(0, 0, k);
    ldg(W[1]);
    ldg(W[2]);
    wait_n(4);  // the pieces are older than the groups of steps 1 and 2 (group 0 is older still)
    rt_barrier();
    ld_x(xa, 0, 0, 0);
#pragma unroll 1
    for (int t = 0; t < ntrips; ++t) {
      const bool more = t + 1 < ntrips;  // another trip follows: its first window and weight groups are requested in this one
      // (the step index MUST be a compile-time constant: with a run-time index the register-set selection becomes copies of
      //  the asm loads' destination registers, made before the data has arrived -- a 68-step `#pragma unroll` loop was only
      //  unrolled by half)
      rt_static_for<NSTEP>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int cc = i / CS, j = i - cc * CS;  // chunk of the trip (= LDS window), step in the chunk
        constexpr int tap = j >> 1, kh = j & 1;
        (void)tap; (void)kh;
        constexpr bool pieces_here = j == 0;                       // the next chunk's window goes out at a chunk's first step
        constexpr bool pieces_exist = cc == 0;                      // ... unconditionally inside a trip, else only if (more)
        if (pieces_here && (pieces_exist || more)) {
#pragma unroll
          for (int k = 0; k < NPW; ++k) piece(2 * t + cc + 1, cc ^ 1, k);
        }
        if (i + 3 < NSTEP || more) ldg(W[(i + 3) & 3]);
        // the next step's x fragments, unless it opens a chunk (those follow the window barrier below)
        if (j + 1 < CS) {
          if (i & 1) ld_x(xa, cc, (j + 1) >> 1, (j + 1) & 1);
          else ld_x(xb_, cc, (j + 1) >> 1, (j + 1) & 1);
        }
        // younger than this step's weights: the groups of the next three steps (two loads each) and, during a chunk's first
        // three steps, the window pieces issued at its first step (between the groups of +2 and +3)
        {
          const int yg_static = NSTEP - 1 - i < 3 ? NSTEP - 1 - i : 3;  // groups that exist without another trip
          const bool pz = j < 3;
          if (i + 3 < NSTEP) {
            if (pz && !pieces_exist) {
              if (more) wait_n(6 + NPW);
              else wait_n(6);
            } else {
              wait_n(6 + (pz ? NPW : 0));
            }
          } else {  // (the trip's last three steps: j >= 3 for every KS >= 3, no pieces in flight)
            if (more) wait_n(6);
            else wait_n(2 * yg_static);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i & 1) mfma(W[i & 3], xb_);
        else mfma(W[i & 3], xa);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 == CS && (cc == 0 || more)) {  // window swap: everyone's pieces of the next chunk have landed (in-order waits above)
          if constexpr (CS < 4) {  // (a chunk of two steps: the waits above have not reached the pieces yet -- younger than them are
                                   //  the weight groups requested at this chunk's steps)
            if (cc == 0 && !more) wait_n(2);
            else wait_n(4);
          }
          rt_barrier();
          if (i & 1) ld_x(xa, cc ^ 1, 0, 0);
          else ld_x(xb_, cc ^ 1, 0, 0);
        }
      });
    }
  } else {
    glds_wait<0>();
  }

  if constexpr (EPI == 1) {
    const bf16_raw* ab = p.gate_a + (int64_t)b * T * (2 * RT_N);
    bf16_raw* dab = p.gate_da + (int64_t)b * T * p.ldda;
    const int ch = wave * 32 + lg * 8;
#pragma unroll
    for (int f0 = 0; f0 < FM; f0 += 2) {
      uint4 rs[2], rf[2];
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        rs[df] = rf[df] = make_uint4(0, 0, 0, 0);
        const int t = t0 + (f0 + df) * 16 + lr;
        if (f0 + df < FM && t < T) {
          rs[df] = *reinterpret_cast<const uint4*>(ab + (int64_t)t * (2 * RT_N) + ch);
          rf[df] = *reinterpret_cast<const uint4*>(ab + (int64_t)t * (2 * RT_N) + RT_N + ch);
        }
      }
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const int fm = f0 + df < FM ? f0 + df : FM - 1;
        const int t = t0 + fm * 16 + lr;
        if (f0 + df >= FM || t >= T) continue;
        const uint4 s4 = rs[df], f4 = rf[df];
        const float sv[8] = {__uint_as_float(s4.x << 16), __uint_as_float(s4.x & 0xffff0000u), __uint_as_float(s4.y << 16),
                             __uint_as_float(s4.y & 0xffff0000u), __uint_as_float(s4.z << 16), __uint_as_float(s4.z & 0xffff0000u),
                             __uint_as_float(s4.w << 16), __uint_as_float(s4.w & 0xffff0000u)};
        const float fv[8] = {__uint_as_float(f4.x << 16), __uint_as_float(f4.x & 0xffff0000u), __uint_as_float(f4.y << 16),
                             __uint_as_float(f4.y & 0xffff0000u), __uint_as_float(f4.z << 16), __uint_as_float(f4.z & 0xffff0000u),
                             __uint_as_float(f4.w << 16), __uint_as_float(f4.w & 0xffff0000u)};
        float ds[8], dfv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = bf16_to_f32(f32_to_bf16(acc[fm][e >> 2][e & 3] * p.out_scale));
          float sg, th;
          gate_fast_parts(sv[e], fv[e], sg, th);
          ds[e] = d * th * sg * (1.f - sg);
          dfv[e] = d * sg * (1.f - th * th);
        }
        uint4 o1, o2;
        o1.x = (uint32_t)f32_to_bf16(ds[0]) | ((uint32_t)f32_to_bf16(ds[1]) << 16);
        o1.y = (uint32_t)f32_to_bf16(ds[2]) | ((uint32_t)f32_to_bf16(ds[3]) << 16);
        o1.z = (uint32_t)f32_to_bf16(ds[4]) | ((uint32_t)f32_to_bf16(ds[5]) << 16);
        o1.w = (uint32_t)f32_to_bf16(ds[6]) | ((uint32_t)f32_to_bf16(ds[7]) << 16);
        o2.x = (uint32_t)f32_to_bf16(dfv[0]) | ((uint32_t)f32_to_bf16(dfv[1]) << 16);
        o2.y = (uint32_t)f32_to_bf16(dfv[2]) | ((uint32_t)f32_to_bf16(dfv[3]) << 16);
        o2.z = (uint32_t)f32_to_bf16(dfv[4]) | ((uint32_t)f32_to_bf16(dfv[5]) << 16);
        o2.w = (uint32_t)f32_to_bf16(dfv[6]) | ((uint32_t)f32_to_bf16(dfv[7]) << 16);
        *reinterpret_cast<uint4*>(dab + (int64_t)t * p.ldda + ch) = o1;
        *reinterpret_cast<uint4*>(dab + (int64_t)t * p.ldda + RT_N + ch) = o2;
      }
    }
    return;
  }
  if constexpr (EPI == 2) {  // split over Cin: raw f32 partial sums [split][B][T][cout_total], epilogue in conv_splitk_finish_kernel
    float* wsb = p.ws + ((int64_t)split * p.B + b) * T * p.cout_total + grp * RT_N + wave * 32 + lg * 8;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const int t = t0 + fm * 16 + lr;
      if (t < T) {
        *reinterpret_cast<f32x4*>(wsb + (int64_t)t * p.cout_total) = acc[fm][0];
        *reinterpret_cast<f32x4*>(wsb + (int64_t)t * p.cout_total + 4) = acc[fm][1];
      }
    }
    return;
  }
  // ---- epilogue: a lane holds 8 consecutive channels (wave * 32 + lg * 8 ..) of row fm * 16 + lr; conv_epilogue_act arithmetic
  {
    bf16_raw* yb = p.y + (int64_t)b * T * p.ldy;
    const bf16_raw* rb = p.res ? p.res + (int64_t)b * T * p.ldr : nullptr;
    const float e_scale = p.out_scale, e_rscale = p.res_scale;
    const int ch = grp * RT_N + wave * 32 + lg * 8;
    const unsigned e_dth = p.drop_thresh16;
    f32x4 bia[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) bia[u] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ch + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // (colpart) this lane's rows of the rounded output, per channel
#pragma unroll
    for (int f0 = 0; f0 < FM; f0 += 4) {
      uint4 rr[4];
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        rr[df] = make_uint4(0, 0, 0, 0);
        const int t = t0 + (f0 + df) * 16 + lr;
        if (rb && f0 + df < FM && t < T) rr[df] = *reinterpret_cast<const uint4*>(rb + (int64_t)t * p.ldr + ch);
      }
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const int fm = f0 + df < FM ? f0 + df : FM - 1;
        const int t = t0 + fm * 16 + lr;
        if (f0 + df >= FM || t >= T) continue;
        const bool keep = !(p.out_mask && t >= len);
        f32x4 v[2] = {acc[fm][0], acc[fm][1]};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (p.bias) v[u] += bia[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][e] = keep ? act_apply_c<ACT>(v[u][e]) * e_scale : 0.f;
          if (e_dth) v[u] *= drop_mask4(p.drop_seed, (uint64_t)(((int64_t)b * T + t) * p.cout_total + ch + 4 * u) >> 2, e_dth, p.drop_inv_keep);
        }
        if (rb) {
          const uint4 r = rr[df];
          v[0][0] += __uint_as_float(r.x << 16) * e_rscale; v[0][1] += __uint_as_float(r.x & 0xffff0000u) * e_rscale;
          v[0][2] += __uint_as_float(r.y << 16) * e_rscale; v[0][3] += __uint_as_float(r.y & 0xffff0000u) * e_rscale;
          v[1][0] += __uint_as_float(r.z << 16) * e_rscale; v[1][1] += __uint_as_float(r.z & 0xffff0000u) * e_rscale;
          v[1][2] += __uint_as_float(r.w << 16) * e_rscale; v[1][3] += __uint_as_float(r.w & 0xffff0000u) * e_rscale;
        }
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(v[0][0]) | ((uint32_t)f32_to_bf16(v[0][1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[0][2]) | ((uint32_t)f32_to_bf16(v[0][3]) << 16);
        o.z = (uint32_t)f32_to_bf16(v[1][0]) | ((uint32_t)f32_to_bf16(v[1][1]) << 16);
        o.w = (uint32_t)f32_to_bf16(v[1][2]) | ((uint32_t)f32_to_bf16(v[1][3]) << 16);
        if (p.relu_src) {  // (y itself is not stored in this form)
          const uint4 hv = *reinterpret_cast<const uint4*>(p.relu_src + ((int64_t)b * T + t) * p.ldrs + ch);
          const float iv = p.relu_inv;
          const bool rk = t < len;
          auto gate2 = [&](uint32_t ov, uint32_t hw) __attribute__((always_inline)) {
            const float lo = (rk && __uint_as_float(hw << 16) > 0.f) ? __uint_as_float(ov << 16) * iv : 0.f;
            const float hi = (rk && __uint_as_float(hw & 0xffff0000u) > 0.f) ? __uint_as_float(ov & 0xffff0000u) * iv : 0.f;
            return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
          };
          uint4 q;
          q.x = gate2(o.x, hv.x); q.y = gate2(o.y, hv.y); q.z = gate2(o.z, hv.z); q.w = gate2(o.w, hv.w);
          *reinterpret_cast<uint4*>(p.relu_dz + ((int64_t)b * T + t) * p.lddz + ch) = q;
          continue;
        }
        *reinterpret_cast<uint4*>(yb + (int64_t)t * p.ldy + ch) = o;
        if (p.colpart) {
          cs[0] += __uint_as_float(o.x << 16); cs[1] += __uint_as_float(o.x & 0xffff0000u);
          cs[2] += __uint_as_float(o.y << 16); cs[3] += __uint_as_float(o.y & 0xffff0000u);
          cs[4] += __uint_as_float(o.z << 16); cs[5] += __uint_as_float(o.z & 0xffff0000u);
          cs[6] += __uint_as_float(o.w << 16); cs[7] += __uint_as_float(o.w & 0xffff0000u);
        }
        if (p.aux) {  // from the ROUNDED output, like a separate pass over y would compute it
          const float sc = t < len ? p.aux_scale : 0.f;
          uint4 q;
          q.x = (uint32_t)f32_to_bf16(__uint_as_float(o.x << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.x & 0xffff0000u) * sc) << 16);
          q.y = (uint32_t)f32_to_bf16(__uint_as_float(o.y << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.y & 0xffff0000u) * sc) << 16);
          q.z = (uint32_t)f32_to_bf16(__uint_as_float(o.z << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.z & 0xffff0000u) * sc) << 16);
          q.w = (uint32_t)f32_to_bf16(__uint_as_float(o.w << 16) * sc) | ((uint32_t)f32_to_bf16(__uint_as_float(o.w & 0xffff0000u) * sc) << 16);
          *reinterpret_cast<uint4*>(p.aux + ((int64_t)b * T + t) * p.ldaux + ch) = q;
        }
      }
    }
    if (p.colpart) {  // the 16 lanes of a channel group hold the tile's rows: fixed butterfly, then lane lr = k owns slot k of the tile
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = cs[e];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        cs[e] = v;
      }
      const int nslot = (T + 31) >> 5, s0 = t0 >> 5;
      if (lr < BM / 32 && s0 + lr < nslot) {
        float* d = p.colpart + ((int64_t)b * nslot + s0 + lr) * RT_N + wave * 32 + lg * 8;
        const float z = lr == 0 ? 1.f : 0.f;
        *reinterpret_cast<f32x4*>(d) = f32x4{cs[0] * z, cs[1] * z, cs[2] * z, cs[3] * z};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{cs[4] * z, cs[5] * z, cs[6] * z, cs[7] * z};
      }
    }
  }
}

template <int FM, int KS, int ACT, int EPI = 0>
int rt_gw_launch(const RtP& p, hipStream_t st) {
  constexpr int BM = 16 * FM;
  const int xrows = (BM + (KS - 1) * p.dil + 7) & ~7;
  const size_t smem = (size_t)2 * xrows * 128;
  auto kern = conv1d_rt_gw_kernel<FM, KS, ACT, EPI>;
  if (smem > 64 * 1024) {
    const void* kp = reinterpret_cast<const void*>(kern);
    if (!lds_limit_raised(kp)) {
      const hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        ptpp_set_error("conv1d_rt_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        return PTPP_ELAUNCH;
      }
      lds_limit_mark(kp);
    }
  }
  const unsigned ngrp = (unsigned)(p.cout_total / RT_N);
  const unsigned nsplit = (unsigned)(((p.Cin >> 7) + p.split_trips - 1) / p.split_trips);
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT), ngrp, nsplit), dim3(512), smem, st, p);
  PTPP_CHECK_LAUNCH("conv1d_rt_fwd");
  return PTPP_OK;
}
template <int KS>
int rt_gw_dispatch(const RtP& p, int bm, bool relu, hipStream_t st) {
  if (bm == 160) return relu ? rt_gw_launch<10, KS, PTPP_ACT_RELU>(p, st) : rt_gw_launch<10, KS, PTPP_ACT_NONE>(p, st);
  if (bm == 128) return relu ? rt_gw_launch<8, KS, PTPP_ACT_RELU>(p, st) : rt_gw_launch<8, KS, PTPP_ACT_NONE>(p, st);
  if (bm == 96) return relu ? rt_gw_launch<6, KS, PTPP_ACT_RELU>(p, st) : rt_gw_launch<6, KS, PTPP_ACT_NONE>(p, st);
  return relu ? rt_gw_launch<4, KS, PTPP_ACT_RELU>(p, st) : rt_gw_launch<4, KS, PTPP_ACT_NONE>(p, st);
}

template <int FM, int ACT>
int rt_launch(const RtP& p, hipStream_t st) {
  constexpr int NS = 5;
  constexpr int BM = 32 * FM;
  const int xrows = (BM + (p.ks - 1) * p.dil + 7) & ~7;
  const size_t smem = (size_t)NS * RT_STAGE_U4 * 16 + (size_t)2 * xrows * 128;
  auto kern = conv1d_rt_kernel<NS, FM, ACT>;
  {
    const void* kp = reinterpret_cast<const void*>(kern);
    if (!lds_limit_raised(kp)) {
      const hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        ptpp_set_error("conv1d_rt_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        return PTPP_ELAUNCH;
      }
      lds_limit_mark(kp);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)p.B * p.nMT)), dim3(512), smem, st, p);
  PTPP_CHECK_LAUNCH("conv1d_rt_fwd");
  return PTPP_OK;
}

}  // namespace

static int rt_bm_for(int B, int T, bool tall);

extern "C" int ptpp_conv1d_rt_supported(int cin, int cout, int ks, int dil, int act, int dtype) {
  if (dtype != PTPP_BF16 || cout != RT_N || cin <= 0 || (cin & 63) || ks < 1 || dil < 1) return 0;
  if (act != PTPP_ACT_NONE && act != PTPP_ACT_RELU) return 0;
  if (ks < 3) {  // 1 x 1 (round 6: the DiffNet conditioner's data gradient, K = 20 x 512): the global-weights form only
    const char* gwe = getenv("PTPP_CONV_RT_GW");
    return ks == 1 && (cin & 127) == 0 && !(gwe && gwe[0] == '0');
  }
  const int xrows = (128 + (ks - 1) * dil + 7) & ~7;
  return xrows <= 160;  // two windows of 20 KiB beside the 80 KiB ring
}

extern "C" int ptpp_conv1d_rt_fwd(const ptpp_conv1d_args* a, const void* wstream, float res_scale, void* stream) {
  return ptpp_conv1d_rt_fwd_aux(a, wstream, res_scale, nullptr, 0, 0.f, stream);
}

// whether a launch of this geometry takes the global-weights form (the only one with the column-sum epilogue), and its tile height
static bool rt_takes_gw(int Cin, int ks, int dil, int B, int T, int* bm_out) {
  const char* gwe = getenv("PTPP_CONV_RT_GW");
  const bool gw_ok = !(gwe && gwe[0] == '0') && (Cin & 127) == 0 && (ks == 1 || ks == 3 || ks == 5 || ks == 17);
  int bm = rt_bm_for(B, T, gw_ok);
  if (bm == 160 && ((160 + (ks - 1) * dil + 7) >> 3) > 24) bm = rt_bm_for(B, T, false);  // (window pieces: 8 waves x 3)
  if (bm_out) *bm_out = bm;
  const int xr = (bm + (ks - 1) * dil + 7) & ~7;
  const int npw = (bm + 7) / 8 + 6 > 16 ? 3 : 2;
  return gw_ok && (xr >> 3) <= 8 * npw;
}

extern "C" int ptpp_conv1d_rt_colpart_supported(int cin, int ks, int dil, int B, int T) {
  return B > 0 && T > 0 && rt_takes_gw(cin, ks, dil, B, T, nullptr) ? 1 : 0;
}

extern "C" int ptpp_conv1d_rt_fwd_aux(const ptpp_conv1d_args* a, const void* wstream, float res_scale, void* aux, int ldaux, float aux_scale,
                                      void* stream) {
  return ptpp_conv1d_rt_fwd_cs(a, wstream, res_scale, aux, ldaux, aux_scale, nullptr, stream);
}

extern "C" int ptpp_conv1d_rt_fwd_cs(const ptpp_conv1d_args* a, const void* wstream, float res_scale, void* aux, int ldaux, float aux_scale,
                                     float* colpart, void* stream) {
  PTPP_CHECK_ARG(a && a->x && a->y && wstream, "conv1d_rt_fwd: null pointer");
  PTPP_CHECK_ARG(!colpart || (((uintptr_t)colpart & 15) == 0 && rt_takes_gw(a->Cin, a->ks, a->dil, a->B, a->T, nullptr)),
                 "conv1d_rt_fwd: the column-sum output needs the global-weights form (ptpp_conv1d_rt_colpart_supported)");
  PTPP_CHECK_ARG(!aux || ((ldaux & 7) == 0 && ldaux >= RT_N && ((uintptr_t)aux & 15) == 0), "conv1d_rt_fwd: bad aux output (ld %d)", ldaux);
  PTPP_CHECK_ARG(ptpp_conv1d_rt_supported(a->Cin, a->Cout, a->ks, a->dil, a->act, a->dtype),
                 "conv1d_rt_fwd: unsupported shape (bf16, Cout = 256, Cin %% 64 == 0, ks >= 3 or 1 x 1 with Cin %% 128 == 0, act none / relu; Cin %d Cout %d ks %d dil %d act %d)",
                 a->Cin, a->Cout, a->ks, a->dil, a->act);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->pad >= 0 && a->pad <= (a->ks - 1) * a->dil && (a->ldx & 7) == 0 && (a->ldy & 7) == 0 &&
                     (!a->res || (a->ldr & 7) == 0),
                 "conv1d_rt_fwd: bad geometry (B %d T %d pad %d ldx %d ldy %d ldr %d)", a->B, a->T, a->pad, a->ldx, a->ldy, a->ldr);
  PTPP_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->res | (uintptr_t)wstream | (uintptr_t)a->bias) & 15) == 0,
                 "conv1d_rt_fwd: operands must be 16-byte aligned");
  PTPP_CHECK_ARG(!(a->in_mask || a->out_mask) || a->lengths, "conv1d_rt_fwd: masks need lengths");
  RtP p;
  p.x = reinterpret_cast<const bf16_raw*>(a->x);
  p.wstream = reinterpret_cast<const uint4*>(wstream);
  p.bias = a->bias;
  p.res = reinterpret_cast<const bf16_raw*>(a->res);
  p.y = reinterpret_cast<bf16_raw*>(a->y);
  p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.ks = a->ks; p.dil = a->dil; p.pad = a->pad;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldr = a->ldr;
  p.act = a->act; p.in_mask = a->in_mask; p.out_mask = a->out_mask;
  p.out_scale = a->out_scale; p.res_scale = res_scale;
  p.aux = reinterpret_cast<bf16_raw*>(aux); p.ldaux = ldaux; p.aux_scale = aux_scale;
  p.gate_a = nullptr; p.gate_da = nullptr; p.ldda = 0;
  rtp_plain(p);
  p.colpart = colpart;
  // the global-weights form (1 x 8 wave grid) for the tap counts of the model's layers; PTPP_CONV_RT_GW=0: the ring form
  int bm = 128;
  const bool gw = rt_takes_gw(a->Cin, a->ks, a->dil, a->B, a->T, &bm);
  p.nMT = (a->T + bm - 1) / bm;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool relu = a->act == PTPP_ACT_RELU;
  {
    if (gw) {
      if (a->ks == 1) return rt_gw_dispatch<1>(p, bm, relu, st);
      if (a->ks == 3) return rt_gw_dispatch<3>(p, bm, relu, st);
      if (a->ks == 5) return rt_gw_dispatch<5>(p, bm, relu, st);
      if (a->ks == 17) return rt_gw_dispatch<17>(p, bm, relu, st);
    }
  }
  if (bm == 128) return relu ? rt_launch<4, PTPP_ACT_RELU>(p, st) : rt_launch<4, PTPP_ACT_NONE>(p, st);
  if (bm == 96) return relu ? rt_launch<3, PTPP_ACT_RELU>(p, st) : rt_launch<3, PTPP_ACT_NONE>(p, st);
  return relu ? rt_launch<2, PTPP_ACT_RELU>(p, st) : rt_launch<2, PTPP_ACT_NONE>(p, st);
}

// ---- round 6: the phone-level feed-forward convs of the Conformer blocks (reference modules/esp/transformer/multi_layer_conv.py:52-67:
// 256 -> 1024 -> 256, k = 9; their data gradients are the same two shapes) on the global-weights form.  ~3 000 rows in ~20 utterances
// are few tiles for 256 CUs: the 1024-channel output is four column groups (blockIdx.y) over the same x window, the 1024-channel
// input is split over blockIdx.z (f32 partial sums, epilogue in conv_splitk_finish_kernel), 64-row blocks keep the padding of
// utterances of ~160 phones at 17 % (128-row tiles: 34 %).  Epilogue with dropout on the conv term (ptpp_conv1d_fwd_ex's
// arithmetic); the K order inside a split is the tile kernel's, the order ACROSS splits is this kernel's own.
int ptpp_conv_splitk_finish_bf16(const ptpp_conv1d_args* a, float res_scale, float drop_p, uint64_t drop_seed, float* ws, int nsplit,
                                 hipStream_t st);

extern "C" int ptpp_conv1d_rt_ex_supported(int cin, int cout, int ks, int dil, int act, int dtype) {
  const char* gwe = getenv("PTPP_CONV_RT_GW");
  if (gwe && gwe[0] == '0') return 0;
  const char* exe = getenv("PTPP_CONV_RT_EX");  // (A/B knob: 0 = the feed-forward convs stay on the tile kernels)
  if (exe && exe[0] == '0') return 0;
  if (dtype != PTPP_BF16 || cout <= 0 || cout % RT_N || cout > 4096 || cin <= 0 || (cin & 127) || ks != 9 || dil != 1) return 0;
  return act == PTPP_ACT_NONE || act == PTPP_ACT_RELU;
}

static int rt_fwd_ex_impl(const ptpp_conv1d_args* a, const void* wstream, float res_scale, float drop_p, uint64_t drop_seed, void* workspace,
                          size_t workspace_bytes, const void* relu_src, void* relu_dz, float relu_inv, int* fused, void* stream,
                          int* partial_nsplit = nullptr);

extern "C" int ptpp_conv1d_rt_fwd_ex(const ptpp_conv1d_args* a, const void* wstream, float res_scale, float drop_p, uint64_t drop_seed,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return rt_fwd_ex_impl(a, wstream, res_scale, drop_p, drop_seed, workspace, workspace_bytes, nullptr, nullptr, 1.f, nullptr, stream);
}

// y = conv(x) is an intermediate: dz = [saved > 0 and row inside its utterance] * y * inv_keep, i.e. ptpp_epilogue_bwd(y, saved, dz,
// lengths, .., scale 1, relu 1, mask 1, p) fused into the conv's epilogue where the launch is not split (a->y is then NOT
// written); a split launch runs the two steps one after the other through a->y.  saved / dz: (B, T, Cout) rows of Cout elements.
extern "C" int ptpp_conv1d_rt_fwd_ex_relu_bwd(const ptpp_conv1d_args* a, const void* wstream, const void* saved, void* dz, float drop_p,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  PTPP_CHECK_ARG(a && saved && dz && a->lengths && !a->res && !a->bias && a->act == PTPP_ACT_NONE && a->out_scale == 1.f && !a->out_mask &&
                     a->ldy == a->Cout && (((uintptr_t)saved | (uintptr_t)dz) & 15) == 0 && drop_p >= 0.f && drop_p < 1.f,
                 "conv1d_rt_fwd_ex_relu_bwd: plain conv (no bias / residual / activation / output mask, dense y) with lengths expected");
  const unsigned th = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  const float inv = drop_p > 0.f ? 1.f / (1.f - th / 65536.f) : 1.f;
  int fused = 0;
  const int rc = rt_fwd_ex_impl(a, wstream, 1.f, 0.f, 0, workspace, workspace_bytes, saved, dz, inv, &fused, stream);
  if (rc != PTPP_OK || fused) return rc;
  return ptpp_epilogue_bwd(a->y, saved, dz, a->lengths, a->B, a->T, a->Cout, 1.0f, 1, 1, drop_p, 1, a->dtype, stream);
}

// The launch WITHOUT its finishing pass where it is split over Cin: *nsplit > 1 on return means `workspace` holds [nsplit][B][T][Cout]
// raw f32 partial sums and a->y was not written -- the caller's next kernel sums them (ptpp_layernorm_bwd_add_splitk: the finishing
// pass of conv_splitk_finish_kernel folded into the LayerNorm backward that reads the result); *nsplit == 1: a->y is complete.
// a: no bias / residual / activation / dropout (the epilogue the consumer reproduces is "sum, output mask, round").
extern "C" int ptpp_conv1d_rt_fwd_ex_partial(const ptpp_conv1d_args* a, const void* wstream, void* workspace, size_t workspace_bytes,
                                             int* nsplit, void* stream) {
  PTPP_CHECK_ARG(a && nsplit && !a->res && !a->bias && a->act == PTPP_ACT_NONE && a->out_scale == 1.f,
                 "conv1d_rt_fwd_ex_partial: plain conv expected (no bias / residual / activation / scale)");
  *nsplit = 1;
  return rt_fwd_ex_impl(a, wstream, 1.f, 0.f, 0, workspace, workspace_bytes, nullptr, nullptr, 1.f, nullptr, stream, nsplit);
}

static int rt_fwd_ex_impl(const ptpp_conv1d_args* a, const void* wstream, float res_scale, float drop_p, uint64_t drop_seed, void* workspace,
                          size_t workspace_bytes, const void* relu_src, void* relu_dz, float relu_inv, int* fused, void* stream,
                          int* partial_nsplit) {
  PTPP_CHECK_ARG(a && a->x && a->y && wstream, "conv1d_rt_fwd_ex: null pointer");
  PTPP_CHECK_ARG(ptpp_conv1d_rt_ex_supported(a->Cin, a->Cout, a->ks, a->dil, a->act, a->dtype),
                 "conv1d_rt_fwd_ex: unsupported shape (bf16, Cout %% 256 == 0, Cin %% 128 == 0, ks = 9, act none / relu; Cin %d Cout %d ks %d dil %d act %d)",
                 a->Cin, a->Cout, a->ks, a->dil, a->act);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && a->pad >= 0 && a->pad <= (a->ks - 1) * a->dil && (a->ldx & 7) == 0 && (a->ldy & 7) == 0 &&
                     (!a->res || (a->ldr & 7) == 0),
                 "conv1d_rt_fwd_ex: bad geometry (B %d T %d pad %d ldx %d ldy %d ldr %d)", a->B, a->T, a->pad, a->ldx, a->ldy, a->ldr);
  PTPP_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->res | (uintptr_t)wstream | (uintptr_t)a->bias) & 15) == 0,
                 "conv1d_rt_fwd_ex: operands must be 16-byte aligned");
  PTPP_CHECK_ARG(!(a->in_mask || a->out_mask) || a->lengths, "conv1d_rt_fwd_ex: masks need lengths");
  PTPP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "conv1d_rt_fwd_ex: bad dropout p");
  RtP p;
  p.x = reinterpret_cast<const bf16_raw*>(a->x);
  p.wstream = reinterpret_cast<const uint4*>(wstream);
  p.bias = a->bias;
  p.res = reinterpret_cast<const bf16_raw*>(a->res);
  p.y = reinterpret_cast<bf16_raw*>(a->y);
  p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.ks = a->ks; p.dil = a->dil; p.pad = a->pad;
  p.ldx = a->ldx; p.ldy = a->ldy; p.ldr = a->ldr;
  p.act = a->act; p.in_mask = a->in_mask; p.out_mask = a->out_mask;
  p.out_scale = a->out_scale; p.res_scale = res_scale;
  p.aux = nullptr; p.ldaux = 0; p.aux_scale = 0.f;
  p.gate_a = nullptr; p.gate_da = nullptr; p.ldda = 0;
  rtp_plain(p);
  p.cout_total = a->Cout;
  const int ngrp = a->Cout / RT_N, ntr = a->Cin >> 7;
  p.grp_stride = (int64_t)(a->Cin >> 6) * a->ks * 2 * RT_STAGE_U4 * 16;  // one group's stream: (Cin / 64) chunks x ks taps x 2 stages of 16 KiB
  p.drop_thresh16 = drop_p > 0.f ? (unsigned)(drop_p * 65536.f + 0.5f) : 0u;
  p.drop_inv_keep = drop_p > 0.f ? 1.f / (1.f - p.drop_thresh16 / 65536.f) : 1.f;
  p.drop_seed = drop_seed;
  // rows per block and the split of the Cin trips: least (rounds of one-block-per-CU) x (relative block time)
  if (workspace && ((uintptr_t)workspace & 15)) workspace = nullptr;
  const int64_t slab = (int64_t)a->B * a->T * a->Cout * (int64_t)sizeof(float);
  const int cand[3] = {128, 96, 64};
  const float tblk[3] = {1.f, 0.78f, 0.6f};
  int bm = 64, nsplit = 1;
  float best = 1e30f;
  const char* fbm = getenv("PTPP_CONV_RT_BM");
  const char* fns = getenv("PTPP_CONV_RT_NSPLIT");
  for (int i = 0; i < 3; ++i) {
    if (fbm && atoi(fbm) != cand[i] && (atoi(fbm) == 64 || atoi(fbm) == 96 || atoi(fbm) == 128)) continue;
    const int64_t nb0 = (int64_t)a->B * ((a->T + cand[i] - 1) / cand[i]) * ngrp;
    int ns = 1;
    if (workspace && ntr >= 2 && nb0 < 160) {
      ns = (int)((224 + nb0 - 1) / nb0);
      if (ns > ntr) ns = ntr;
      if ((int64_t)ns * slab > (int64_t)workspace_bytes) ns = (int)((int64_t)workspace_bytes / slab);
      if (ns < 2) ns = 1;
    }
    if (fns && workspace && atoi(fns) >= 1 && atoi(fns) <= ntr && (int64_t)atoi(fns) * slab <= (int64_t)workspace_bytes) ns = atoi(fns);
    const int st_ = (ntr + ns - 1) / ns;
    ns = (ntr + st_ - 1) / st_;
    const int64_t nb = nb0 * ns;
    const float cost = (float)((nb + 255) / 256) * tblk[i] * (float)st_ + (ns > 1 ? 0.15f * (float)ntr : 0.f);
    if (cost < best - 1e-3f) { best = cost; bm = cand[i]; nsplit = ns; }
  }
  p.split_trips = (ntr + nsplit - 1) / nsplit;
  p.nMT = (a->T + bm - 1) / bm;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool relu = a->act == PTPP_ACT_RELU;
  if (nsplit > 1) {
    p.ws = reinterpret_cast<float*>(workspace);
    int rc;
    if (bm == 128) rc = rt_gw_launch<8, 9, PTPP_ACT_NONE, 2>(p, st);
    else if (bm == 96) rc = rt_gw_launch<6, 9, PTPP_ACT_NONE, 2>(p, st);
    else rc = rt_gw_launch<4, 9, PTPP_ACT_NONE, 2>(p, st);
    if (rc != PTPP_OK) return rc;
    if (partial_nsplit) {  // (the caller's next kernel finishes)
      *partial_nsplit = nsplit;
      return PTPP_OK;
    }
    return ptpp_conv_splitk_finish_bf16(a, res_scale, drop_p, drop_seed, p.ws, nsplit, st);
  }
  if (relu_src) {  // (unsplit: the ReLU / dropout backward rides in the epilogue)
    p.relu_src = reinterpret_cast<const bf16_raw*>(relu_src);
    p.relu_dz = reinterpret_cast<bf16_raw*>(relu_dz);
    p.ldrs = p.lddz = a->Cout;
    p.relu_inv = relu_inv;
    if (fused) *fused = 1;
  }
  if (bm == 128) return relu ? rt_gw_launch<8, 9, PTPP_ACT_RELU>(p, st) : rt_gw_launch<8, 9, PTPP_ACT_NONE>(p, st);
  if (bm == 96) return relu ? rt_gw_launch<6, 9, PTPP_ACT_RELU>(p, st) : rt_gw_launch<6, 9, PTPP_ACT_NONE>(p, st);
  return relu ? rt_gw_launch<4, 9, PTPP_ACT_RELU>(p, st) : rt_gw_launch<4, 9, PTPP_ACT_NONE>(p, st);
}

// ---- the DiffNet output projection's data gradient with the gate backward in its epilogue, on the row-tile engine ---------------
// Same contract as ptpp_conv1d_gate_bwd (conv1d_cl.hip) with the weight as an operand STREAM (pack mode 4 of the (2C, C, 1)
// projection weight): dg = conv1x1(do, W^T) is never stored, da (B, T, 2C view, row stride ldda) is written in place.
extern "C" int ptpp_conv1d_rt_gate_bwd_supported(int C, int cin, int dtype) {
  return dtype == PTPP_BF16 && C == RT_N && cin > 0 && (cin & 127) == 0;
}

// rows per block: least (rounds of 256 one-per-CU blocks) x (relative time of a block).  ``tall``: the global-weights form also has
// 160-row blocks (a wave owns ten row tiles: 80 accumulator registers) -- a token-bucket batch whose 128-row tiling needs 257-300
// blocks (a quarter of the bench's batches: 65 x 459, 33 x 904, ...) otherwise runs a second, nearly empty round.
static int rt_bm_for(int B, int T, bool tall) {
  int bm = 128;
  const int cand[4] = {160, 128, 96, 64};
  const float tblk[4] = {1.22f, 1.f, 0.8f, 0.6f};
  float best = 1e30f;
  const char* te = getenv("PTPP_CONV_RT_TALL");  // (A/B knob: 0 = no 160-row blocks; read per call like PTPP_CONV_RT_BM)
  tall = tall && !(te && te[0] == '0');
  for (int i = tall ? 0 : 1; i < 4; ++i) {
    const int64_t nb = (int64_t)B * ((T + cand[i] - 1) / cand[i]);
    const float cost = (float)((nb + 255) / 256) * tblk[i];
    if (cost < best - 1e-3f) { best = cost; bm = cand[i]; }
  }
  const char* force = getenv("PTPP_CONV_RT_BM");  // (experiments / tests)
  if (force) {
    const int f = atoi(force);
    if (f == 64 || f == 96 || f == 128 || (tall && f == 160)) bm = f;
  }
  return bm;
}

extern "C" int ptpp_conv1d_rt_gate_bwd(const ptpp_conv1d_args* a, const void* wstream, const void* act, void* da, int ldda, void* stream) {
  PTPP_CHECK_ARG(a && a->x && wstream && act && da, "conv1d_rt_gate_bwd: null pointer");
  PTPP_CHECK_ARG(a->ks == 1 && a->dil == 1 && a->pad == 0 && a->act == PTPP_ACT_NONE && !a->res && !a->bias &&
                     ptpp_conv1d_rt_gate_bwd_supported(a->Cout, a->Cin, a->dtype),
                 "conv1d_rt_gate_bwd: unsupported shape (bf16, 1 x 1, C = 256, Cin %% 128 == 0; C %d Cin %d ks %d)", a->Cout, a->Cin, a->ks);
  PTPP_CHECK_ARG(a->B > 0 && a->T > 0 && (a->ldx & 7) == 0 && ldda >= 2 * a->Cout && (ldda & 7) == 0, "conv1d_rt_gate_bwd: bad geometry");
  PTPP_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)wstream | (uintptr_t)act | (uintptr_t)da) & 15) == 0, "conv1d_rt_gate_bwd: operands must be 16-byte aligned");
  PTPP_CHECK_ARG(!(a->in_mask || a->out_mask) || a->lengths, "conv1d_rt_gate_bwd: masks need lengths");
  RtP p;
  p.x = reinterpret_cast<const bf16_raw*>(a->x);
  p.wstream = reinterpret_cast<const uint4*>(wstream);
  p.bias = nullptr; p.res = nullptr; p.y = nullptr;
  p.lengths = a->lengths;
  p.B = a->B; p.T = a->T; p.Cin = a->Cin; p.ks = 1; p.dil = 1; p.pad = 0;
  p.ldx = a->ldx; p.ldy = 0; p.ldr = 0;
  p.act = PTPP_ACT_NONE; p.in_mask = a->in_mask; p.out_mask = 0;  // (as the tile kernel: rows past the end follow from the masked input)
  p.out_scale = a->out_scale; p.res_scale = 1.f;
  p.aux = nullptr; p.ldaux = 0; p.aux_scale = 0.f;
  p.gate_a = reinterpret_cast<const bf16_raw*>(act);
  p.gate_da = reinterpret_cast<bf16_raw*>(da);
  p.ldda = ldda;
  rtp_plain(p);
  const int bm = rt_bm_for(a->B, a->T, true);
  p.nMT = (a->T + bm - 1) / bm;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (bm == 160) return rt_gw_launch<10, 1, PTPP_ACT_NONE, 1>(p, st);
  if (bm == 128) return rt_gw_launch<8, 1, PTPP_ACT_NONE, 1>(p, st);
  if (bm == 96) return rt_gw_launch<6, 1, PTPP_ACT_NONE, 1>(p, st);
  return rt_gw_launch<4, 1, PTPP_ACT_NONE, 1>(p, st);
}
