// bf16-rate weight gradient of the channels-last Conv1d:
//   dw[co][ci][j] += sum_{b,t} dy[b,t,co] * x[b, t + j*dil - pad, ci]
// The reduction runs over ROWS, but v_mfma_f32_16x16x32_bf16 wants each lane to hold 8
// consecutive K (= row) values of one channel, while channels-last memory is channel
// contiguous.  gfx950's LDS transpose read (ds_read_b64_tr_b16) does that transpose for
// free: tiles are staged row-major [row][channel] with plain 16-byte copies, and each
// operand fragment is two tr reads (4 rows x 16 channels each).
//
// A block owns one (TM co x TN ci) tile of a GROUP of up to TG taps and a strided subset of
// the 32-row K-chunks (split-K).  The taps of a group share the dy chunk and one x window of
// 32 + (TG-1)*dil rows (each tap reads it at its own row offset): with one tap per block the
// launch re-read both operands once per tap and was bound by L2 -> LDS traffic (368 MB per
// 256->512 k=3 launch = 6.6 TB/s at 56 us, profiles/r02b_*), with three taps per block it moves a
// third of that for the same MFMAs.  Wave tile = TM/2 x TN/2 (up to 64x64 = 16 accumulator tiles).
// Measured: with one chunk in flight a loop iteration costs a full ~1.2 us memory round
// trip, 5x its MFMA time, and a register pipeline two chunks deep spills.  So the chunks
// go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a
// 4-stage ring, three chunks in flight across raw s_barriers with counted vmcnt waits.
// LDS-DMA writes lane-linear 1 KiB pieces, so the bank swizzle of the transpose reads is
// applied to the per-lane SOURCE address; rows outside the utterance read a zero page.
// The bias gradient is one extra MFMA per dy fragment against an all-ones operand.
//
// Split-K partials: device-scope f32 atomics run at only ~80 G/s on the 8-XCD part
// (measured), so with a caller-provided workspace every block STORES its partial tile
// ([split][tap][co][ci], coalesced) and a second kernel sums the splits in a fixed order
// and adds into dw: deterministic, and ~10x cheaper than the atomics.  Without a
// workspace the partials are combined with atomics (fewer splits).
#include <stdlib.h>
#include <type_traits>

#include "ptpp_common.h"
#include "lds_dma.h"
#include "../../include/ptpp.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(8))) short v8s;

constexpr int KR = 32;  // rows per K chunk (one MFMA K-step)
constexpr int NS_DEFAULT = 4;  // LDS ring stages of the split-K kernels (NS - 1 chunks in flight)

__device__ uint4 g_zero_page[64];  // source of out-of-range rows

// The transpose reads are issued as inline asm (frag_at / frag_at2 below): for a compiler-visible LDS read hipcc drains
// ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)) first, which would serialise the ring.  The
// kernel counts vmcnt / lgkmcnt itself (wait_vm, lds_fence).
struct Frag {
  v4s lo, hi;
};
// wait for this wave's LDS reads; the operands tie the fragments to the wait so that no
// consumer can be scheduled above it
template <int N>
__device__ __forceinline__ void lds_fence(Frag (&f)[N]) {
  if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo),
                   "+v"(f[3].hi));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi));
}
// the same with N younger LDS requests allowed to stay outstanding
template <int NW, int N>
__device__ __forceinline__ void lds_fence_n(Frag (&f)[N]) {
  if constexpr (N == 4)
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi)
                 : "n"(NW));
  else
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi) : "n"(NW));
}
__device__ __forceinline__ bf16x8_t join(const Frag& f) {
  v8s v;
  v[0] = f.lo[0]; v[1] = f.lo[1]; v[2] = f.lo[2]; v[3] = f.lo[3];
  v[4] = f.hi[0]; v[5] = f.hi[1]; v[6] = f.hi[2]; v[7] = f.hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}

// 16-byte-chunk swizzle of an LDS row of CPR chunks.  A 32-lane group of a tr read touches
// rows {r0..r0+3, r0+8..r0+11} x two adjacent chunks: give those 8 rows distinct chunk
// pairs (CPR = 16: one 256-byte row spans all 64 banks) or distinct pairs per row parity
// (CPR = 8: two rows span the banks).
template <int CPR>
__device__ __forceinline__ int sw(int row);
template <>
__device__ __forceinline__ int sw<16>(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }
template <>
__device__ __forceinline__ int sw<8>(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

// A fragment = 8 consecutive rows (k = 8g .. 8g+7) of column (c0 + i) for lane l = 16 g + i, as two transpose reads (rows r and r + 4).
// byte offset (inside a [rows][TW] tile) of the fragment's first transpose read
template <int TW>
__device__ __forceinline__ uint32_t frag_off(int row0, int c0, int lane) {
  constexpr int CPR = TW / 8;
  const int g = lane >> 4, i = lane & 15;
  const int row = row0 + 8 * g + (i >> 2);
  const int col = c0 + 4 * (i & 3);
  return (uint32_t)(row * TW + (((col >> 3) ^ sw<CPR>(row)) << 3) + (col & 7)) * 2u;
}
// ... and of its second read (row + 4: the swizzle moves with the row, so this is a second per-lane constant unless
// row0 % 8 == 0, where it is the first offset + 4 TW elements)
template <int TW>
__device__ __forceinline__ uint32_t frag_off_hi(int row0, int c0, int lane) {
  constexpr int CPR = TW / 8;
  const int g = lane >> 4, i = lane & 15;
  const int row = row0 + 8 * g + (i >> 2) + 4;
  const int col = c0 + 4 * (i & 3);
  return (uint32_t)(row * TW + (((col >> 3) ^ sw<CPR>(row)) << 3) + (col & 7)) * 2u;
}
// the fragment at LDS byte addresses; row0 % 8 == 0 form: one address, the second read at a compile-time offset
template <int TW>
__device__ __forceinline__ Frag frag_at(uint32_t addr) {
  Frag f;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(addr), "n"(4 * TW * 2));
  return f;
}
__device__ __forceinline__ Frag frag_at2(uint32_t lo, uint32_t hi) {
  Frag f;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(lo));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(hi));
  return f;
}

struct WgP {
  const bf16_raw* x;
  const bf16_raw* dy;
  float* dw;
  float* dbias;
  const int* lengths;
  int B, T, Cin, Cout, ks, dil, pad, ldx, lddy, in_mask;
  int nCO, nCI, nsplit, tchunks;
  float* ws;  // [nsplit][ks][Cout][Cin] partials, or nullptr -> atomics into dw
};

// Batched form (ptpp_conv1d_wgrad_batched): up to WG_MAXP problems of ONE shape (the layers of a stack: B, T, Cin, Cout, ks,
// row strides shared; operands, dilation and targets per problem) in one launch.  With the tiles of all problems the grid
// fills the machine WITHOUT splitting the rows over blocks: every dw element has exactly one owner block, which walks all
// rows in a fixed order and adds its total straight into dw -- no partials, no second pass, bit-reproducible.
constexpr int WG_MAXP = 24;
struct WgProb {
  const bf16_raw* x;
  const bf16_raw* dy;
  float* dw;
  float* dbias;
  int dil, pad;
};
struct WgBatch {
  WgProb pr[WG_MAXP];
};

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XH: extra x rows of a tap group's window, (TG - 1) * dil <= XH
// OWNER: every dw element of the launch has exactly one owner block (nsplit = 1), which adds its total straight into dw;
// bid_in >= 0: the block's logical id inside its problem (grouped launch), else derived from blockIdx
template <int FM, int FN, int WR, int WC, int TG, int XH, bool BATCH, int NS, bool OWNER = false>
__device__ __forceinline__ void wgrad_body(const WgP& p, const int* __restrict__ lengths, const WgBatch* __restrict__ batch,
                                           int bid_in = -1);

// launch bounds: the second argument (minimum waves per SIMD) caps the allocation at 256 registers per lane.  Without it a
// 256-thread kernel may use all 512, the compiler then selects the AGPR form of the MFMAs while keeping the accumulator
// arrays in VGPRs, and EVERY MFMA of the 4-wave kernels was wrapped in four v_accvgpr_write, four v_accvgpr_read and their
// s_nops (16 moves per MFMA in the disassembly of round 2's <4, 4, 2, 2, 1, 0>: SQ_ACTIVE_INST_ANY 52 % of the wave cycles
// at 8 % MFMA-busy, profiles/r03_pmc_wgrad.txt) -- the 512-thread tap-group kernel never had them.
template <int FM, int FN, int WR = 2, int WC = 2, int TG = 1, int XH = 0>
__global__ __launch_bounds__(WR * WC * 64, 2) void conv1d_wgrad_bf16_kernel(const WgP p, const int* __restrict__ lengths) {
  wgrad_body<FM, FN, WR, WC, TG, XH, false, NS_DEFAULT>(p, lengths, nullptr);
}
// NS: a one-owner block walks ALL rows alone on its CU (the grid is <= 1 block per CU), so the ring is as deep as LDS allows:
// Little's law at ~1-2 us of loaded memory latency wants ~100 KB in flight per CU, the 4-stage ring holds 48-72 KB
template <int FM, int FN, int WR = 2, int WC = 2, int TG = 1, int XH = 0, int NS = NS_DEFAULT>
__global__ __launch_bounds__(WR * WC * 64, 2) void conv1d_wgrad_bf16_batched_kernel(const WgP p, const int* __restrict__ lengths,
                                                                                 const WgBatch batch) {
  wgrad_body<FM, FN, WR, WC, TG, XH, true, NS>(p, lengths, &batch);
}

// Grouped form (ptpp_conv1d_wgrad_grouped): up to WG_GMAX problems of DIFFERENT shapes (the linear / conv layers of one
// Conformer block: 256 x 256, 256 x 1024 x 9, ...) over few rows (phone level: ~3 k rows, where a split-K launch per layer is
// 30 us of fixed cost for 2 us of arithmetic) in one launch.  Block -> (problem, tap, ci tile, co tile) through the
// problems' first-block offsets; every block owns its dw tile and walks all rows of its problem.
constexpr int WG_GMAX = 16;
struct WgGProb {
  const bf16_raw* x;
  const bf16_raw* dy;
  float* dw;
  float* dbias;
  const int* lengths;  // non-null: rows of x at or after lengths[b] count as zero (in_mask)
  int B, T, Cin, Cout, ks, dil, pad, ldx, lddy, nCO, nCI, tchunks, blk0;
};
struct WgGroup {
  WgGProb pr[WG_GMAX];
  int nprob;
};
template <int FM, int FN, int WR = 2, int WC = 2, int TG = 1, int XH = 0>
__global__ __launch_bounds__(WR * WC * 64, 2) void conv1d_wgrad_bf16_grouped_kernel(const WgGroup g) {
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
  for (int i = 1; i < g.nprob; ++i)
    if (bid >= g.pr[i].blk0) pi = i;
  pi = __builtin_amdgcn_readfirstlane(pi);
  WgP p;
  p.x = g.pr[pi].x; p.dy = g.pr[pi].dy; p.dw = g.pr[pi].dw; p.dbias = g.pr[pi].dbias; p.lengths = g.pr[pi].lengths;
  p.B = g.pr[pi].B; p.T = g.pr[pi].T; p.Cin = g.pr[pi].Cin; p.Cout = g.pr[pi].Cout; p.ks = g.pr[pi].ks; p.dil = g.pr[pi].dil;
  p.pad = g.pr[pi].pad; p.ldx = g.pr[pi].ldx; p.lddy = g.pr[pi].lddy; p.in_mask = g.pr[pi].lengths != nullptr;
  p.nCO = g.pr[pi].nCO; p.nCI = g.pr[pi].nCI; p.tchunks = g.pr[pi].tchunks; p.nsplit = 1; p.ws = nullptr;
  wgrad_body<FM, FN, WR, WC, TG, XH, false, NS_DEFAULT, true>(p, p.lengths, nullptr, bid - g.pr[pi].blk0);
}

template <int FM, int FN, int WR, int WC, int TG, int XH, bool BATCH, int NS, bool OWNER>
__device__ __forceinline__ void wgrad_body(const WgP& p, const int* __restrict__ lengths, const WgBatch* __restrict__ batch, int bid_in) {
  constexpr int NW = WR * WC;  // waves per block, WR x WC over (co, ci)
  constexpr int TM = WR * FM * 16, TN = WC * FN * 16;
  constexpr int XR = KR + XH;                       // x rows per stage
  constexpr int CY = TM / 8, CX = TN / 8;          // 16-byte chunks per row
  constexpr int RY = 64 / CY, RX = 64 / CX;        // rows per 1 KiB piece
  constexpr int LY = KR / RY / NW, LX = XR / RX / NW;  // pieces per wave and chunk
  static_assert(LY >= 1 && LX >= 1 && LY * RY * NW == KR && LX * RX * NW == XR, "pieces must divide over the waves");
  constexpr int LPW = LY + LX;                     // LDS-DMA instructions per wave and chunk
  constexpr int STAGE = KR * TM + XR * TN;         // elements per ring stage
  extern __shared__ __attribute__((aligned(16))) char smem[];  // the ONLY LDS object (see header)
  bf16_raw* S = reinterpret_cast<bf16_raw*>(smem);  // [NS][dy: KR x TM | x: KR x TN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;

  int bid = OWNER ? bid_in : xcd_remap(blockIdx.x, gridDim.x);
  const int cot = bid % p.nCO; bid /= p.nCO;
  const int cit = bid % p.nCI; bid /= p.nCI;
  const int ntg = (p.ks + TG - 1) / TG;
  const int j = (bid % ntg) * TG; bid /= ntg;  // first tap of this block's group
  int split = bid;
  const bf16_raw* px = p.x;
  const bf16_raw* pdy = p.dy;
  float* pdw = p.dw;
  float* pdb = p.dbias;
  float* wsb = p.ws;
  int pdil = p.dil, ppad = p.pad;
  if constexpr (BATCH) {  // block-uniform problem index: the operands come from the table (scalar loads)
    const int pi = __builtin_amdgcn_readfirstlane(bid / p.nsplit);
    split = bid - pi * p.nsplit;
    px = batch->pr[pi].x; pdy = batch->pr[pi].dy; pdw = batch->pr[pi].dw; pdb = batch->pr[pi].dbias;
    pdil = batch->pr[pi].dil; ppad = batch->pr[pi].pad;
    // (round 6: a batched launch may split the rows too -- nsplit > 1, fixed chunk interleave -- its partials of problem pi live
    //  at ws + pi * nsplit * (ks Cout Cin + Cout) and wgrad_reduce_batched_kernel adds them in split order)
    if (p.ws) wsb = p.ws + (int64_t)pi * p.nsplit * ((int64_t)p.ks * p.Cout * p.Cin + p.Cout);
  }
  const int co0 = cot * TM, ci0 = cit * TN;
  const int shift = j * pdil - ppad;
  const int ntap = min(TG, p.ks - j);  // taps of the group (block-uniform)

  f32x4 acc[TG][FM][FN], accb[FM];
#pragma unroll
  for (int a = 0; a < FM; ++a) {
    accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < TG; ++g)
#pragma unroll
      for (int c = 0; c < FN; ++c) acc[g][a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = pdb && cit == 0 && j == 0 && wc == 0;  // wave-uniform
  const int total = p.B * p.tchunks;
  const int n = split < total ? (total - split + p.nsplit - 1) / p.nsplit : 0;  // chunks of this block
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + lane * 16;

  // per-lane source coordinates inside a chunk (fixed for the whole loop)
  int yrow[LY], ycol[LY], xrow[LX], xcol[LX];
#pragma unroll
  for (int q = 0; q < LY; ++q) {
    const int pi = wave * LY + q;
    yrow[q] = pi * RY + lane / CY;
    ycol[q] = ((lane % CY) ^ sw<CY>(yrow[q])) * 8;
  }
#pragma unroll
  for (int q = 0; q < LX; ++q) {
    const int pi = wave * LX + q;
    xrow[q] = pi * RX + lane / CX;
    xcol[q] = ((lane % CX) ^ sw<CX>(xrow[q])) * 8;
  }

  // The loop's bookkeeping is kept off the issue ports (round 4 spent ~105 scalar + ~70 vector instructions per chunk on it --
  // a division, the lengths load, 64-bit lane addresses, the fragment swizzles -- against 24 MFMAs: two waves per SIMD then
  // need ~2100 cycles per chunk for 768 cycles of matrix work, profiles/r05_wgrad.txt).  Chunks are ISSUED in order, so the
  // issue side keeps (utterance, chunk in utterance, ring slot) incrementally; an interior chunk (all rows of both operands
  // inside the utterance, all columns inside the tensors) is LPW loads from a wave-uniform base plus a per-lane offset that
  // never changes; only chunks touching an utterance's edge compute lane addresses and pick the zero page.
  int ib, itc;  // utterance and chunk-in-utterance of the NEXT chunk to issue
  {
    ib = split / p.tchunks;
    itc = split - ib * p.tchunks;
  }
  int islot = 0;
  int iTin = 0, iTin_b = -1;
  // row pointers of the next chunk to issue (utterance ib, row itc * KR), advanced by additions only
  const char* yptr = reinterpret_cast<const char*>(pdy + ((int64_t)ib * p.T + (int64_t)itc * KR) * p.lddy + co0);
  const char* xptr = reinterpret_cast<const char*>(px + ((int64_t)ib * p.T + (int64_t)itc * KR + shift) * p.ldx + ci0);
  const int64_t ystep = (int64_t)p.nsplit * KR * p.lddy * 2, xstep = (int64_t)p.nsplit * KR * p.ldx * 2;
  const int wrap_rows = p.T - p.tchunks * KR;  // row correction when the chunk index wraps into the next utterance (<= 0)
  const int64_t ywrap = (int64_t)wrap_rows * p.lddy * 2, xwrap = (int64_t)wrap_rows * p.ldx * 2;
  uint32_t yoff[LY], xoff[LX];
#pragma unroll
  for (int q = 0; q < LY; ++q) yoff[q] = (uint32_t)(yrow[q] * p.lddy + ycol[q]) * 2u;
#pragma unroll
  for (int q = 0; q < LX; ++q) xoff[q] = (uint32_t)(xrow[q] * p.ldx + xcol[q]) * 2u;
  const bool cols_in = co0 + TM <= p.Cout && ci0 + TN <= p.Cin;  // block-uniform
  const uint32_t s_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) bf16_raw*)S;
  auto issue = [&]() {  // the next chunk of this block -> ring slot islot
    const int b = ib, tb = itc * KR;
    if (b != iTin_b) {  // (scalar load: once per utterance)
      iTin = lengths ? min(lengths[b], p.T) : p.T;
      iTin_b = b;
    }
    const int Tin = iTin;
    const uint32_t st_lds = s_lds + (uint32_t)islot * (uint32_t)(STAGE * 2);
    const bool interior = cols_in && tb + KR <= p.T && tb + shift >= 0 && tb + shift + XR <= Tin;
    if (interior) {
      glds16_s_n<LY>(yptr, yoff, st_lds + (uint32_t)(wave * LY * 1024));
      glds16_s_n<LX>(xptr, xoff, st_lds + (uint32_t)(KR * TM * 2 + wave * LX * 1024));
    } else {
      const bf16_raw* dyb = pdy + (int64_t)b * p.T * p.lddy + co0;
      const bf16_raw* xb = px + (int64_t)b * p.T * p.ldx + ci0;
#pragma unroll
      for (int q = 0; q < LY; ++q) {
        const int t = tb + yrow[q];
        const bool ok = t < p.T && co0 + ycol[q] < p.Cout;
        const char* src = ok ? reinterpret_cast<const char*>(dyb + (int64_t)t * p.lddy + ycol[q]) : zero;
        glds16(src, __builtin_amdgcn_readfirstlane(st_lds + (uint32_t)((wave * LY + q) * 1024)));
      }
#pragma unroll
      for (int q = 0; q < LX; ++q) {
        const int ts = tb + xrow[q] + shift;  // (rows paired only with dy rows past T meet zeros there)
        const bool ok = ts >= 0 && ts < Tin && ci0 + xcol[q] < p.Cin;
        const char* src = ok ? reinterpret_cast<const char*>(xb + (int64_t)ts * p.ldx + xcol[q]) : zero;
        glds16(src, __builtin_amdgcn_readfirstlane(st_lds + (uint32_t)(KR * TM * 2 + (wave * LX + q) * 1024)));
      }
    }
    islot = islot + 1 == NS ? 0 : islot + 1;
    itc += p.nsplit;
    yptr += ystep;
    xptr += xstep;
    while (itc >= p.tchunks) {
      itc -= p.tchunks;
      ++ib;
      yptr += ywrap;
      xptr += xwrap;
    }
  };

  // fragment addresses: one LDS byte offset per lane and fragment, relative to the stage (they never change); the second
  // transpose read of a fragment is 4 rows further = a compile-time offset
  uint32_t offA[FM], offB[TG][FN], offBh[TG][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a) offA[a] = frag_off<TM>(0, (wr * FM + a) * 16, lane);
#pragma unroll
  for (int g = 0; g < TG; ++g)
#pragma unroll
    for (int c = 0; c < FN; ++c) {
      offB[g][c] = (uint32_t)(KR * TM * 2) + frag_off<TN>(g * pdil, (wc * FN + c) * 16, lane);
      offBh[g][c] = (uint32_t)(KR * TM * 2) + frag_off_hi<TN>(g * pdil, (wc * FN + c) * 16, lane);
    }

  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, v8s{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < n) issue();
  uint32_t cslot_lds = s_lds;  // ring slot of the chunk being multiplied
  // the whole row loop exists twice: FULL = all TG taps of the group exist (the common case: no tap tests, the accumulators
  // never move between the two forms' register assignments)
  auto kloop = [&](auto full_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
  auto chunk = [&](int i, auto steady_c) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_c)::value;  // NS - 2 younger chunks in flight and one more to issue: no tests
    // chunk i has landed once at most the loads of the (up to NS - 2) younger chunks are outstanding
    static_assert((NS - 2) * LPW <= 63, "vmcnt is a 6-bit counter");
    if constexpr (STEADY) {
      wait_vm<(NS - 2) * LPW>();
    } else {  // drain at the end of the block's rows
      const int younger = min(n - 1 - i, NS - 2);
      if (younger == NS - 2) wait_vm<(NS - 2) * LPW>();
      else if constexpr (NS > 3) {
        if (younger >= 2) wait_vm<2 * LPW>();  // (waits for more than necessary for 2 < younger < NS - 2: tail only)
        else if (younger == 1) wait_vm<LPW>();
        else wait_vm<0>();
      } else {
        if (younger == 1) wait_vm<LPW>();
        else wait_vm<0>();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's pieces of chunk i visible; everyone done reading chunk i - 1
    if (STEADY || i + NS - 1 < n) issue();  // into the stage chunk i - 1 just vacated
    // One exposed LDS round trip per chunk: the dy fragments and the first tap's x fragments are requested together; every
    // further tap's fragments are requested BEFORE the previous tap's MFMAs and waited for with a counted lgkmcnt (the
    // requests retire in order: 2 FN transpose reads of the younger tap may stay outstanding).  Round 4 fenced each tap's
    // reads right after issuing them: 1 + TG round trips per chunk at 24 MFMAs per wave (profiles/r05_wgrad.txt).
    Frag fa[FM], fb[2][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a) fa[a] = frag_at<TM>(cslot_lds + offA[a]);
#pragma unroll
    for (int c = 0; c < FN; ++c) fb[0][c] = frag_at2(cslot_lds + offB[0][c], cslot_lds + offBh[0][c]);
    lds_fence(fa);
    bf16x8_t af[FM];
#pragma unroll
    for (int a = 0; a < FM; ++a) af[a] = join(fa[a]);
    {
#pragma unroll
      for (int g = 0; g < TG; ++g) {
        if (FULL || g < ntap) {
          const bool more = g + 1 < TG && (FULL || g + 1 < ntap);
          if (g + 1 < TG) {
            if (more) {
#pragma unroll
              for (int c = 0; c < FN; ++c) fb[(g + 1) & 1][c] = frag_at2(cslot_lds + offB[g + 1 < TG ? g + 1 : 0][c], cslot_lds + offBh[g + 1 < TG ? g + 1 : 0][c]);
            }
          }
          if (more) lds_fence_n<2 * FN>(fb[g & 1]);
          else lds_fence(fb[g & 1]);
          bf16x8_t bfr[FN];
#pragma unroll
          for (int c = 0; c < FN; ++c) bfr[c] = join(fb[g & 1][c]);
#pragma unroll
          for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int c = 0; c < FN; ++c)
              acc[g][a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[c], acc[g][a][c], 0, 0, 0);
        }
      }
    }
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < FM; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], ones, accb[a], 0, 0, 0);
    }
    cslot_lds = cslot_lds + (uint32_t)(STAGE * 2) == s_lds + (uint32_t)(NS * STAGE * 2) ? s_lds : cslot_lds + (uint32_t)(STAGE * 2);
  };
    const int nsteady = n - (NS - 1);  // chunks i with i + NS - 1 < n
    int i = 0;
    for (; i < nsteady; ++i) chunk(i, std::true_type{});
    for (; i < n; ++i) chunk(i, std::false_type{});
  };
  if (ntap == TG) kloop(std::true_type{});
  else kloop(std::false_type{});

  // D[i = co (rows 4*(lane>>4) + r)][j = ci (col lane&15)]
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int g = 0; g < TG; ++g) {
    if (g >= ntap) break;
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int c = 0; c < FN; ++c) {
        const int ci = ci0 + (wc * FN + c) * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + (wr * FM + a) * 16 + lg * 4 + r;
          if (co < p.Cout && ci < p.Cin) {
            if (OWNER || (BATCH && p.nsplit == 1)) pdw[((int64_t)co * p.Cin + ci) * p.ks + j + g] += acc[g][a][c][r];  // the only owner
            else if (wsb) wsb[(((int64_t)split * p.ks + j + g) * p.Cout + co) * p.Cin + ci] = acc[g][a][c][r];
            else atomicAdd(pdw + ((int64_t)co * p.Cin + ci) * p.ks + j + g, acc[g][a][c][r]);
          }
        }
      }
  }
  if (do_bias && lr == 0) {  // every column of accb holds the row sums
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + (wr * FM + a) * 16 + lg * 4 + r;
        if (co < p.Cout) {
          if (OWNER || (BATCH && p.nsplit == 1)) pdb[co] += accb[a][r];  // one block per (problem, co tile) reaches here
          else if (wsb) wsb[(int64_t)p.nsplit * p.ks * p.Cout * p.Cin + (int64_t)split * p.Cout + co] = accb[a][r];
          else atomicAdd(pdb + co, accb[a][r]);
        }
      }
  }
}

// dw[co][ci][j] += sum_s ws[s][j][co][ci]  (fixed summation order: deterministic); the bias partials [nsplit][Cout] follow
// the weight partials in the workspace and are summed by the blocks past the weight range (round 2 added them with f32
// atomics from every split block: the last bits of the bias gradients depended on the arrival order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           float* __restrict__ dbias, int nsplit, int Cout, int Cin, int ks) {
  const int64_t E = (int64_t)ks * Cout * Cin;
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= E) {
    const int64_t co = e - E;
    if (dbias && co < Cout) {
      const float* b = ws + (int64_t)nsplit * E + co;
      f32x4 s = *reinterpret_cast<const f32x4*>(b);
      for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(b + (int64_t)k * Cout);
#pragma unroll
      for (int k = 0; k < 4; ++k) dbias[co + k] += s[k];  // (a gradient view may start at any float offset of the flat buffer)
    }
    return;
  }
  f32x4 s = *reinterpret_cast<const f32x4*>(ws + e);
  for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(ws + (int64_t)k * E + e);
  const int j = (int)(e / ((int64_t)Cout * Cin));
  const int64_t rem = e - (int64_t)j * Cout * Cin;  // co * Cin + ci  (Cin % 4 == 0: the 4 share co)
  float* d = dw + rem * ks + j;
#pragma unroll
  for (int k = 0; k < 4; ++k) d[(int64_t)k * ks] += s[k];
}

// the same for the problems of a batched launch whose rows were split (blockIdx.y = problem; targets from the table)
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const float* __restrict__ ws, const WgBatch batch, int nsplit, int Cout,
                                                                   int Cin, int ks) {
  const int64_t E = (int64_t)ks * Cout * Cin;
  const int pi = blockIdx.y;
  const float* wsp = ws + (int64_t)pi * nsplit * (E + Cout);
  float* dw = batch.pr[pi].dw;
  float* dbias = batch.pr[pi].dbias;
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= E) {
    const int64_t co = e - E;
    if (dbias && co < Cout) {
      const float* b = wsp + (int64_t)nsplit * E + co;
      f32x4 s = *reinterpret_cast<const f32x4*>(b);
      for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(b + (int64_t)k * Cout);
#pragma unroll
      for (int k = 0; k < 4; ++k) dbias[co + k] += s[k];
    }
    return;
  }
  f32x4 s = *reinterpret_cast<const f32x4*>(wsp + e);
  for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(wsp + (int64_t)k * E + e);
  const int j = (int)(e / ((int64_t)Cout * Cin));
  const int64_t rem = e - (int64_t)j * Cout * Cin;
  float* d = dw + rem * ks + j;
#pragma unroll
  for (int k = 0; k < 4; ++k) d[(int64_t)k * ks] += s[k];
}

template <int FM, int FN, int WR = 2, int WC = 2, int TG = 1, int XH = 0>
int launch(WgP& p, size_t ws_bytes, hipStream_t st) {
  constexpr int TM = WR * FM * 16, TN = WC * FN * 16;
  constexpr int XR = KR + XH;
  p.nCO = (p.Cout + TM - 1) / TM;
  p.nCI = (p.Cin + TN - 1) / TN;
  p.tchunks = (p.T + KR - 1) / KR;
  const int total = p.B * p.tchunks;
  const int tiles = p.nCO * p.nCI * ((p.ks + TG - 1) / TG);
  const size_t ebytes = ((size_t)p.ks * p.Cout * p.Cin + p.Cout) * sizeof(float);  // weight + bias partials of one split
  int nsplit;
  if (p.ws && ws_bytes >= ebytes && p.Cout % 4 == 0) {
    nsplit = ((TG > 1 ? 256 : 384) + tiles - 1) / tiles;  // 1-2 resident blocks per CU (measured optimum 256..512 blocks;
                                                           // a tap-group block fills a CU on its own)
    if (nsplit > 48) nsplit = 48;        // few output tiles: more splits only add partial traffic
    if ((size_t)nsplit * ebytes > ws_bytes) nsplit = (int)(ws_bytes / ebytes);
    if (nsplit > (total + 7) / 8) nsplit = (total + 7) / 8;
  } else {
    p.ws = nullptr;
    // atomics: ~80 G/s device-wide, so few splits -- about 2 M atomics per launch
    nsplit = (int)((size_t)(8u << 20) / ebytes);
    if (nsplit > (512 + tiles - 1) / tiles) nsplit = (512 + tiles - 1) / tiles;
    if (nsplit > (total + 3) / 4) nsplit = (total + 3) / 4;
  }
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
  const size_t smem = (size_t)NS_DEFAULT * (KR * TM + XR * TN) * sizeof(bf16_raw);
  auto kern = conv1d_wgrad_bf16_kernel<FM, FN, WR, WC, TG, XH>;
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_wgrad(bf16)")) return PTPP_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)tiles * nsplit)), dim3(WR * WC * 64), smem, st, p,
                     p.in_mask ? p.lengths : nullptr);
  PTPP_CHECK_LAUNCH("conv1d_wgrad(bf16)");
  if (p.ws) {
    const int64_t n4 = ((int64_t)p.ks * p.Cout * p.Cin + (p.dbias ? p.Cout : 0)) / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, p.ws, p.dw, p.dbias, nsplit,
                       p.Cout, p.Cin, p.ks);
    PTPP_CHECK_LAUNCH("conv1d_wgrad(reduce)");
  }
  return PTPP_OK;
}

template <int FM, int FN, int WR = 2, int WC = 2, int TG = 1, int XH = 0, int NS = NS_DEFAULT>
int launch_batched(WgP& p, const WgBatch& batch, int nprob, int nsplit, float* ws, hipStream_t st) {
  constexpr int TM = WR * FM * 16, TN = WC * FN * 16;
  constexpr int XR = KR + XH;
  p.nCO = (p.Cout + TM - 1) / TM;
  p.nCI = (p.Cin + TN - 1) / TN;
  p.tchunks = (p.T + KR - 1) / KR;
  p.nsplit = nsplit > 1 && ws ? nsplit : 1;
  p.ws = p.nsplit > 1 ? ws : nullptr;
  const int tiles = p.nCO * p.nCI * ((p.ks + TG - 1) / TG);
  const size_t smem = (size_t)NS * (KR * TM + XR * TN) * sizeof(bf16_raw);
  static_assert((size_t)NS * (KR * TM + XR * TN) * sizeof(bf16_raw) <= 160 * 1024, "ring must fit LDS");
  auto kern = conv1d_wgrad_bf16_batched_kernel<FM, FN, WR, WC, TG, XH, NS>;
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_wgrad(bf16)")) return PTPP_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)((int64_t)tiles * nprob * p.nsplit)), dim3(WR * WC * 64), smem, st, p,
                     p.in_mask ? p.lengths : nullptr, batch);
  if (p.nsplit > 1) {
    const int64_t E4 = ((int64_t)p.ks * p.Cout * p.Cin + p.Cout + 3) / 4;
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)((E4 + 255) / 256), (unsigned)nprob), dim3(256), 0, st, ws, batch,
                       p.nsplit, p.Cout, p.Cin, p.ks);
  }
  PTPP_CHECK_LAUNCH("conv1d_wgrad_batched(bf16)");
  return PTPP_OK;
}

}  // namespace

// Tiles per problem of the batched launch (0: shape not served by the batched kernels) -- the caller batches only when
// nprob * tiles fills the machine without a row split.
int ptpp_wgrad_bf16_batched_tiles(int Cin, int Cout, int ks, int max_dil) {
  if (Cout <= 64 || Cin <= 64) return 0;
  if (ks > 1 && 2 * max_dil <= 32) return ((Cout + 127) / 128) * ((Cin + 127) / 128) * ((ks + 2) / 3);
  return ((Cout + 127) / 128) * ((Cin + 127) / 128) * ks;
}

// called by ptpp_conv1d_wgrad_batched for bf16 tensors with 16-byte aligned rows; nprob <= WG_MAXP
int ptpp_wgrad_bf16_launch_batched(const ptpp_wgrad_problem* probs, int nprob, const int32_t* lengths, int B, int T, int Cin,
                                   int Cout, int ks, int ldx, int lddy, int in_mask, int nsplit, float* ws, hipStream_t st) {
  WgP p;
  p.x = nullptr; p.dy = nullptr; p.dw = nullptr; p.dbias = nullptr; p.lengths = lengths;
  p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.ks = ks; p.dil = 1; p.pad = 0; p.ldx = ldx; p.lddy = lddy;
  p.in_mask = in_mask;
  WgBatch batch;
  int max_dil = 1;
  for (int i = 0; i < nprob; ++i) {
    batch.pr[i].x = (const bf16_raw*)probs[i].x; batch.pr[i].dy = (const bf16_raw*)probs[i].dy;
    batch.pr[i].dw = probs[i].dw; batch.pr[i].dbias = probs[i].dbias;
    batch.pr[i].dil = probs[i].dil; batch.pr[i].pad = probs[i].pad;
    if (probs[i].dil > max_dil) max_dil = probs[i].dil;
  }
  // (deeper rings for these one-block-per-CU launches -- 6 x 24 KB / 9 x 16 KB instead of 4 stages -- measured equal:
  //  959 vs 1008 us for the 20 k = 3 layers, 1025 vs 1027 us for k = 1, profiles/r03_wgrad_batched.txt; PTPP_WGRAD_DEEP=1)
  static const char* deep = getenv("PTPP_WGRAD_DEEP");
  if (deep && deep[0] == '1') {
    if (ks > 1 && 2 * max_dil <= 32) return launch_batched<4, 2, 2, 4, 3, 32, 6>(p, batch, nprob, nsplit, ws, st);
    return launch_batched<4, 4, 2, 2, 1, 0, 9>(p, batch, nprob, nsplit, ws, st);
  }
  if (ks > 1 && 2 * max_dil <= 32) return launch_batched<4, 2, 2, 4, 3, 32>(p, batch, nprob, nsplit, ws, st);
  return launch_batched<4, 4>(p, batch, nprob, nsplit, ws, st);
}

// called by ptpp_conv1d_wgrad_grouped for bf16 problems with 16-byte aligned rows; nprob <= 16.  Two launches at most: the
// problems with taps go to the tap-group instantiation (three taps share a block's dy chunk and x window: a k = 9 layer
// moves a third of the operand bytes of nine one-tap blocks), the 1 x 1 problems to the one-tap instantiation.
template <int FM, int FN, int WR, int WC, int TG, int XH>
static int launch_grouped(const ptpp_wgrad_gproblem* const* probs, int nprob, hipStream_t st) {
  constexpr int TM = WR * FM * 16, TN = WC * FN * 16, XR = KR + XH;
  WgGroup g;
  g.nprob = nprob;
  int64_t nblk = 0;
  for (int i = 0; i < nprob; ++i) {
    const ptpp_wgrad_gproblem& q = *probs[i];
    WgGProb& d = g.pr[i];
    d.x = (const bf16_raw*)q.x; d.dy = (const bf16_raw*)q.dy; d.dw = q.dw; d.dbias = q.dbias; d.lengths = q.lengths;
    d.B = q.B; d.T = q.T;
    if (q.ks == 1 && !q.lengths && q.B > 1) { d.T = q.B * q.T; d.B = 1; }  // (as ptpp_conv1d_wgrad: one flat row sequence)
    d.Cin = q.Cin; d.Cout = q.Cout; d.ks = q.ks; d.dil = q.dil; d.pad = q.pad; d.ldx = q.ldx; d.lddy = q.lddy;
    d.nCO = (q.Cout + TM - 1) / TM; d.nCI = (q.Cin + TN - 1) / TN; d.tchunks = (d.T + KR - 1) / KR;
    d.blk0 = (int)nblk;
    nblk += (int64_t)d.nCO * d.nCI * ((q.ks + TG - 1) / TG);
  }
  const size_t smem = (size_t)NS_DEFAULT * (KR * TM + XR * TN) * sizeof(bf16_raw);
  auto kern = conv1d_wgrad_bf16_grouped_kernel<FM, FN, WR, WC, TG, XH>;
  if (smem > 64 * 1024 && !ptpp_lds_limit(reinterpret_cast<const void*>(kern), (int)smem, "conv1d_wgrad(bf16)")) return PTPP_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(WR * WC * 64), smem, st, g);
  PTPP_CHECK_LAUNCH("conv1d_wgrad_grouped(bf16)");
  return PTPP_OK;
}
int ptpp_wgrad_bf16_launch_grouped(const ptpp_wgrad_gproblem* probs, int nprob, hipStream_t st, hipStream_t st2) {
  const ptpp_wgrad_gproblem* taps[WG_GMAX];
  const ptpp_wgrad_gproblem* flat[WG_GMAX];
  int nt = 0, nf = 0;
  for (int i = 0; i < nprob; ++i) {
    if (probs[i].ks > 1 && 2 * probs[i].dil <= 32) taps[nt++] = probs + i;
    else flat[nf++] = probs + i;
  }
  if (nt) {
    const int rc = launch_grouped<4, 2, 2, 4, 3, 32>(taps, nt, st);
    if (rc != PTPP_OK) return rc;
  }
  if (nf) return launch_grouped<4, 4, 2, 2, 1, 0>(flat, nf, st2 ? st2 : st);
  return PTPP_OK;
}

// called by ptpp_conv1d_wgrad for bf16 tensors with 16-byte aligned rows
int ptpp_wgrad_bf16_launch(const void* x, const void* dy, float* dw, float* dbias, const int32_t* lengths, int B, int T,
                           int Cin, int Cout, int ks, int dil, int pad, int ldx, int lddy, int in_mask, void* ws,
                           size_t ws_bytes, hipStream_t st) {
  WgP p;
  p.ws = ((uintptr_t)ws % 16) == 0 ? (float*)ws : nullptr;
  p.x = (const bf16_raw*)x; p.dy = (const bf16_raw*)dy; p.dw = dw; p.dbias = dbias; p.lengths = lengths;
  p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.ks = ks; p.dil = dil; p.pad = pad; p.ldx = ldx; p.lddy = lddy;
  p.in_mask = in_mask;
  const bool bigM = Cout > 64, bigN = Cin > 64;
  // (8 waves of 32 x 64 / 64 x 32 measured equal to 4 waves of 64 x 64 here: this kernel is not latency-bound)
  // several taps per block when their common x window fits the stage (see the header); PTPP_WGRAD_TAPS=1 keeps one
  static const char* tg = getenv("PTPP_WGRAD_TAPS");
  // (8 waves of 64 x 32 with three taps: 57 us against 87 us for the 256->512 k=3 layer over 30 k rows; 4 waves of
  //  64 x 64 with two or three taps spend their time moving accumulators between the register files and lose)
  if (bigM && bigN && ks > 1 && 2 * dil <= 32 && !(tg && tg[0] == '1')) return launch<4, 2, 2, 4, 3, 32>(p, ws_bytes, st);
  if (bigM && bigN) return launch<4, 4>(p, ws_bytes, st);
  if (bigM) return launch<4, 2>(p, ws_bytes, st);
  if (bigN) return launch<2, 4>(p, ws_bytes, st);
  return launch<2, 2>(p, ws_bytes, st);
}
