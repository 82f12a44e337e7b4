// Weight / bias gradient of the channels-last Conv1d (see conv1d_cl.hip):
//   dw[co][ci][j] += sum_{b,t} dy[b,t,co] * x[b, t + j*dil - pad, ci]
//   dbias[co]     += sum_{b,t} dy[b,t,co]
// GEMM per tap: M = Cout, N = Cin, K = B*T (the reduction runs over ROWS).  Both
// operands are channels-last, i.e. K-major, which is exactly the operand layout
// of v_mfma_f32_16x16x4_f32 (lane l holds A[l&15][l>>4]): 16 lanes read 16
// consecutive channels of one row, conflict-free, no transposes.  Rows are split
// across blocks (split-K) and combined with f32 atomics, so the result is f32
// regardless of the activation dtype.
//
// This file holds the exact-f32 kernel (f32 activations, and the rare bf16 shapes with
// unaligned rows, which are widened while staging).  bf16 tensors normally take the
// bf16-rate kernel in conv1d_wgrad_bf16.hip (ds_read_b64_tr_b16 operand transposes).
#include "ptpp_common.h"

namespace {

constexpr int KR = 32;    // rows per K chunk
constexpr int TS = 80;    // LDS row stride in floats (64 + 16: conflict-free b32 reads)

template <typename T>
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* __restrict__ dw, float* __restrict__ dbias,
                                                           const int* __restrict__ lengths, int B, int T_, int Cin,
                                                           int Cout, int ks, int dil, int pad, int ldx, int lddy,
                                                           int in_mask, int nCO, int nCI, int nsplit, int tchunks) {
  __shared__ float dYs[KR * TS];
  __shared__ float Xs[KR * TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  int bid = blockIdx.x;
  const int cot = bid % nCO; bid /= nCO;
  const int cit = bid % nCI; bid /= nCI;
  const int j = bid % ks;    bid /= ks;
  const int split = bid;
  const int co0 = cot * 64, ci0 = cit * 64;
  const int shift = j * dil - pad;

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = dbias && cit == 0 && j == 0;

  const int total = B * tchunks;
  for (int ch = split; ch < total; ch += nsplit) {
    const int b = ch / tchunks, tb = (ch % tchunks) * KR;
    const int Tin = (in_mask && lengths) ? min(lengths[b], T_) : T_;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 4, c4 = (idx & 15) * 4;
      const int t = tb + row;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < T_) {
        const T* src = dy + ((int64_t)b * T_ + t) * lddy + co0 + c4;
        if (co0 + c4 + 3 < Cout) v = Elem<T>::ld4(src);
        else
          for (int e = 0; e < 4; ++e)
            if (co0 + c4 + e < Cout) v[e] = Elem<T>::ld(src + e);
      }
      *reinterpret_cast<f32x4*>(&dYs[row * TS + c4]) = v;
      f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
      const int ts = t + shift;
      if (t < T_ && ts >= 0 && ts < Tin) {
        const T* src = x + ((int64_t)b * T_ + ts) * ldx + ci0 + c4;
        if (ci0 + c4 + 3 < Cin) u = Elem<T>::ld4(src);
        else
          for (int e = 0; e < 4; ++e)
            if (ci0 + c4 + e < Cin) u[e] = Elem<T>::ld(src + e);
      }
      *reinterpret_cast<f32x4*>(&Xs[row * TS + c4]) = u;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KR / 4; ++kk) {
      const int k = kk * 4 + lg;
      float af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = dYs[k * TS + wr * 32 + a * 16 + lr];
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[c] = Xs[k * TS + wc * 32 + c * 16 + lr];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[c], acc[a][c], 0, 0, 0);
    }
    if (do_bias && tid < 64) {
#pragma unroll 8
      for (int r = 0; r < KR; ++r) bsum += dYs[r * TS + tid];
    }
    __syncthreads();
  }

  // D[i = co (rows 4*lg + r)][j = ci (col lr)]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ci = ci0 + wc * 32 + c * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wr * 32 + a * 16 + lg * 4 + r;
        if (co < Cout && ci < Cin) atomicAdd(dw + ((int64_t)co * Cin + ci) * ks + j, acc[a][c][r]);
      }
    }
  if (do_bias && tid < 64 && co0 + tid < Cout) atomicAdd(dbias + co0 + tid, bsum);
}

}  // namespace

int ptpp_wgrad_bf16_launch(const void* x, const void* dy, float* dw, float* dbias, const int32_t* lengths, int B, int T,
                           int Cin, int Cout, int ks, int dil, int pad, int ldx, int lddy, int in_mask, void* ws,
                           size_t ws_bytes, hipStream_t st);

extern "C" int ptpp_conv1d_wgrad(const void* x, const void* dy, float* dw, float* dbias, const int32_t* lengths, int B,
                                 int T, int Cin, int Cout, int ks, int dil, int pad, int ldx, int lddy, int in_mask,
                                 int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  PTPP_CHECK_ARG(x && dy && dw, "conv1d_wgrad: null pointer");
  PTPP_CHECK_ARG(B > 0 && T > 0 && Cin > 0 && Cout > 0 && ks > 0 && dil > 0, "conv1d_wgrad: bad shape");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "conv1d_wgrad: bad dtype %d", dtype);
  PTPP_CHECK_ARG(ldx % 4 == 0 && lddy % 4 == 0, "conv1d_wgrad: row strides must be multiples of 4");
  PTPP_CHECK_ARG(!in_mask || lengths, "conv1d_wgrad: in_mask needs lengths");
  // ks = 1 without an input mask: the reduction runs over rows, utterance boundaries do not matter -- one
  // flat sequence (rows are linear in memory) instead of a padded 32-row chunk per short utterance
  if (ks == 1 && !in_mask && B > 1) {
    T = B * T;
    B = 1;
  }
  if (dtype == PTPP_BF16 && ldx % 8 == 0 && lddy % 8 == 0 && Cin % 8 == 0 && Cout % 8 == 0 &&
      ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0) {
    // bf16-rate path (LDS transpose reads)
    return ptpp_wgrad_bf16_launch(x, dy, dw, dbias, lengths, B, T, Cin, Cout, ks, dil, pad, ldx, lddy, in_mask, workspace,
                                  workspace_bytes, reinterpret_cast<hipStream_t>(stream));
  }
  const int nCO = (Cout + 63) / 64, nCI = (Cin + 63) / 64;
  const int tchunks = (T + KR - 1) / KR;
  const int total = B * tchunks;
  const int tiles = nCO * nCI * ks;
  int nsplit = (2048 + tiles - 1) / tiles;  // aim for ~8 blocks per CU
  if (nsplit > total) nsplit = total;
  if (nsplit < 1) nsplit = 1;
  dim3 grid((unsigned)((int64_t)tiles * nsplit)), blk(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(conv1d_wgrad_kernel<float>, grid, blk, 0, st, (const float*)x, (const float*)dy, dw, dbias,
                       lengths, B, T, Cin, Cout, ks, dil, pad, ldx, lddy, in_mask, nCO, nCI, nsplit, tchunks);
  else
    hipLaunchKernelGGL(conv1d_wgrad_kernel<bf16_raw>, grid, blk, 0, st, (const bf16_raw*)x, (const bf16_raw*)dy, dw,
                       dbias, lengths, B, T, Cin, Cout, ks, dil, pad, ldx, lddy, in_mask, nCO, nCI, nsplit, tchunks);
  PTPP_CHECK_LAUNCH("conv1d_wgrad");
  return PTPP_OK;
}

int ptpp_wgrad_bf16_batched_tiles(int Cin, int Cout, int ks, int max_dil);
int ptpp_wgrad_bf16_launch_batched(const ptpp_wgrad_problem* probs, int nprob, const int32_t* lengths, int B, int T, int Cin,
                                   int Cout, int ks, int ldx, int lddy, int in_mask, int nsplit, float* ws, hipStream_t st);

extern "C" int ptpp_conv1d_wgrad_batched(const ptpp_wgrad_problem* probs, int nprob, const int32_t* lengths, int B, int T, int Cin,
                                         int Cout, int ks, int ldx, int lddy, int in_mask, int dtype, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  PTPP_CHECK_ARG(probs && nprob > 0, "conv1d_wgrad_batched: no problems");
  PTPP_CHECK_ARG(B > 0 && T > 0 && Cin > 0 && Cout > 0 && ks > 0, "conv1d_wgrad_batched: bad shape");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "conv1d_wgrad_batched: bad dtype %d", dtype);
  PTPP_CHECK_ARG(!in_mask || lengths, "conv1d_wgrad_batched: in_mask needs lengths");
  bool fast = dtype == PTPP_BF16 && ldx % 8 == 0 && lddy % 8 == 0 && Cin % 8 == 0 && Cout % 8 == 0;
  int max_dil = 1;
  for (int i = 0; i < nprob; ++i) {
    PTPP_CHECK_ARG(probs[i].x && probs[i].dy && probs[i].dw && probs[i].dil > 0, "conv1d_wgrad_batched: bad problem %d", i);
    fast = fast && ((uintptr_t)probs[i].x % 16) == 0 && ((uintptr_t)probs[i].dy % 16) == 0;
    if (probs[i].dil > max_dil) max_dil = probs[i].dil;
  }
  static const char* off = getenv("PTPP_WGRAD_NO_BATCH");
  const int tiles = fast && !(off && off[0] == '1') ? ptpp_wgrad_bf16_batched_tiles(Cin, Cout, ks, max_dil) : 0;
  // one owner block per tile walks ALL rows: worth it only when the tiles of the batch occupy most of the 256 CUs
  if (tiles > 0 && (long long)tiles * nprob >= 128) {
    int Bf = B, Tf = T;
    if (ks == 1 && !in_mask && B > 1) { Tf = B * T; Bf = 1; }  // (as ptpp_conv1d_wgrad: one flat row sequence)
    constexpr int MAXP = 24;
    // Round 6 (opt-in, PTPP_WGRAD_BATCH_SPLIT=1): one owner block per tile leaves CUs idle when the batch has fewer tiles than the
    // chip has CUs (the DiffNet stack: 20 layers x 8 tiles = 160 of 256).  With the workspace at hand the rows are split 3 ways
    // (fixed chunk interleave, partials summed in split order by a second launch: still bit-reproducible) and the problems go out
    // in groups that fill ONE round of 256 blocks: 2 x 240 blocks of a third of the rows instead of 1 x 160 blocks of all rows.
    // Measured (profiles/r06_wgrad_split.md): the k = 3 batch 931 -> 611 us (773 TFLOP/s), the k = 1 batch 470 -> 321 us -- and the
    // training step 13.60 -> 13.655 ms.  These launches run on the side stream next to the data-gradient chain; the CUs the split
    // takes are CUs the main stream's kernels were using, and the main stream is the step's critical path.  Hence off by default:
    // the right setting for a caller whose weight gradients are NOT overlapped (one stream), the wrong one for the training step.
    const char* bse = getenv("PTPP_WGRAD_BATCH_SPLIT");
    int nsplit = 1, group = MAXP;
    const size_t per = ((size_t)ks * Cout * Cin + Cout) * sizeof(float);
    if (bse && bse[0] == '1' && workspace && ((uintptr_t)workspace & 15) == 0 && Cout % 4 == 0 && (long long)tiles * nprob <= 192 &&
        (long long)Bf * ((Tf + 31) / 32) >= 96) {
      nsplit = 3;
      group = 256 / (tiles * nsplit);
      if (group < 1) group = 1;
      if (group > MAXP) group = MAXP;
      while (group > 1 && (size_t)group * nsplit * per > workspace_bytes) --group;
      if ((size_t)group * nsplit * per > workspace_bytes) { nsplit = 1; group = MAXP; }
    }
    for (int i0 = 0; i0 < nprob; i0 += group) {
      const int n = nprob - i0 < group ? nprob - i0 : group;
      const int rc = ptpp_wgrad_bf16_launch_batched(probs + i0, n, lengths, Bf, Tf, Cin, Cout, ks, ldx, lddy, in_mask, nsplit,
                                                    reinterpret_cast<float*>(workspace), reinterpret_cast<hipStream_t>(stream));
      if (rc != PTPP_OK) return rc;
    }
    return PTPP_OK;
  }
  for (int i = 0; i < nprob; ++i) {
    const int rc = ptpp_conv1d_wgrad(probs[i].x, probs[i].dy, probs[i].dw, probs[i].dbias, lengths, B, T, Cin, Cout, ks, probs[i].dil,
                                     probs[i].pad, ldx, lddy, in_mask, dtype, workspace, workspace_bytes, stream);
    if (rc != PTPP_OK) return rc;
  }
  return PTPP_OK;
}


int ptpp_wgrad_bf16_launch_grouped(const ptpp_wgrad_gproblem* probs, int nprob, hipStream_t st, hipStream_t st2);

extern "C" int ptpp_conv1d_wgrad_grouped(const ptpp_wgrad_gproblem* probs, int nprob, int dtype, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  return ptpp_conv1d_wgrad_grouped2(probs, nprob, dtype, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int ptpp_conv1d_wgrad_grouped2(const ptpp_wgrad_gproblem* probs, int nprob, int dtype, void* workspace, size_t workspace_bytes,
                                          void* stream, void* stream2) {
  PTPP_CHECK_ARG(probs && nprob > 0, "conv1d_wgrad_grouped: no problems");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "conv1d_wgrad_grouped: bad dtype %d", dtype);
  static const char* off = getenv("PTPP_WGRAD_NO_GROUP");
  bool fast = dtype == PTPP_BF16 && !(off && off[0] == '1');
  long long blocks = 0;
  for (int i = 0; i < nprob; ++i) {
    const ptpp_wgrad_gproblem& q = probs[i];
    PTPP_CHECK_ARG(q.x && q.dy && q.dw && q.B > 0 && q.T > 0 && q.Cin > 0 && q.Cout > 0 && q.ks > 0 && q.dil > 0,
                   "conv1d_wgrad_grouped: bad problem %d", i);
    // the one-owner blocks walk ALL rows of their problem: only for short problems (phone level); the tiles are 128 x 128
    fast = fast && q.ldx % 8 == 0 && q.lddy % 8 == 0 && q.Cin % 8 == 0 && q.Cout % 8 == 0 && ((uintptr_t)q.x % 16) == 0 &&
           ((uintptr_t)q.dy % 16) == 0 && q.Cin > 64 && q.Cout > 64 && (long long)q.B * ((q.T + 31) / 32) <= 256;
    blocks += (long long)((q.Cout + 127) / 128) * ((q.Cin + 127) / 128) * q.ks;
  }
  if (fast && blocks >= 64) {
    constexpr int GMAX = 16;
    for (int i0 = 0; i0 < nprob; i0 += GMAX) {
      const int rc = ptpp_wgrad_bf16_launch_grouped(probs + i0, nprob - i0 < GMAX ? nprob - i0 : GMAX, reinterpret_cast<hipStream_t>(stream),
                                                    reinterpret_cast<hipStream_t>(stream2));
      if (rc != PTPP_OK) return rc;
    }
    return PTPP_OK;
  }
  for (int i = 0; i < nprob; ++i) {
    const ptpp_wgrad_gproblem& q = probs[i];
    const int rc = ptpp_conv1d_wgrad(q.x, q.dy, q.dw, q.dbias, q.lengths, q.B, q.T, q.Cin, q.Cout, q.ks, q.dil, q.pad, q.ldx, q.lddy,
                                     q.lengths != nullptr, dtype, workspace, workspace_bytes, stream);
    if (rc != PTPP_OK) return rc;
  }
  return PTPP_OK;
}
