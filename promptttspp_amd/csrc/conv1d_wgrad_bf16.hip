// bf16-rate weight gradient of the channels-last Conv1d:
//   dw[co][ci][j] += sum_{b,t} dy[b,t,co] * x[b, t + j*dil - pad, ci]
// The reduction runs over ROWS, but v_mfma_f32_16x16x32_bf16 wants each lane to hold 8
// consecutive K (= row) values of one channel, while channels-last memory is channel
// contiguous.  gfx950's LDS transpose read (ds_read_b64_tr_b16) does that transpose for
// free: tiles are staged row-major [row][channel] with plain 16-byte copies, and each
// operand fragment is two tr reads (4 rows x 16 channels each).
//
// A block owns a 64(co) x 64(ci) tile for a GROUP of up to 5 taps: the dy chunk (32 rows)
// is staged once and reused by every tap, the x window (32 + (taps-1)*dil rows) serves
// all taps at shifted row offsets.  Rows are split across blocks (split-K) and combined
// with f32 atomics.
#include "ptpp_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(8))) short v8s;

constexpr int KR = 32;   // rows per K chunk (= one MFMA K)
constexpr int LS = 72;   // LDS row stride in bf16 elements (64 + 8: 16-byte aligned rows, de-phased banks)
constexpr int NT = 5;    // taps per block (accumulators: NT x 16 VGPRs)

__device__ __forceinline__ v4s tr_read(const bf16_raw* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
}

// 8 consecutive rows (k = 8g .. 8g+7) of column (c0 + i) for lane l = 16 g + i
__device__ __forceinline__ bf16x8_t frag(const bf16_raw* tile, int row0, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const bf16_raw* p = tile + (row0 + 8 * g + (i >> 2)) * LS + c0 + 4 * (i & 3);
  const v4s lo = tr_read(p), hi = tr_read(p + 4 * LS);
  v8s v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256) void conv1d_wgrad_bf16_kernel(const bf16_raw* __restrict__ x,
                                                                const bf16_raw* __restrict__ dy, float* __restrict__ dw,
                                                                float* __restrict__ dbias, const int* __restrict__ lengths,
                                                                int B, int T_, int Cin, int Cout, int ks, int dil, int pad,
                                                                int ldx, int lddy, int in_mask, int nCO, int nCI, int nTG,
                                                                int nsplit, int tchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_raw* dYs = reinterpret_cast<bf16_raw*>(smem);  // [KR][LS]
  bf16_raw* Xs = dYs + KR * LS;                       // [XR][LS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  int bid = blockIdx.x;
  const int cot = bid % nCO; bid /= nCO;
  const int cit = bid % nCI; bid /= nCI;
  const int tg = bid % nTG;  bid /= nTG;
  const int split = bid;
  const int co0 = cot * 64, ci0 = cit * 64;
  const int j0 = tg * NT, nt = min(NT, ks - j0);
  const int XR = KR + (nt - 1) * dil;

  f32x4 acc[NT][2][2];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[j][a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = dbias && cit == 0 && tg == 0;

  const int total = B * tchunks;
  for (int ch = split; ch < total; ch += nsplit) {
    const int b = ch / tchunks, tb = (ch % tchunks) * KR;
    const int Tin = (in_mask && lengths) ? min(lengths[b], T_) : T_;
    // stage dy chunk: KR rows x 8 chunks of 8 channels
    {
      const int row = tid >> 3, c8 = (tid & 7) * 8;
      const int t = tb + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (t < T_ && co0 + c8 < Cout) v = *reinterpret_cast<const uint4*>(dy + ((int64_t)b * T_ + t) * lddy + co0 + c8);
      *reinterpret_cast<uint4*>(dYs + row * LS + c8) = v;
    }
    // stage x window
    const int xbase = tb + j0 * dil - pad;
    for (int idx = tid; idx < XR * 8; idx += 256) {
      const int row = idx >> 3, c8 = (idx & 7) * 8;
      const int ts = xbase + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ts >= 0 && ts < Tin && ci0 + c8 < Cin) v = *reinterpret_cast<const uint4*>(x + ((int64_t)b * T_ + ts) * ldx + ci0 + c8);
      *reinterpret_cast<uint4*>(Xs + row * LS + c8) = v;
    }
    __syncthreads();
    bf16x8_t af[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = frag(dYs, 0, wr * 32 + a * 16, lane);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < nt) {
        bf16x8_t bf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[c] = frag(Xs, j * dil, wc * 32 + c * 16, lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[j][a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[c], acc[j][a][c], 0, 0, 0);
      }
    }
    if (do_bias && tid < 64) {
#pragma unroll 8
      for (int r = 0; r < KR; ++r) bsum += bf16_to_f32(dYs[r * LS + tid]);
    }
    __syncthreads();
  }

  // D[i = co (rows 4*(lane>>4) + r)][j = ci (col lane&15)]
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j < nt) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int ci = ci0 + wc * 32 + c * 16 + lr;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + wr * 32 + a * 16 + lg * 4 + r;
            if (co < Cout && ci < Cin) atomicAdd(dw + ((int64_t)co * Cin + ci) * ks + j0 + j, acc[j][a][c][r]);
          }
        }
    }
  }
  if (do_bias && tid < 64 && co0 + tid < Cout) atomicAdd(dbias + co0 + tid, bsum);
}

}  // namespace

// called by ptpp_conv1d_wgrad for bf16 tensors with 16-byte aligned rows
int ptpp_wgrad_bf16_launch(const void* x, const void* dy, float* dw, float* dbias, const int32_t* lengths, int B, int T,
                           int Cin, int Cout, int ks, int dil, int pad, int ldx, int lddy, int in_mask, hipStream_t st) {
  const int nCO = (Cout + 63) / 64, nCI = (Cin + 63) / 64, nTG = (ks + NT - 1) / NT;
  const int tchunks = (T + KR - 1) / KR;
  const int total = B * tchunks;
  const int tiles = nCO * nCI * nTG;
  int nsplit = (1024 + tiles - 1) / tiles;  // ~4 blocks per CU
  if (nsplit > total) nsplit = total;
  if (nsplit < 1) nsplit = 1;
  const int ntmax = ks < NT ? ks : NT;
  const size_t smem = (size_t)(KR + KR + (ntmax - 1) * dil) * LS * sizeof(bf16_raw);
  if (smem > 64 * 1024) return PTPP_ENOTSUP;
  hipLaunchKernelGGL(conv1d_wgrad_bf16_kernel, dim3((unsigned)((int64_t)tiles * nsplit)), dim3(256), smem, st,
                     (const bf16_raw*)x, (const bf16_raw*)dy, dw, dbias, lengths, B, T, Cin, Cout, ks, dil, pad, ldx, lddy,
                     in_mask, nCO, nCI, nTG, nsplit, tchunks);
  PTPP_CHECK_LAUNCH("conv1d_wgrad(bf16)");
  return PTPP_OK;
}
