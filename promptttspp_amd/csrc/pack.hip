// Weight packing and (B,C,T) <-> (B,T,C) layout bridges.
#include "ptpp_common.h"

namespace {

// mode 0: wp[co][j][c]  = w[co][c][j]          (c <  Cin, else 0), rows = Cout, inner = CinP
// mode 1: wp[ci][j][c]  = w[c][ci][ks-1-j]     (c < Cout, else 0), rows = Cin,  inner = CoutP
// mode 2: mode 0 with the rows of a (gate | filter) weight (Cout = 2C) in the interleaved order of the fused DiffNet gate
//         epilogue (PTPP_ACT_GATE): packed row 8g + e (e < 4) = gate row 4g + e, packed row 8g + 4 + e = filter row C + 4g + e
__device__ __forceinline__ int gate_row_src(int r, int cout) {  // source row of packed row r
  const int g = r >> 3, e = r & 7;
  return e < 4 ? 4 * g + e : (cout >> 1) + 4 * g + (e - 4);
}
__device__ __forceinline__ int gate_row_dst(int co, int cout) {  // packed row of source row co
  const int C = cout >> 1;
  return co < C ? 8 * (co >> 2) + (co & 3) : 8 * ((co - C) >> 2) + 4 + ((co - C) & 3);
}
// modes 3 / 4: the operand of mode 0 / 1 as the OPERAND STREAM of the row-tile conv kernel (conv1d_rt.hip; 256 operand rows,
// inner % 64 == 0): stage s = ((k / 64) * ks + tap) * 2 + (k / 32) % 2 holds [256 rows][32 k] as the LDS image the MFMA
// fragments are read from -- LDS row q = wperm(n) (conv1d_common.h), 16-byte chunk (k / 8) % 4 at position chunk ^ swz4(q) --
// so that a block's weight traffic is ONE contiguous read in consumption order.  Same elements as modes 0 / 1, another order.
__device__ __forceinline__ int64_t stream_index(int n, int tap, int k, int ks) {
  const int ci = k >> 6, kh = (k >> 5) & 1, c = (k >> 3) & 3, e = k & 7;
  const int u = n & 63;
  const int q = (n - u) + (2 * (u >> 5) + ((u >> 2) & 1)) * 16 + 4 * ((u >> 3) & 3) + (u & 3);
  const int cp = c ^ ((-(q >> 2)) & 3);
  return ((int64_t)(ci * ks + tap) * 2 + kh) * 8192 + (q * 4 + cp) * 8 + e;
}

// operand rows beyond 256 (round 6: the 1024-channel side of the Conformer feed-forward convs): one stream per group of 256 rows,
// groups back to back; `inner` = the K length of a tap (all of it: the stream of a group has inner / 64 chunks)
__device__ __forceinline__ int64_t stream_index_g(int n, int tap, int k, int ks, int inner) {
  return (int64_t)(n >> 8) * ((int64_t)(inner >> 6) * ks * 2 * 8192) + stream_index(n & 255, tap, k, ks);
}

template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, T* __restrict__ wp, int cout, int cin, int ks,
                                 int mode, int rows, int inner, int innerp) {
  const int64_t n = (int64_t)rows * ks * innerp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % innerp);
    const int j = (int)((i / innerp) % ks);
    const int r = (int)(i / ((int64_t)innerp * ks));
    float v = 0.f;
    if (c < inner) {
      if (mode == 0 || mode == 3)
        v = w[((int64_t)r * cin + c) * ks + j];
      else if (mode == 2)
        v = w[((int64_t)gate_row_src(r, cout) * cin + c) * ks + j];
      else
        v = w[((int64_t)c * cin + r) * ks + (ks - 1 - j)];
    }
    Elem<T>::st(wp + (mode >= 3 ? stream_index_g(r, j, c, ks, innerp) : i), v);
  }
}

// Batched re-pack after an optimiser step: ONE launch for every cached operand (a training step uses
// ~350 packed operands; one launch each was ~4 ms of the step).  Table row (int64 x 10):
//   src f32 (cout, cin, ks) | dst base | cout | cin | ks | mode | dtype | innerp of dst | row/col offset | first block
// A block serves one 32 x 32 (co x ci) tile of one source, all taps.  Each source element goes to
//   mode 0: dst[((off + co) * ks + j) * innerp + ci]          (rows of several sources stack: fused QKV ...)
//   mode 1: dst[(ci * ks + ks-1-j) * innerp + off + co]       (their columns concatenate)
// The padding columns of dst were zeroed when the buffer was created.
// A block owns one 32 (co) x 32 (ci) tile of a source, all taps: reads run along ci, mode-0 writes along
// ci, mode-1 writes along co after a transpose through LDS (element-wise scattered 2-byte writes made
// the first version of this kernel 1 ms per step).
// LDS of a block: 32 rows of up to PB_PITCH_BYTES -- in the operand's own element type, so a bf16 operand needs half the
// space (eight resident blocks per CU instead of four: the launch is latency-bound, bytes in flight are what it lacks).
constexpr int PB_PITCH_BYTES = 580;  // 290 bf16 = 32 ci x 9 taps + 2
constexpr int PB_MAXV = 9;           // float4 loads per thread and pass (32 rows x 288 floats / 256 threads / 4)

template <typename T>
__device__ __forceinline__ void pack_batched_tile(const int64_t* e, int lb, unsigned char* lds) {
  const float* src = reinterpret_cast<const float*>(e[0]);
  T* dst = reinterpret_cast<T*>(e[1]);
  const int cout = (int)e[2], cin = (int)e[3], ks = (int)e[4], mode = (int)e[5];
  const int64_t innerp = e[7], off = e[8];
  const int nci = (cin + 31) / 32;
  const int co0 = (lb / nci) * 32, ci0 = (lb % nci) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  T* buf = reinterpret_cast<T*>(lds);
  // ci columns per pass: a source row segment [ci, ci + cw) x all taps is CONTIGUOUS (cw * ks floats) -- copied once, fully
  // coalesced, every tap then served from LDS (k = 9: one pass of 32 columns; k = 17: two of 16; f32 operands: half of that)
  constexpr int P = PB_PITCH_BYTES / (int)sizeof(T) - 2;
  constexpr int PAD = sizeof(T) == 2 ? 2 : 1;  // odd pitch in dwords: the transposed reads below hit distinct banks
  if (ks > P) {  // (no such conv exists; kept correct: one tap at a time through a 32 x 33 tile)
    for (int j = 0; j < ks; ++j) {
      for (int r = ty; r < 32; r += 8)
        Elem<T>::st(buf + r * 33 + tx, (co0 + r < cout && ci0 + tx < cin) ? src[((int64_t)(co0 + r) * cin + ci0 + tx) * ks + j] : 0.f);
      __syncthreads();
      for (int r = ty; r < 32; r += 8) {
        if (mode != 1 && mode != 4) {
          const int co = co0 + r, ci = ci0 + tx;
          if (co < cout && ci < cin)
            dst[mode == 3 ? stream_index_g(off + co, j, ci, ks, innerp) : ((off + (mode == 2 ? gate_row_dst(co, cout) : co)) * ks + j) * innerp + ci] = buf[r * 33 + tx];
        } else {
          const int ci = ci0 + r, co = co0 + tx;
          if (co < cout && ci < cin)
            dst[mode == 4 ? stream_index_g(ci, ks - 1 - j, off + co, ks, innerp) : ((int64_t)ci * ks + (ks - 1 - j)) * innerp + off + co] = buf[tx * 33 + r];
        }
      }
      __syncthreads();
    }
    return;
  }
  int cw = P / ks;
  cw = cw >= 32 ? 32 : (cw >= 4 ? (cw & ~3) : cw);
  const bool src_al = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && (((int64_t)cin * ks) & 3) == 0;
  for (int cs = 0; cs < 32 && ci0 + cs < cin; cs += cw) {
    const int cv = min(cw, cin - ci0 - cs);  // valid columns of this pass
    const int seg = cv * ks;                 // floats per row
    const int pitch = cw * ks + PAD;
    if (cs) __syncthreads();
    if (src_al && (seg & 3) == 0 && (((ci0 + cs) * ks) & 3) == 0) {
      const int nv = seg >> 2, items = 32 * nv;  // float4 per row / per pass: all requested before the first is used
      float4 v[PB_MAXV];
#pragma unroll
      for (int i = 0; i < PB_MAXV; ++i) {
        const int idx = threadIdx.x + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < items) {
          const int r = idx / nv, q = idx - r * nv;
          if (co0 + r < cout) v[i] = *reinterpret_cast<const float4*>(src + ((int64_t)(co0 + r) * cin + ci0 + cs) * ks + 4 * q);
        }
      }
#pragma unroll
      for (int i = 0; i < PB_MAXV; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < items) {
          const int r = idx / nv, q = idx - r * nv;
          T* d = buf + r * pitch + 4 * q;
          Elem<T>::st(d, v[i].x); Elem<T>::st(d + 1, v[i].y); Elem<T>::st(d + 2, v[i].z); Elem<T>::st(d + 3, v[i].w);
        }
      }
    } else {
      for (int idx = threadIdx.x; idx < 32 * seg; idx += 256) {
        const int r = idx / seg, k = idx - r * seg;
        Elem<T>::st(buf + r * pitch + k, co0 + r < cout ? src[((int64_t)(co0 + r) * cin + ci0 + cs) * ks + k] : 0.f);
      }
    }
    __syncthreads();
    if (mode != 1 && mode != 4) {  // writes run along ci
      const int co_w = co0 + ty, ci = ci0 + cs + tx;
      if (tx < cv) {
        for (int j = 0; j < ks; ++j) {
#pragma unroll
          for (int r = 0; r < 32; r += 8) {
            const int co = co_w + r;
            if (co < cout) {
              const int64_t d = mode == 3 ? stream_index_g(off + co, j, ci, ks, innerp)
                                          : ((off + (mode == 2 ? gate_row_dst(co, cout) : co)) * ks + j) * innerp + ci;
              dst[d] = buf[(ty + r) * pitch + tx * ks + j];
            }
          }
        }
      }
    } else {  // the tile transposed: r indexes ci, writes run along co
      const int co = co0 + tx;
      if (co < cout) {
        for (int j = 0; j < ks; ++j) {
          for (int r = ty; r < cv; r += 8) {
            const int ci = ci0 + cs + r;
            const int64_t d = mode == 4 ? stream_index_g(ci, ks - 1 - j, off + co, ks, innerp) : ((int64_t)ci * ks + (ks - 1 - j)) * innerp + off + co;
            dst[d] = buf[tx * pitch + r * ks + j];
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void pack_batched_kernel(const int64_t* __restrict__ tab, int n,
                                                           const int* __restrict__ block_map) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32 * PB_PITCH_BYTES];
  int lo = 0, hi = n - 1;
  if (block_map) lo = hi = block_map[blockIdx.x];
  while (lo < hi) {  // last row whose first block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid * 10 + 9] <= (int64_t)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const int64_t* e = tab + lo * 10;
  const int lb = (int)((int64_t)blockIdx.x - e[9]);
  if ((int)e[6] == PTPP_F32) pack_batched_tile<float>(e, lb, lds);
  else pack_batched_tile<bf16_raw>(e, lb, lds);
}

// 32x32 LDS-tiled transpose of the two inner dims with dtype conversion.
template <typename TI, typename TO>
__global__ void transpose_kernel(const TI* __restrict__ x, TO* __restrict__ y, int R, int S) {
  // x: (B, R, S) -> y: (B, S, R)
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const TI* xb = x + (int64_t)b * R * S;
  TO* yb = y + (int64_t)b * R * S;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, s = s0 + threadIdx.x;
    if (r < R && s < S) tile[i][threadIdx.x] = Elem<TI>::ld(xb + (int64_t)r * S + s);
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int s = s0 + i, r = r0 + threadIdx.x;
    if (r < R && s < S) Elem<TO>::st(yb + (int64_t)s * R + r, tile[threadIdx.x][i]);
  }
}

// nn.Conv2d weight (Cout, Cin, 3, 3) -> the two operands of the im2col GEMM, K index = (kh 3 + kw) cinq + ci (channels
// zero-padded to cinq):  wf[co][k] (k < K, else 0; Kp columns) forward, wb[k][co] (co < Cout, else 0; coutp columns) data
// gradient -- byte for byte ptpp_pack_conv_weight modes 0 / 1 of the (Cout, K) matrix the per-launch path builds with torch ops.
template <typename T>
__global__ void pack_conv2d3x3_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int cout, int cin, int cinq,
                                      int Kp, int coutp) {
  const int K = 9 * cinq;
  const int64_t nf = (int64_t)cout * Kp, nb = wb ? (int64_t)K * coutp : 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nf + nb; i += (int64_t)gridDim.x * blockDim.x) {
    int co, k;
    if (i < nf) { co = (int)(i / Kp); k = (int)(i % Kp); }
    else { k = (int)((i - nf) / coutp); co = (int)((i - nf) % coutp); }
    float v = 0.f;
    if (co < cout && k < K) {
      const int tap = k / cinq, ci = k % cinq;
      if (ci < cin) v = w[((int64_t)co * cin + ci) * 9 + tap];
    }
    if (i < nf) Elem<T>::st(wf + i, v);
    else Elem<T>::st(wb + (i - nf), v);
  }
}

}  // namespace

extern "C" int ptpp_pack_conv2d_3x3(const float* w, void* wp_fwd, void* wp_bwd, int cout, int cin, int cinq, int dtype, void* stream) {
  PTPP_CHECK_ARG(w && wp_fwd && cout > 0 && cin > 0 && cinq >= cin, "pack_conv2d_3x3: bad args");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16, "pack_conv2d_3x3: bad dtype");
  const int Kp = ptpp_conv_cin_padded(9 * cinq, dtype), coutp = ptpp_conv_cin_padded(cout, dtype);
  const int64_t n = (int64_t)cout * Kp + (wp_bwd ? (int64_t)9 * cinq * coutp : 0);
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(pack_conv2d3x3_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)wp_fwd, (float*)wp_bwd, cout, cin, cinq, Kp, coutp);
  else
    hipLaunchKernelGGL(pack_conv2d3x3_kernel<bf16_raw>, dim3(grid), dim3(256), 0, st, w, (bf16_raw*)wp_fwd, (bf16_raw*)wp_bwd, cout, cin, cinq,
                       Kp, coutp);
  PTPP_CHECK_LAUNCH("pack_conv2d_3x3");
  return PTPP_OK;
}

extern "C" int ptpp_pack_conv_weight(const float* w, void* wp, int cout, int cin, int ks, int mode, int dtype,
                                     void* stream) {
  PTPP_CHECK_ARG(w && wp, "pack_conv_weight: null pointer");
  PTPP_CHECK_ARG(cout > 0 && cin > 0 && ks > 0 && (mode == 0 || mode == 1 || (mode == 2 && cout % 8 == 0) || mode == 3 || mode == 4),
                 "pack_conv_weight: bad args");
  PTPP_CHECK_ARG(dtype == PTPP_F32 || dtype == PTPP_BF16 || (dtype == PTPP_F16 && mode <= 2), "pack_conv_weight: bad dtype (f16: modes 0 / 1 / 2)");
  const bool tr = mode == 1 || mode == 4;
  const int rows = !tr ? cout : cin;
  const int inner = !tr ? cin : cout;
  PTPP_CHECK_ARG(mode < 3 || (dtype == PTPP_BF16 && rows % 256 == 0 && inner % 64 == 0),
                 "pack_conv_weight: the operand stream (modes 3 / 4) needs bf16, operand rows %% 256 == 0 and inner %% 64 == 0 (rows %d, inner %d)",
                 rows, inner);
  const int innerp = ptpp_conv_cin_padded(inner, dtype);
  const int64_t n = (int64_t)rows * ks * innerp;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)wp, cout, cin, ks, mode, rows,
                       inner, innerp);
  else if (dtype == PTPP_F16)
    hipLaunchKernelGGL(pack_conv_kernel<f16_raw>, dim3(grid), dim3(256), 0, st, w, (f16_raw*)wp, cout, cin, ks, mode,
                       rows, inner, innerp);
  else
    hipLaunchKernelGGL(pack_conv_kernel<bf16_raw>, dim3(grid), dim3(256), 0, st, w, (bf16_raw*)wp, cout, cin, ks, mode,
                       rows, inner, innerp);
  PTPP_CHECK_LAUNCH("pack_conv_weight");
  return PTPP_OK;
}

extern "C" int ptpp_pack_conv_weights_batched(const int64_t* table, int n_entries, const int32_t* block_map,
                                              int total_blocks, void* stream) {
  PTPP_CHECK_ARG(table && n_entries > 0 && total_blocks > 0, "pack_conv_weights_batched: bad args");
  hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table,
                     n_entries, block_map);
  PTPP_CHECK_LAUNCH("pack_conv_weights_batched");
  return PTPP_OK;
}

extern "C" int ptpp_bct_to_btc(const float* x, void* y, int B, int C, int T, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0, "bct_to_btc: bad args");
  dim3 grid((T + 31) / 32, (C + 31) / 32, B), blk(32, 8);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, blk, 0, st, x, (float*)y, C, T);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL((transpose_kernel<float, bf16_raw>), grid, blk, 0, st, x, (bf16_raw*)y, C, T);
  else if (dtype == PTPP_F16)
    hipLaunchKernelGGL((transpose_kernel<float, f16_raw>), grid, blk, 0, st, x, (f16_raw*)y, C, T);
  else
    PTPP_CHECK_ARG(false, "bct_to_btc: bad dtype");
  PTPP_CHECK_LAUNCH("bct_to_btc");
  return PTPP_OK;
}

extern "C" int ptpp_btc_to_bct(const void* x, float* y, int B, int T, int C, int dtype, void* stream) {
  PTPP_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0, "btc_to_bct: bad args");
  dim3 grid((C + 31) / 32, (T + 31) / 32, B), blk(32, 8);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PTPP_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, blk, 0, st, (const float*)x, y, T, C);
  else if (dtype == PTPP_BF16)
    hipLaunchKernelGGL((transpose_kernel<bf16_raw, float>), grid, blk, 0, st, (const bf16_raw*)x, y, T, C);
  else if (dtype == PTPP_F16)
    hipLaunchKernelGGL((transpose_kernel<f16_raw, float>), grid, blk, 0, st, (const f16_raw*)x, y, T, C);
  else
    PTPP_CHECK_ARG(false, "btc_to_bct: bad dtype");
  PTPP_CHECK_LAUNCH("btc_to_bct");
  return PTPP_OK;
}
