// GRU cell gate algebra for the GST reference encoder (modules/reference_encoder.py:108-123:
// a packed single-layer nn.GRU over <= ~12 steps): one launch per step instead of ~12 elementwise
// launches forward and ~25 backward.  The two projections (W_ih x, W_hh h) are GEMMs on the conv kernel.
//   r = sigmoid(gi_r + gh_r)   z = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)
//   h' = (1 - z) n + z h        rows whose sequence has ended (step >= len[b]) keep h.
// Everything f32 (the recurrence is run in f32 like the reference).  The backward kernel recomputes
// r, z, n from the saved pre-activations.
#include "ptpp_common.h"

namespace {

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(const float* __restrict__ gi, int64_t ldgi,
                                                           const float* __restrict__ gh, const float* __restrict__ h,
                                                           const int* __restrict__ lens, int step, float* __restrict__ hout,
                                                           int B, int H) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  const float hp = h[i];
  if (lens && step >= lens[b]) {
    hout[i] = hp;
    return;
  }
  const float* gib = gi + (int64_t)b * ldgi;
  const float* ghb = gh + (int64_t)b * 3 * H;
  const float r = sigm(gib[j] + ghb[j]);
  const float z = sigm(gib[H + j] + ghb[H + j]);
  const float n = tanhf(gib[2 * H + j] + r * ghb[2 * H + j]);
  hout[i] = (1.f - z) * n + z * hp;
}

__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const float* __restrict__ gi, int64_t ldgi,
                                                           const float* __restrict__ gh, const float* __restrict__ h,
                                                           const int* __restrict__ lens, int step,
                                                           const float* __restrict__ dhout, float* __restrict__ dgi,
                                                           int64_t lddgi, float* __restrict__ dgh, float* __restrict__ dh,
                                                           int B, int H) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  float* dgib = dgi + (int64_t)b * lddgi;
  float* dghb = dgh + (int64_t)b * 3 * H;
  const float d = dhout[i];
  if (lens && step >= lens[b]) {
    dh[i] = d;
    dgib[j] = dgib[H + j] = dgib[2 * H + j] = 0.f;
    dghb[j] = dghb[H + j] = dghb[2 * H + j] = 0.f;
    return;
  }
  const float* gib = gi + (int64_t)b * ldgi;
  const float* ghb = gh + (int64_t)b * 3 * H;
  const float hp = h[i];
  const float ghn = ghb[2 * H + j];
  const float r = sigm(gib[j] + ghb[j]);
  const float z = sigm(gib[H + j] + ghb[H + j]);
  const float n = tanhf(gib[2 * H + j] + r * ghn);
  const float dn_pre = d * (1.f - z) * (1.f - n * n);
  const float dz_pre = d * (hp - n) * z * (1.f - z);
  const float dr_pre = dn_pre * ghn * r * (1.f - r);
  dh[i] = d * z;
  dgib[j] = dr_pre;
  dgib[H + j] = dz_pre;
  dgib[2 * H + j] = dn_pre;
  dghb[j] = dr_pre;
  dghb[H + j] = dz_pre;
  dghb[2 * H + j] = dn_pre * r;
}


// ---- the whole recurrence in ONE launch (H = 128 or 256) ------------------------------------------------------------
// The per-step path costs 2 (forward) / 3 (backward) launches per step of <= 10 us of work each on (B, H) = (~20, 256)
// operands; with ~24 steps that is ~120 launches per training step.  Here a block owns one sequence for all its steps:
// thread j keeps the first 128 entries of row j of W_hh (forward) / of its slice of column k (backward) in registers and
// streams the rest (H = 256: another 128 per step, coalesced, from L2 -- the forward reads them from the TRANSPOSED copy
// w_hh_t so that neighbouring threads read neighbouring addresses); h / dgh travel through LDS, and the pre-activations
// W_hh h + b_hh and the hidden states are written once for the backward.  Steps at or after len[b] are skipped (the
// packed-sequence semantics: h frozen, zero gradient).
template <int H> struct GruKreg { static constexpr int v = H > 128 ? 96 : 128; };  // (H = 256: 768 threads cap a lane at 168 registers)

template <int H>
__global__ __launch_bounds__(3 * H) void gru_seq_fwd_kernel(const float* __restrict__ gi_all, const float* __restrict__ w_hh,
                                                             const float* __restrict__ w_hh_t, int ldt,
                                                             const float* __restrict__ b_hh, const int* __restrict__ lens,
                                                             float* __restrict__ hs_all, float* __restrict__ gh_all,
                                                             float* __restrict__ hout, int B, int L) {
  __shared__ __attribute__((aligned(16))) float hs[H];
  __shared__ float ghs[3 * H];
  const int b = blockIdx.x, j = threadIdx.x;
  constexpr int GRU_KREG = GruKreg<H>::v;
  float w[GRU_KREG];
#pragma unroll
  for (int k = 0; k < GRU_KREG; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(w_hh + (int64_t)j * H + k);
    w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
  }
  const float bias = b_hh[j];
  const int n = lens ? min(max(lens[b], 0), L) : L;
  float hreg = 0.f;  // threads j < H: h[j]
  if (j < H) {
    hs[j] = 0.f;
    hs_all[(int64_t)b * H + j] = 0.f;
  }
  __syncthreads();
  for (int s = 0; s < n; ++s) {
    float a0 = bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < GRU_KREG; k += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(hs + k);
      a0 = fmaf(w[k], hv.x, a0); a1 = fmaf(w[k + 1], hv.y, a1); a2 = fmaf(w[k + 2], hv.z, a2); a3 = fmaf(w[k + 3], hv.w, a3);
    }
    if constexpr (H > GRU_KREG) {
#pragma unroll 8
      for (int k = GRU_KREG; k < H; k += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(hs + k);
        const float* wt = w_hh_t + (int64_t)k * ldt + j;
        a0 = fmaf(wt[0], hv.x, a0); a1 = fmaf(wt[ldt], hv.y, a1); a2 = fmaf(wt[2 * ldt], hv.z, a2); a3 = fmaf(wt[3 * ldt], hv.w, a3);
      }
    }
    const float gh = (a0 + a1) + (a2 + a3);
    ghs[j] = gh;
    gh_all[((int64_t)s * B + b) * 3 * H + j] = gh;
    __syncthreads();
    if (j < H) {
      const float* gib = gi_all + ((int64_t)b * L + s) * 3 * H;
      const float r = sigm(gib[j] + ghs[j]);
      const float z = sigm(gib[H + j] + ghs[H + j]);
      const float nn = tanhf(gib[2 * H + j] + r * ghs[2 * H + j]);
      hreg = (1.f - z) * nn + z * hreg;
      hs[j] = hreg;
      hs_all[((int64_t)(s + 1) * B + b) * H + j] = hreg;
    }
    __syncthreads();
  }
  if (j < H) hout[(int64_t)b * H + j] = hreg;
}

template <int H>
__global__ __launch_bounds__(3 * H) void gru_seq_bwd_kernel(const float* __restrict__ gi_all, const float* __restrict__ w_hh,
                                                             const int* __restrict__ lens, const float* __restrict__ hs_all,
                                                             const float* __restrict__ gh_all, const float* __restrict__ dhout,
                                                             float* __restrict__ dgi_all, float* __restrict__ dgh_all, int B, int L) {
  __shared__ __attribute__((aligned(16))) float dghs[3 * H];
  __shared__ float part[3][H];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int p = tid / H, k = tid % H;
  constexpr int GRU_KREG = GruKreg<H>::v;
  float w[GRU_KREG];  // W_hh[p H + i][k], i < GRU_KREG: this thread's slice of column k
#pragma unroll
  for (int i = 0; i < GRU_KREG; ++i) w[i] = w_hh[(int64_t)(p * H + i) * H + k];
  const int n = lens ? min(max(lens[b], 0), L) : L;
  float d = tid < H ? dhout[(int64_t)b * H + tid] : 0.f;  // threads < H: dL/dh_s[tid]
  for (int s = L - 1; s >= n; --s) {  // steps after the sequence's end: h passed through, no gradient to the gates
    dgi_all[((int64_t)b * L + s) * 3 * H + tid] = 0.f;
    dgh_all[((int64_t)s * B + b) * 3 * H + tid] = 0.f;
  }
  for (int s = n - 1; s >= 0; --s) {
    float direct = 0.f;
    if (tid < H) {
      const int j = tid;
      const float* gib = gi_all + ((int64_t)b * L + s) * 3 * H;
      const float* ghb = gh_all + ((int64_t)s * B + b) * 3 * H;
      const float hp = hs_all[((int64_t)s * B + b) * H + j];
      const float ghn = ghb[2 * H + j];
      const float r = sigm(gib[j] + ghb[j]);
      const float z = sigm(gib[H + j] + ghb[H + j]);
      const float nn = tanhf(gib[2 * H + j] + r * ghn);
      const float dn_pre = d * (1.f - z) * (1.f - nn * nn);
      const float dz_pre = d * (hp - nn) * z * (1.f - z);
      const float dr_pre = dn_pre * ghn * r * (1.f - r);
      direct = d * z;
      float* dgib = dgi_all + ((int64_t)b * L + s) * 3 * H;
      float* dghb = dgh_all + ((int64_t)s * B + b) * 3 * H;
      dgib[j] = dr_pre; dgib[H + j] = dz_pre; dgib[2 * H + j] = dn_pre;
      dghb[j] = dr_pre; dghb[H + j] = dz_pre; dghb[2 * H + j] = dn_pre * r;
      dghs[j] = dr_pre; dghs[H + j] = dz_pre; dghs[2 * H + j] = dn_pre * r;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < GRU_KREG; i += 4) {
      const float4 g = *reinterpret_cast<const float4*>(dghs + p * H + i);
      a0 = fmaf(w[i], g.x, a0); a1 = fmaf(w[i + 1], g.y, a1); a2 = fmaf(w[i + 2], g.z, a2); a3 = fmaf(w[i + 3], g.w, a3);
    }
    if constexpr (H > GRU_KREG) {
#pragma unroll 8
      for (int i = GRU_KREG; i < H; i += 4) {
        const float4 g = *reinterpret_cast<const float4*>(dghs + p * H + i);
        const float* wr = w_hh + (int64_t)(p * H + i) * H + k;
        a0 = fmaf(wr[0], g.x, a0); a1 = fmaf(wr[H], g.y, a1); a2 = fmaf(wr[2 * H], g.z, a2); a3 = fmaf(wr[3 * H], g.w, a3);
      }
    }
    part[p][k] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < H) d = direct + (part[0][tid] + part[1][tid] + part[2][tid]);
  }
}

}  // namespace

extern "C" int ptpp_gru_gate_fwd(const float* gi, int64_t ldgi, const float* gh, const float* h, const int32_t* lens,
                                 int step, float* hout, int B, int H, void* stream) {
  PTPP_CHECK_ARG(gi && gh && h && hout && B > 0 && H > 0 && ldgi >= 3 * (int64_t)H, "gru_gate_fwd: bad args");
  const int64_t n = (int64_t)B * H;
  hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     gi, ldgi, gh, h, lens, step, hout, B, H);
  PTPP_CHECK_LAUNCH("gru_gate_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_gru_gate_bwd(const float* gi, int64_t ldgi, const float* gh, const float* h, const int32_t* lens,
                                 int step, const float* dhout, float* dgi, int64_t lddgi, float* dgh, float* dh, int B,
                                 int H, void* stream) {
  PTPP_CHECK_ARG(gi && gh && h && dhout && dgi && dgh && dh && B > 0 && H > 0 && ldgi >= 3 * (int64_t)H &&
                     lddgi >= 3 * (int64_t)H,
                 "gru_gate_bwd: bad args");
  const int64_t n = (int64_t)B * H;
  hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     gi, ldgi, gh, h, lens, step, dhout, dgi, lddgi, dgh, dh, B, H);
  PTPP_CHECK_LAUNCH("gru_gate_bwd");
  return PTPP_OK;
}

extern "C" int ptpp_gru_seq_supported(int H) { return H == 128 || H == 256; }

extern "C" int ptpp_gru_seq_fwd(const float* gi_all, const float* w_hh, const float* w_hh_t, const float* b_hh, const int32_t* lens,
                                float* hs_all, float* gh_all, float* hout, int B, int L, int H, void* stream) {
  PTPP_CHECK_ARG(gi_all && w_hh && b_hh && hs_all && gh_all && hout && B > 0 && L > 0, "gru_seq_fwd: bad args");
  PTPP_CHECK_ARG(H == 128 || H == 256, "gru_seq_fwd: H = %d (the one-launch recurrence is built for H = 128 / 256; use the per-step entry points)", H);
  PTPP_CHECK_ARG(((uintptr_t)w_hh & 15) == 0, "gru_seq_fwd: w_hh must be 16-byte aligned");
  PTPP_CHECK_ARG(H == 128 || w_hh_t, "gru_seq_fwd: H = 256 needs the transposed copy w_hh_t");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ldt = ptpp_conv_cin_padded(3 * H, PTPP_F32);
  if (H == 128)
    hipLaunchKernelGGL(gru_seq_fwd_kernel<128>, dim3((unsigned)B), dim3(3 * 128), 0, st, gi_all, w_hh, w_hh_t, ldt, b_hh, lens, hs_all,
                       gh_all, hout, B, L);
  else
    hipLaunchKernelGGL(gru_seq_fwd_kernel<256>, dim3((unsigned)B), dim3(3 * 256), 0, st, gi_all, w_hh, w_hh_t, ldt, b_hh, lens, hs_all,
                       gh_all, hout, B, L);
  PTPP_CHECK_LAUNCH("gru_seq_fwd");
  return PTPP_OK;
}

extern "C" int ptpp_gru_seq_bwd(const float* gi_all, const float* w_hh, const int32_t* lens, const float* hs_all, const float* gh_all,
                                const float* dhout, float* dgi_all, float* dgh_all, int B, int L, int H, void* stream) {
  PTPP_CHECK_ARG(gi_all && w_hh && hs_all && gh_all && dhout && dgi_all && dgh_all && B > 0 && L > 0, "gru_seq_bwd: bad args");
  PTPP_CHECK_ARG(H == 128 || H == 256, "gru_seq_bwd: H = %d (built for H = 128 / 256)", H);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (H == 128)
    hipLaunchKernelGGL(gru_seq_bwd_kernel<128>, dim3((unsigned)B), dim3(3 * 128), 0, st, gi_all, w_hh, lens, hs_all, gh_all, dhout, dgi_all,
                       dgh_all, B, L);
  else
    hipLaunchKernelGGL(gru_seq_bwd_kernel<256>, dim3((unsigned)B), dim3(3 * 256), 0, st, gi_all, w_hh, lens, hs_all, gh_all, dhout, dgi_all,
                       dgh_all, B, L);
  PTPP_CHECK_LAUNCH("gru_seq_bwd");
  return PTPP_OK;
}
