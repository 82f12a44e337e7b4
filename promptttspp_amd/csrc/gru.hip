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
