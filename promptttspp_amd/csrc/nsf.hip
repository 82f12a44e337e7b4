// Harmonic-plus-noise source of the F0-aware vocoder in ONE launch (reference promptttspp/vocoders/nsf.py:31-206: SineGen._f02sine,
// SineGen.forward, SourceModuleHnNSF.forward).  Per utterance b and harmonic h (DIM = harmonic_num + 1 of them):
//
//   rad[t]     = frac(f0[t] * (h + 1) / sr)              (+ rand_ini[b, h] at t = 0)
//   wrapped[t] = frac(cumsum(rad)[t])
//   shift[t]   = wrapped[t] < wrapped[t - 1] ? -1 : 0    (t >= 1: drops the integer part of the phase)
//   sine[t]    = sin(2 pi cumsum(rad + shift)[t]) * amp
//   sw[t]      = sine[t] * uv[t] + (uv[t] * noise_std + (1 - uv[t]) * amp / 3) * noise[b, t, h],   uv = f0 > threshold
//   out[t]     = tanh(sum_h w[h] * sw[t, h] + bias)
//
// The tensor-op chain was ~20 elementwise passes over (B, L, DIM) f32 plus two torch.cumsum scans (2 x 0.59 ms at 32 x 141 600 x 9,
// profiles/r06_app_path.md).  Here a workgroup owns an utterance: every thread owns a contiguous piece of the time axis for ALL
// harmonics; three passes over its piece (local sums of rad -> offsets; local sums of rad + shift -> offsets; outputs), the two
// block-wide exclusive scans through LDS.  The three passes recompute rad / wrapped / shift with the same operations in the same
// order, so the shift decisions of pass 3 are those pass 2 summed.  A wrap detected one sample early or late (the first running
// sum reaches ~7e4, where an ulp is 0.008 of a period) moves the phase by exactly one period: sin() does not see it -- the output
// does not depend on the rounding of the first scan, and the second scan stays O(1).  Summation order: per-thread pieces, then a
// fixed tree over the threads (bit-reproducible; not torch.cumsum's order -- tests bound the difference).
#include "ptpp_common.h"
#include "../../include/ptpp.h"

namespace {

constexpr int NSF_THREADS = 1024;

template <int DIM>
__device__ __forceinline__ void nsf_block_exscan(float (&v)[DIM], float* lds /* [DIM][NSF_THREADS / 64] */) {
  // exclusive prefix over the block's threads, per harmonic: in-wave inclusive scan by shuffles, wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NWV = NSF_THREADS / 64;
  float inc[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) {
    float x = v[h];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    inc[h] = x;
    if (lane == 63) lds[h * NWV + wave] = x;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < DIM; ++h) {
    float base = 0.f;
    for (int w = 0; w < wave; ++w) base += lds[h * NWV + w];  // (fixed order; <= 15 terms)
    v[h] = base + inc[h] - v[h];
  }
  __syncthreads();
}

template <int DIM>
__global__ __launch_bounds__(NSF_THREADS) void nsf_source_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                                const float* __restrict__ noise, const float* __restrict__ w,
                                                                float bias, float* __restrict__ out, int L, float inv_sr, float amp,
                                                                float noise_std, float thr) {
#pragma clang fp contract(off)
  __shared__ float lds[DIM * (NSF_THREADS / 64)];
  const int b = blockIdx.x;
  const int ch = (L + NSF_THREADS - 1) / NSF_THREADS;
  const int t0 = min((int)threadIdx.x * ch, L), t1 = min(t0 + ch, L);
  const float* fb = f0 + (int64_t)b * L;
  float ini[DIM], wv[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) {
    ini[h] = rand_ini[b * DIM + h];
    wv[h] = w[h];
  }
  auto rad = [&](float f, int t, int h) __attribute__((always_inline)) {
    const float v = f * (float)(h + 1) * inv_sr;
    float r = v - floorf(v);
    if (t == 0) r += ini[h];
    return r;
  };
  // pass 1: local sums of rad
  float s[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) s[h] = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float f = fb[t];
#pragma unroll
    for (int h = 0; h < DIM; ++h) s[h] += rad(f, t, h);
  }
  nsf_block_exscan<DIM>(s, lds);
  float off1[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) off1[h] = s[h];
  // pass 2: local sums of rad + shift
#pragma unroll
  for (int h = 0; h < DIM; ++h) s[h] = 0.f;
  {
    float c[DIM], pw[DIM];
#pragma unroll
    for (int h = 0; h < DIM; ++h) {
      c[h] = off1[h];
      pw[h] = c[h] - floorf(c[h]);
    }
    for (int t = t0; t < t1; ++t) {
      const float f = fb[t];
#pragma unroll
      for (int h = 0; h < DIM; ++h) {
        const float r = rad(f, t, h);
        c[h] += r;
        const float wr = c[h] - floorf(c[h]);
        const float sh = (t > 0 && wr - pw[h] < 0.f) ? -1.f : 0.f;
        s[h] += r + sh;
        pw[h] = wr;
      }
    }
  }
  nsf_block_exscan<DIM>(s, lds);
  // pass 3: outputs
  {
    float c[DIM], pw[DIM], c2[DIM];
#pragma unroll
    for (int h = 0; h < DIM; ++h) {
      c[h] = off1[h];
      pw[h] = c[h] - floorf(c[h]);
      c2[h] = s[h];
    }
    const float* nb = noise + (int64_t)b * L * DIM;
    float* ob = out + (int64_t)b * L;
    for (int t = t0; t < t1; ++t) {
      const float f = fb[t];
      const float uv = f > thr ? 1.f : 0.f;
      const float na = uv * noise_std + (1.f - uv) * amp / 3.f;
      float acc = 0.f;
#pragma unroll
      for (int h = 0; h < DIM; ++h) {
        const float r = rad(f, t, h);
        c[h] += r;
        const float wr = c[h] - floorf(c[h]);
        const float sh = (t > 0 && wr - pw[h] < 0.f) ? -1.f : 0.f;
        pw[h] = wr;
        c2[h] += r + sh;
        const float sine = sinf(c2[h] * 2.f * 3.14159265358979323846f) * amp;
        const float sw = sine * uv + na * nb[(int64_t)t * DIM + h];
        acc += sw * wv[h];
      }
      ob[t] = tanhf(acc + bias);
    }
  }
}

}  // namespace

extern "C" int ptpp_nsf_source_supported(int dim) { return dim == 9 || dim == 1 ? 1 : 0; }

extern "C" int ptpp_nsf_source(const float* f0, const float* rand_ini, const float* noise, const float* w, float bias, float* out, int B,
                               int L, int dim, float sampling_rate, float sine_amp, float noise_std, float voiced_threshold, void* stream) {
  PTPP_CHECK_ARG(f0 && rand_ini && noise && w && out && B > 0 && L > 0, "nsf_source: bad args");
  PTPP_CHECK_ARG(ptpp_nsf_source_supported(dim), "nsf_source: %d harmonics + 1 not instantiated (9 or 1)", dim);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float inv = 1.f / sampling_rate;
  if (dim == 9)
    hipLaunchKernelGGL(nsf_source_kernel<9>, dim3((unsigned)B), dim3(NSF_THREADS), 0, st, f0, rand_ini, noise, w, bias, out, L, inv, sine_amp,
                       noise_std, voiced_threshold);
  else
    hipLaunchKernelGGL(nsf_source_kernel<1>, dim3((unsigned)B), dim3(NSF_THREADS), 0, st, f0, rand_ini, noise, w, bias, out, L, inv, sine_amp,
                       noise_std, voiced_threshold);
  PTPP_CHECK_LAUNCH("nsf_source");
  return PTPP_OK;
}
