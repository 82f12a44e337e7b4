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
// profiles/r06_app_path.md).  Here a workgroup owns an utterance and each of its 16 waves a contiguous range of the time axis for ALL
// harmonics; three passes over the range (sum of rad -> offsets; sum of rad + shift -> offsets; outputs), the offsets of the ranges
// through LDS.  (A first version gave every THREAD a contiguous piece: its (B, L, DIM) noise reads were 64 cache lines per wave
// instruction and the launch slower than the tensor ops.)  The passes recompute rad / wrapped / shift with the same operations in
// the same order, so the shift decisions of pass 3 are those pass 2 summed.  A wrap detected one sample early or late (the first running
// sum reaches ~7e4, where an ulp is 0.008 of a period) moves the phase by exactly one period: sin() does not see it -- the output
// does not depend on the rounding of the first scan, and the second scan stays O(1).  Summation order: a fixed shuffle tree inside
// a tile of 64 steps, tiles and ranges in sequence (bit-reproducible; not torch.cumsum's order -- tests bound the difference).
#include "ptpp_common.h"
#include "../../include/ptpp.h"

namespace {

constexpr int NSF_THREADS = 1024;
constexpr int NSF_WAVES = NSF_THREADS / 64;

// Inclusive prefix over the 64 lanes by DPP adds (GFX9 row shifts + row broadcasts; a source lane outside the row / a masked row
// contributes 0).  The shuffle (ds_bpermute) form of the same tree has ~10x the latency per step, and a tile needs 18 of these
// scans back to back: 1.31 ms per launch against 0.2 ms.
#define NSF_DPP(x, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rows, 0xf, true))
__device__ __forceinline__ float nsf_wave_incl(float x, int) {
  x += NSF_DPP(x, 0x111, 0xf);  // row_shr:1
  x += NSF_DPP(x, 0x112, 0xf);  // row_shr:2
  x += NSF_DPP(x, 0x114, 0xf);  // row_shr:4
  x += NSF_DPP(x, 0x118, 0xf);  // row_shr:8 -- every row of 16 lanes now holds its own prefix
  x += NSF_DPP(x, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  x += NSF_DPP(x, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ float nsf_lane63(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63)); }
__device__ __forceinline__ float nsf_prev_lane(float x) {  // lane l gets lane l - 1's value (lane 0: 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));  // wave_shr:1
}
__device__ __forceinline__ float nsf_wave_sum(float x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
  return x;
}

// A wave owns a contiguous range of the time axis and walks it in tiles of 64 steps, one step per lane (the (B, L, DIM) noise and
// the output are read / written coalesced): prefix sums inside a tile by DPP adds, the running totals of the range in wave-uniform
// registers.  An utterance is G workgroups x 16 waves = 16 G ranges; the offsets of the ranges travel through a small global table
// between THREE launches of this kernel (PASS 1: range sums of rad; 2: of rad + shift; 3: outputs) -- one workgroup per utterance
// used 32 of 256 CUs at the config-5 batch and took 0.88 ms.
template <int DIM, int PASS>
__global__ __launch_bounds__(NSF_THREADS) void nsf_source_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini,
                                                                const float* __restrict__ noise, const float* __restrict__ w,
                                                                float bias, float* __restrict__ out, float* __restrict__ tab, int L,
                                                                float inv_sr, float amp, float noise_std, float thr) {
#pragma clang fp contract(off)
  const int b = blockIdx.y, nr = gridDim.x * NSF_WAVES;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rid = blockIdx.x * NSF_WAVES + wave;  // range of this wave
  const int per = ((L + nr - 1) / nr + 63) & ~63;
  const int w0 = min(rid * per, L), w1 = min(w0 + per, L);
  const float* fb = f0 + (int64_t)b * L;
  float* t1 = tab + (int64_t)b * nr * DIM;                                  // [range][DIM] sums of rad
  float* t2 = tab + ((int64_t)gridDim.y * nr + (int64_t)b * nr) * DIM;      // ... of rad + shift
  float ini[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) ini[h] = rand_ini[b * DIM + h];
  auto rad = [&](float f, int t, int h) __attribute__((always_inline)) {
    const float v = f * (float)(h + 1) * inv_sr;
    float r = v - floorf(v);
    if (t == 0) r += ini[h];
    return r;
  };
  if constexpr (PASS == 1) {
    float s[DIM];
#pragma unroll
    for (int h = 0; h < DIM; ++h) s[h] = 0.f;
    for (int t = w0 + lane; t < w1; t += 64) {
      const float f = fb[t];
#pragma unroll
      for (int h = 0; h < DIM; ++h) s[h] += rad(f, t, h);
    }
#pragma unroll
    for (int h = 0; h < DIM; ++h) {
      const float tot = nsf_wave_sum(s[h]);
      if (lane == 0) t1[rid * DIM + h] = tot;
    }
    return;
  }
  // offsets of this range: the sums of the ranges before it, in range order (wave-uniform loads)
  float off1[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) off1[h] = 0.f;
  for (int k = 0; k < rid; ++k) {
#pragma unroll
    for (int h = 0; h < DIM; ++h) off1[h] += t1[k * DIM + h];
  }
  // one tile: rad, running first sum, wrapped, shift (identical arithmetic in pass 2 and pass 3)
  auto tile_step = [&](int t, bool valid, float f, float (&c)[DIM], float (&pw)[DIM], float (&r)[DIM], float (&sh)[DIM]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < DIM; ++h) {
      r[h] = valid ? rad(f, t, h) : 0.f;
      const float inc = nsf_wave_incl(r[h], lane);
      const float ct = c[h] + inc;
      const float wr = ct - floorf(ct);
      float prev = nsf_prev_lane(wr);
      if (lane == 0) prev = pw[h];
      sh[h] = (valid && t > 0 && wr - prev < 0.f) ? -1.f : 0.f;
      c[h] += nsf_lane63(inc);
      pw[h] = nsf_lane63(wr);  // (lanes past the end carry the last valid step's sum: r = 0 there)
    }
  };
  float c[DIM], pw[DIM], r[DIM], sh[DIM];
#pragma unroll
  for (int h = 0; h < DIM; ++h) {
    c[h] = off1[h];
    pw[h] = c[h] - floorf(c[h]);
  }
  if constexpr (PASS == 2) {
    float s2[DIM];
#pragma unroll
    for (int h = 0; h < DIM; ++h) s2[h] = 0.f;
    for (int tile = w0; tile < w1; tile += 64) {
      const int t = tile + lane;
      const bool valid = t < w1;
      const float f = valid ? fb[t] : 0.f;
      tile_step(t, valid, f, c, pw, r, sh);
#pragma unroll
      for (int h = 0; h < DIM; ++h) s2[h] += r[h] + sh[h];
    }
#pragma unroll
    for (int h = 0; h < DIM; ++h) {
      const float tot = nsf_wave_sum(s2[h]);
      if (lane == 0) t2[rid * DIM + h] = tot;
    }
    return;
  }
  if constexpr (PASS == 3) {
    float c2[DIM], wv[DIM];
#pragma unroll
    for (int h = 0; h < DIM; ++h) {
      c2[h] = 0.f;
      wv[h] = w[h];
    }
    for (int k = 0; k < rid; ++k) {
#pragma unroll
      for (int h = 0; h < DIM; ++h) c2[h] += t2[k * DIM + h];
    }
    const float* nb = noise + (int64_t)b * L * DIM;
    float* ob = out + (int64_t)b * L;
    for (int tile = w0; tile < w1; tile += 64) {
      const int t = tile + lane;
      const bool valid = t < w1;
      const float f = valid ? fb[t] : 0.f;
      tile_step(t, valid, f, c, pw, r, sh);
      const float uv = f > thr ? 1.f : 0.f;
      const float na = uv * noise_std + (1.f - uv) * amp / 3.f;
      float acc = 0.f;
#pragma unroll
      for (int h = 0; h < DIM; ++h) {
        const float inc2 = nsf_wave_incl(r[h] + sh[h], lane);
        const float ph = c2[h] + inc2;
        c2[h] += nsf_lane63(inc2);
        const float sine = sinf(ph * 2.f * 3.14159265358979323846f) * amp;
        const float nz = valid ? nb[(int64_t)t * DIM + h] : 0.f;
        acc += (sine * uv + na * nz) * wv[h];
      }
      if (valid) ob[t] = tanhf(acc + bias);
    }
  }
}

template <int DIM>
void nsf_launch(const float* f0, const float* rand_ini, const float* noise, const float* w, float bias, float* out, float* tab, int B, int L,
                int G, float inv, float amp, float noise_std, float thr, hipStream_t st) {
  const dim3 grid((unsigned)G, (unsigned)B);
  hipLaunchKernelGGL((nsf_source_kernel<DIM, 1>), grid, dim3(NSF_THREADS), 0, st, f0, rand_ini, noise, w, bias, out, tab, L, inv, amp, noise_std, thr);
  hipLaunchKernelGGL((nsf_source_kernel<DIM, 2>), grid, dim3(NSF_THREADS), 0, st, f0, rand_ini, noise, w, bias, out, tab, L, inv, amp, noise_std, thr);
  hipLaunchKernelGGL((nsf_source_kernel<DIM, 3>), grid, dim3(NSF_THREADS), 0, st, f0, rand_ini, noise, w, bias, out, tab, L, inv, amp, noise_std, thr);
}

}  // namespace

extern "C" int ptpp_nsf_source_supported(int dim) { return dim == 9 || dim == 1 ? 1 : 0; }

static int nsf_groups(int B, int L) {  // workgroups per utterance: fill the chip, keep >= 4 tiles per wave
  int G = 256 / (B > 0 ? B : 1);
  if (G > 16) G = 16;
  while (G > 1 && (int64_t)L < (int64_t)G * NSF_WAVES * 256) --G;
  return G < 1 ? 1 : G;
}

extern "C" size_t ptpp_nsf_source_scratch_bytes(int B, int L, int dim) {
  return (size_t)2 * B * nsf_groups(B, L) * NSF_WAVES * dim * sizeof(float);
}

extern "C" int ptpp_nsf_source(const float* f0, const float* rand_ini, const float* noise, const float* w, float bias, float* out, int B,
                               int L, int dim, float sampling_rate, float sine_amp, float noise_std, float voiced_threshold, void* scratch,
                               size_t scratch_bytes, void* stream) {
  PTPP_CHECK_ARG(f0 && rand_ini && noise && w && out && scratch && B > 0 && L > 0, "nsf_source: bad args");
  PTPP_CHECK_ARG(ptpp_nsf_source_supported(dim), "nsf_source: %d harmonics + 1 not instantiated (9 or 1)", dim);
  PTPP_CHECK_ARG(scratch_bytes >= ptpp_nsf_source_scratch_bytes(B, L, dim), "nsf_source: scratch too small (%zu bytes)", scratch_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float inv = 1.f / sampling_rate;
  const int G = nsf_groups(B, L);
  float* tab = reinterpret_cast<float*>(scratch);
  if (dim == 9) nsf_launch<9>(f0, rand_ini, noise, w, bias, out, tab, B, L, G, inv, sine_amp, noise_std, voiced_threshold, st);
  else nsf_launch<1>(f0, rand_ini, noise, w, bias, out, tab, B, L, G, inv, sine_amp, noise_std, voiced_threshold, st);
  PTPP_CHECK_LAUNCH("nsf_source");
  return PTPP_OK;
}
