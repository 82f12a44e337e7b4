// Multi-tensor fused optimiser step for the trainer (trainers/tts.py:206-211):
//   clip_grad_norm_(max_norm) -> AdamW(lr, betas, eps, weight_decay)
// Two launches for ALL parameters (the reference issues ~10 small kernels per
// parameter tensor): (1) sum of squares of every gradient into one f32 scalar,
// (2) the update, which derives the clip coefficient from that scalar on the
// device -- no host synchronisation, so the step can sit inside a hipGraph.
// HBM-bound: 28 B/parameter (p r/w, g r, m r/w, v r/w).  A block finds its tensor through the
// caller's block -> tensor map (one load; the binary search over ~700 records it replaces was ten
// dependent loads before the first byte of payload), and the sum of squares is spread over 64
// slots (17k blocks adding into ONE float serialised the launch: 246 us for a 280 MB read).
#include "ptpp_common.h"

namespace {

struct TensorRef {
  float* p;
  const float* g;
  float* m;
  float* v;
  long long n;      // elements
  long long block0; // first block index of this tensor in the flat grid
};

constexpr int CHUNK = 256 * 4 * 4;  // elements per block (4 float4 per thread)
static_assert(PTPP_SUMSQ_SLOTS == 64, "the update reads one slot per lane of a wave");

__device__ __forceinline__ int find_tensor(const TensorRef* __restrict__ refs, int nt, long long blk,
                                           const int* __restrict__ block_map) {
  if (block_map) return block_map[blk];
  int lo = 0, hi = nt - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (refs[mid].block0 <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const TensorRef* __restrict__ refs, int nt,
                                                         const int* __restrict__ block_map, float* __restrict__ out) {
  const int ti = find_tensor(refs, nt, blockIdx.x, block_map);
  const TensorRef r = refs[ti];
  const long long base = (blockIdx.x - r.block0) * (long long)CHUNK;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
    if (i + 3 < r.n && (reinterpret_cast<uintptr_t>(r.g) & 15) == 0) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + i);
      s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
    } else {
      for (long long j = i; j < r.n && j < i + 4; ++j) s += r.g[j] * r.g[j];
    }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + (blockIdx.x % PTPP_SUMSQ_SLOTS), red[0] + red[1] + red[2] + red[3]);
}

// Deterministic variant: every block STORES its partial sum (no atomics), a second launch adds the partials of each
// slot in a fixed order.  With f32 atomics the order of the ~275 additions per slot varies from run to run, the norm --
// hence the clip factor, hence every parameter -- could differ in the last bit between the ranks of a data-parallel job
// although their gradients were identical (tests/test_dp_gpu.py).
__global__ __launch_bounds__(256) void grad_sumsq_partial_kernel(const TensorRef* __restrict__ refs, int nt,
                                                                 const int* __restrict__ block_map, float* __restrict__ part) {
  const int ti = find_tensor(refs, nt, blockIdx.x, block_map);
  const TensorRef r = refs[ti];
  const long long base = (blockIdx.x - r.block0) * (long long)CHUNK;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
    if (i + 3 < r.n && (reinterpret_cast<uintptr_t>(r.g) & 15) == 0) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + i);
      s += g[0] * g[0] + g[1] * g[1] + g[2] * g[2] + g[3] * g[3];
    } else {
      for (long long j = i; j < r.n && j < i + 4; ++j) s += r.g[j] * r.g[j];
    }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// slot s = sum of part[b], b = s, s + SLOTS, s + 2 SLOTS, ... : per thread a strided serial sum, then a fixed tree
__global__ __launch_bounds__(256) void grad_sumsq_finish_kernel(const float* __restrict__ part, long long nblk,
                                                                float* __restrict__ out) {
  const int slot = blockIdx.x;
  float s = 0.f;
  for (long long b = slot + (long long)threadIdx.x * PTPP_SUMSQ_SLOTS; b < nblk; b += 256LL * PTPP_SUMSQ_SLOTS) s += part[b];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[slot] = red[0];
}

__global__ __launch_bounds__(256) void adamw_kernel(const TensorRef* __restrict__ refs, int nt,
                                                    const int* __restrict__ block_map,
                                                    const float* __restrict__ sumsq, const float* __restrict__ lr_ptr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                    float max_norm) {
  const int ti = find_tensor(refs, nt, blockIdx.x, block_map);
  const TensorRef r = refs[ti];
  const long long base = (blockIdx.x - r.block0) * (long long)CHUNK;
  const float lr = *lr_ptr;
  float clip = 1.f;
  if (max_norm > 0.f) {
    const float norm = sqrtf(wave_sum(sumsq[threadIdx.x & (PTPP_SUMSQ_SLOTS - 1)]));
    clip = fminf(1.f, max_norm / (norm + 1e-6f));  // torch.nn.utils.clip_grad_norm_
  }
  const float step = lr / bc1, rbc2 = rsqrtf(bc2);
  const bool vec = ((reinterpret_cast<uintptr_t>(r.p) | reinterpret_cast<uintptr_t>(r.g) | reinterpret_cast<uintptr_t>(r.m) |
                     reinterpret_cast<uintptr_t>(r.v)) & 15) == 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i0 = base + ((long long)k * 256 + threadIdx.x) * 4;
    if (vec && i0 + 3 < r.n) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + i0) * clip;
      f32x4 p = *reinterpret_cast<const f32x4*>(r.p + i0), m = *reinterpret_cast<const f32x4*>(r.m + i0),
            v = *reinterpret_cast<const f32x4*>(r.v + i0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[e] *= 1.f - lr * wd;
        m[e] = beta1 * m[e] + (1.f - beta1) * g[e];
        v[e] = beta2 * v[e] + (1.f - beta2) * g[e] * g[e];
        p[e] -= step * m[e] / (sqrtf(v[e]) * rbc2 + eps);
      }
      *reinterpret_cast<f32x4*>(r.p + i0) = p;
      *reinterpret_cast<f32x4*>(r.m + i0) = m;
      *reinterpret_cast<f32x4*>(r.v + i0) = v;
      continue;
    }
    for (int e = 0; e < 4; ++e) {
      const long long i = i0 + e;
      if (i >= r.n) break;
      const float g = r.g[i] * clip;
      float p = r.p[i], m = r.m[i], v = r.v[i];
      p *= 1.f - lr * wd;
      m = beta1 * m + (1.f - beta1) * g;
      v = beta2 * v + (1.f - beta2) * g * g;
      p -= step * m / (sqrtf(v) * rbc2 + eps);
      r.p[i] = p; r.m[i] = m; r.v[i] = v;
    }
  }
}

}  // namespace

// refs: device array of `nt` records {p, g, m, v, n, block0} (6 x 8 bytes each, see
// promptttspp_amd/optim.py); total_blocks = sum over tensors of ceil(n / 4096); block_map (nullable):
// tensor index of every block.
extern "C" int ptpp_grad_sumsq(const void* refs, int nt, const int32_t* block_map, long long total_blocks, float* sumsq,
                               void* stream) {
  PTPP_CHECK_ARG(refs && sumsq && nt > 0 && total_blocks > 0, "grad_sumsq: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(sumsq, 0, sizeof(float) * PTPP_SUMSQ_SLOTS, st) != hipSuccess) { ptpp_set_error("grad_sumsq: memset failed"); return PTPP_ELAUNCH; }
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, (const TensorRef*)refs, nt, block_map,
                     sumsq);
  PTPP_CHECK_LAUNCH("grad_sumsq");
  return PTPP_OK;
}

// As ptpp_grad_sumsq, bit-reproducible: `partials` = total_blocks floats of caller-owned scratch.
extern "C" int ptpp_grad_sumsq_det(const void* refs, int nt, const int32_t* block_map, long long total_blocks, float* sumsq,
                                   float* partials, void* stream) {
  PTPP_CHECK_ARG(refs && sumsq && partials && nt > 0 && total_blocks > 0, "grad_sumsq_det: bad args");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(grad_sumsq_partial_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, (const TensorRef*)refs, nt,
                     block_map, partials);
  hipLaunchKernelGGL(grad_sumsq_finish_kernel, dim3(PTPP_SUMSQ_SLOTS), dim3(256), 0, st, partials, total_blocks, sumsq);
  PTPP_CHECK_LAUNCH("grad_sumsq_det");
  return PTPP_OK;
}

extern "C" int ptpp_adamw_step(const void* refs, int nt, const int32_t* block_map, long long total_blocks,
                               const float* sumsq, const float* lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float max_norm, void* stream) {
  PTPP_CHECK_ARG(refs && lr && nt > 0 && total_blocks > 0 && step >= 1, "adamw_step: bad args");
  PTPP_CHECK_ARG(max_norm <= 0.f || sumsq, "adamw_step: clipping needs sumsq");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, (const TensorRef*)refs, nt, block_map,
                     sumsq, lr, beta1, beta2, eps, weight_decay, bc1, bc2, max_norm);
  PTPP_CHECK_LAUNCH("adamw_step");
  return PTPP_OK;
}
