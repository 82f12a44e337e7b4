// Shared device/host helpers for libptpp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ptpp.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef unsigned short bf16_raw;  // storage type of a bf16 element
struct f16_raw {                  // storage type of an IEEE half element (a distinct type: the kernels are templated on it)
  unsigned short v;
};

// ---- error plumbing ---------------------------------------------------------
void ptpp_set_error(const char* fmt, ...);

#define PTPP_CHECK_ARG(cond, ...)  \
  do {                             \
    if (!(cond)) {                 \
      ptpp_set_error(__VA_ARGS__); \
      return PTPP_EINVAL;          \
    }                              \
  } while (0)

#define PTPP_CHECK_LAUNCH(name)                                       \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      ptpp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return PTPP_ELAUNCH;                                            \
    }                                                                 \
  } while (0)

// ---- "the dynamic-LDS limit of this kernel was raised on this device" (hipFuncSetAttribute is per device; the launchers call it
// once per (device, kernel) instead of before every launch).  Per translation unit, guarded by a mutex: launches may come from
// the autograd thread and the main thread.
#include <mutex>
namespace {
struct LdsLimitSeen {
  std::mutex mu;
  struct Key { int dev; const void* k; } seen[64];
  int n = 0;
};
inline LdsLimitSeen& lds_limit_table() { static LdsLimitSeen t; return t; }
inline bool lds_limit_raised(const void* kern) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  LdsLimitSeen& t = lds_limit_table();
  std::lock_guard<std::mutex> g(t.mu);
  for (int i = 0; i < t.n; ++i)
    if (t.seen[i].dev == dev && t.seen[i].k == kern) return true;
  return false;
}
inline void lds_limit_mark(const void* kern) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  LdsLimitSeen& t = lds_limit_table();
  std::lock_guard<std::mutex> g(t.mu);
  if (t.n < 64) t.seen[t.n++] = {dev, kern};
}
}  // namespace

// raise a kernel's dynamic-LDS limit; false -- with the error text set -- when the runtime refuses (every launcher checks it)
inline bool ptpp_lds_limit(const void* kern, int bytes, const char* who) {
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) return true;
  ptpp_set_error("%s: cannot raise the dynamic LDS limit to %d bytes: %s", who, bytes, hipGetErrorString(e));
  return false;
}

// ---- bf16 <-> f32 (round to nearest even, like torch) -------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_raw v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_raw)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_raw)(u >> 16);
}

// Element traits: how many elements in a 16-byte chunk and vector load/store of
// 4 consecutive elements as f32.
template <typename T>
struct Elem;

template <>
struct Elem<float> {
  static constexpr int PER16 = 4;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ f32x4 ld4(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
  }
  static __device__ __forceinline__ void st4(float* p, f32x4 v) {
    *reinterpret_cast<f32x4*>(p) = v;
  }
};

template <>
struct Elem<bf16_raw> {
  static constexpr int PER16 = 8;
  static __device__ __forceinline__ float ld(const bf16_raw* p) {
    return bf16_to_f32(*p);
  }
  static __device__ __forceinline__ void st(bf16_raw* p, float v) {
    *p = f32_to_bf16(v);
  }
  static __device__ __forceinline__ f32x4 ld4(const bf16_raw* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    f32x4 v;
    v[0] = __uint_as_float(r.x << 16);
    v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16);
    v[3] = __uint_as_float(r.y & 0xffff0000u);
    return v;
  }
  static __device__ __forceinline__ void st4(bf16_raw* p, f32x4 v) {
    uint2 r;
    r.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    r.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = r;
  }
};

template <>
struct Elem<f16_raw> {
  static constexpr int PER16 = 8;
  static __device__ __forceinline__ float ld(const f16_raw* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
  static __device__ __forceinline__ void st(f16_raw* p, float v) { p->v = __builtin_bit_cast(unsigned short, (_Float16)v); }
  static __device__ __forceinline__ f32x4 ld4(const f16_raw* p) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 r = *reinterpret_cast<const h4*>(p);
    return f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
  }
  static __device__ __forceinline__ void st4(f16_raw* p, f32x4 v) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    *reinterpret_cast<h4*>(p) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  }
};

// two 16-bit elements in a 32-bit word (element 0 in the low half)
template <typename T>
struct H2;
template <>
struct H2<bf16_raw> {
  static __device__ __forceinline__ float lo(uint32_t r) { return __uint_as_float(r << 16); }
  static __device__ __forceinline__ float hi(uint32_t r) { return __uint_as_float(r & 0xffff0000u); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }
};
template <>
struct H2<f16_raw> {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  static __device__ __forceinline__ float lo(uint32_t r) { return (float)__builtin_bit_cast(h2, r)[0]; }
  static __device__ __forceinline__ float hi(uint32_t r) { return (float)__builtin_bit_cast(h2, r)[1]; }
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return __builtin_bit_cast(uint32_t, h2{(_Float16)a, (_Float16)b}); }
};
template <>
struct H2<float> {  // (never used: the paired-fragment epilogue is 16-bit only; keeps the template well-formed)
  static __device__ __forceinline__ float lo(uint32_t r) { return __uint_as_float(r); }
  static __device__ __forceinline__ float hi(uint32_t r) { return __uint_as_float(r); }
  static __device__ __forceinline__ uint32_t pack(float a, float) { return __float_as_uint(a); }
};
template <typename T>
struct IsBf16 { static constexpr bool value = false; };
template <>
struct IsBf16<bf16_raw> { static constexpr bool value = true; };

// ---- activations ------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case PTPP_ACT_RELU:
      return v > 0.f ? v : 0.f;
    case PTPP_ACT_GELU:
      return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case PTPP_ACT_SWISH:
      return v / (1.f + __expf(-v));
    case PTPP_ACT_TANH:
      return tanhf(v);
    case PTPP_ACT_MISH: {
      // x * tanh(softplus(x)); softplus with torch's threshold of 20
      float sp = v > 20.f ? v : log1pf(__expf(v));
      return v * tanhf(sp);
    }
    default:
      return v;
  }
}

// The activation as a compile-time constant.  A kernel that applies act_apply(v, act) to every element of an
// unrolled epilogue gets one copy of the whole switch (erff, tanhf, log1pf expansions) per element: the 128 x 128
// conv kernel was 18 k instructions and the 64 x 64-per-wave variant 110 k, far beyond the 64 KiB instruction cache,
// and its epilogue ran at instruction-fetch speed (measured 11 us for 64 elements per lane with all memory
// operations removed).  Dispatch ONCE per wave (act_dispatch) to code specialised for the activation instead.
template <int ACT>
__device__ __forceinline__ float act_apply_c(float v, int act_rt = 0) {
  if constexpr (ACT < 0) {  // not specialised: the run-time switch (paths where speed does not matter)
    return act_apply(v, act_rt);
  } else if constexpr (ACT == PTPP_ACT_RELU) {
    return v > 0.f ? v : 0.f;
  } else if constexpr (ACT == PTPP_ACT_GELU) {
    return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  } else if constexpr (ACT == PTPP_ACT_SWISH) {
    return v / (1.f + __expf(-v));
  } else if constexpr (ACT == PTPP_ACT_TANH) {
    return tanhf(v);
  } else if constexpr (ACT == PTPP_ACT_MISH) {
    float sp = v > 20.f ? v : log1pf(__expf(v));
    return v * tanhf(sp);
  } else {
    return v;
  }
}
// f(std::integral_constant<int, ACT>) for the run-time activation code (wave-uniform branch)
template <int A>
struct ActTag {
  static constexpr int value = A;
};
// Specialised: the activations the model's conv / linear epilogues use (none, ReLU, GELU, the DiffNet gate); Swish, tanh
// and Mish (offered by the ABI, used by no layer's epilogue) share ONE generic copy (ActTag<-1>: run-time switch).
template <typename F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
  switch (act) {
    case PTPP_ACT_NONE: f(ActTag<PTPP_ACT_NONE>()); break;
    case PTPP_ACT_RELU: f(ActTag<PTPP_ACT_RELU>()); break;
    case PTPP_ACT_GELU: f(ActTag<PTPP_ACT_GELU>()); break;
    case PTPP_ACT_GATE: f(ActTag<PTPP_ACT_GATE>()); break;
    default: f(ActTag<-1>()); break;
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// The WaveNet gate of the bf16 kernels (modules/denoiser.py:76-77): two v_exp_f32 + one v_rcp_f32 instead of libm's tanhf and
// an IEEE division (the fused gate epilogues were VALU-bound: 10 us of a 67 us DiffNet layer launch).  Relative error ~1e-6
// (absolute 1e-7 near tanh's zero), more than two orders below the bf16 rounding of the result; the f32 parity kernels keep
// libm.  Every bf16 site (gate_fwd / gate_bwd kernels, the fused conv epilogues, the one-launch layer) uses these same
// expressions, so the paths stay bit-identical to each other.
__device__ __forceinline__ float gate_fast(float s, float f) {  // sigmoid(s) * tanh(f)
  const float e2 = __expf(2.f * fminf(fmaxf(f, -15.f), 15.f));
  const float es = __expf(-s);
  return (e2 - 1.f) * __builtin_amdgcn_rcpf((e2 + 1.f) * (1.f + es));
}
__device__ __forceinline__ void gate_fast_parts(float s, float f, float& sg, float& th) {  // sigmoid(s), tanh(f)
  const float e2 = __expf(2.f * fminf(fmaxf(f, -15.f), 15.f));
  sg = __builtin_amdgcn_rcpf(1.f + __expf(-s));
  th = (e2 - 1.f) * __builtin_amdgcn_rcpf(e2 + 1.f);
}

// ---- wave reductions (wave = 64 lanes) ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- cross-block column sums --------------------------------------------------------
// f32 atomics from many blocks on the same cache line serialise on MI355X: measured ~50 ns per
// block visit (2125 blocks adding 32 columns into one 128-byte line = 114 us around a 35 MB read).
// A ticket counter + "last block sums the partials" is no better here: the ticket is the same
// contended atomic and its release fence writes the XCD's dirty L2 lines back (173 us measured
// for the LayerNorm backward).  So: block b adds its KC block totals into replica b % 32 of a
// caller-provided scratch (32 x fewer visits per line, the replicas sit on different lines /
// channels), and a one-wave-per-64-columns finishing launch sums the replicas, delivers the
// totals and zeroes the scratch again (include/ptpp.h "Reduction scratch").
constexpr int PTPP_RED_NREP = PTPP_RED_REPLICAS;
// every thread of the block calls this; tot = the block's KC totals in LDS (visible to the block)
__device__ __forceinline__ void red_block_add(void* scratch, const float* tot, int KC) {
  float* mine = reinterpret_cast<float*>(scratch) + (size_t)(blockIdx.x % PTPP_RED_NREP) * KC;
  for (int c = threadIdx.x; c < KC; c += blockDim.x) atomicAdd(mine + c, tot[c]);
}
// column c < n0 goes to dst0[c], the rest to dst1[c - n0] (either may be NULL: dropped);
// accumulate: dst += total, else dst = total
static __global__ void red_sum_kernel(float* __restrict__ scratch, int KC, float* __restrict__ dst0, int n0,
                                      float* __restrict__ dst1, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= KC) return;
  float v[PTPP_RED_NREP];
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) v[r] = scratch[(size_t)r * KC + c];
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) {
    s += v[r];
    scratch[(size_t)r * KC + c] = 0.f;
  }
  float* d = c < n0 ? (dst0 ? dst0 + c : nullptr) : (dst1 ? dst1 + (c - n0) : nullptr);
  if (d) *d = accumulate ? *d + s : s;
}
// ... the totals OVERWRITE dst[0 .. KC) (a consumer launch reads them at once) and are also ADDED to acc0 / acc1 (columns < n0 /
// the rest; either may be NULL) -- BatchNorm's backward needs its two sums for dx and owes them to the parameter gradients
static __global__ void red_sum_both_kernel(float* __restrict__ scratch, int KC, float* __restrict__ dst, float* __restrict__ acc0, int n0,
                                           float* __restrict__ acc1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= KC) return;
  float v[PTPP_RED_NREP];
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) v[r] = scratch[(size_t)r * KC + c];
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < PTPP_RED_NREP; ++r) {
    s += v[r];
    scratch[(size_t)r * KC + c] = 0.f;
  }
  dst[c] = s;
  float* d = c < n0 ? (acc0 ? acc0 + c : nullptr) : (acc1 ? acc1 + (c - n0) : nullptr);
  if (d) *d += s;
}
inline void red_sum_launch(void* scratch, int KC, float* dst0, int n0, float* dst1, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(red_sum_kernel, dim3((KC + 63) / 64), dim3(64), 0, st, reinterpret_cast<float*>(scratch), KC, dst0,
                     n0, dst1, accumulate);
}
inline bool red_scratch_ok(const void* scratch, size_t bytes, int KC) {
  return scratch && bytes >= (size_t)PTPP_RED_NREP * KC * sizeof(float);
}

// ---- deferred finishing of PARAMETER-GRADIENT sums (red.hip; include/ptpp.h "Deferred reduction") ---------------------------
// A parameter gradient (LayerNorm dgamma / dbeta, the attention position biases, the pitch embedding) is read by nobody before the
// optimiser / the gradient exchange, so its finishing launch need not follow its producer: with deferral on (ptpp_red_defer) the
// producer gets a private slice of a per-stream arena instead of the shared scratch, the (slice, destination) pair is queued, and
// ptpp_red_flush finishes every queued sum of a stream in ONE launch on that stream.  43 finishing launches per training step
// became 1-2 (profiles/r06_main_order_before.txt).  Sums whose result the next launch reads (BatchNorm statistics) keep
// red_sum_launch.
void* ptpp_red_arena_take(size_t bytes, hipStream_t st);  // NULL: deferral off / suspended / arena full -> finish immediately
void ptpp_red_arena_push(void* slice, int KC, float* dst0, int n0, float* dst1, int accumulate, hipStream_t st);
struct RedSlot {
  void* ptr;
  int deferred;
};
inline RedSlot red_take(void* scratch, size_t scratch_bytes, int KC, hipStream_t st) {
  void* p = ptpp_red_arena_take((size_t)PTPP_RED_NREP * KC * sizeof(float), st);
  if (p) return RedSlot{p, 1};
  return RedSlot{red_scratch_ok(scratch, scratch_bytes, KC) ? scratch : nullptr, 0};
}
inline void red_finish(const RedSlot& s, int KC, float* dst0, int n0, float* dst1, int accumulate, hipStream_t st) {
  if (s.deferred) ptpp_red_arena_push(s.ptr, KC, dst0, n0, dst1, accumulate, st);
  else red_sum_launch(s.ptr, KC, dst0, n0, dst1, accumulate, st);
}

// One element of the reverse-diffusion update (modules/diffusion.py:283-302: predict_start_from_noise, clamp, q_posterior mean,
// + sigma * noise) as the reference's UNFUSED f32 operation sequence -- shared by ptpp_ddpm_step and ptpp_sampler_head so that
// both give the same bits (left to the optimiser, the two kernels contracted different multiply-add pairs).
__device__ __forceinline__ float ddpm_update(float ca, float cb, float k1, float k2, float sg, float xv, float ev, float nv) {
#pragma clang fp contract(off)
  float x0 = ca * xv - cb * ev;
  x0 = fminf(fmaxf(x0, -1.f), 1.f);
  const float mean = k1 * x0 + k2 * xv;
  return mean + sg * nv;
}

// XCD-aware block remap: hardware places block b on XCD b % 8; give each XCD a
// contiguous range of logical tiles so neighbours share L2 (bijective for any n).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// ---- counter-based dropout ---------------------------------------------------
// One 64-bit splitmix hash of (seed, vector index) yields four 16-bit lanes, one
// per element of a 4-wide vector: element e of vector v is KEPT iff
// bits16(e) >= thresh16, thresh16 = round(p * 65536).  The same (seed, index)
// regenerates the mask in the backward pass, so no mask tensor is stored.
__device__ __forceinline__ uint64_t drop_hash(uint64_t seed, uint64_t vec_idx) {
  uint64_t z = seed + (vec_idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ f32x4 drop_mask4(uint64_t seed, uint64_t vec_idx, uint32_t thresh16, float inv_keep) {
  const uint64_t h = drop_hash(seed, vec_idx);
  f32x4 m;
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] = ((uint32_t)(h >> (16 * e)) & 0xffffu) >= thresh16 ? inv_keep : 0.f;
  return m;
}
