// Micro-benchmark (round 5): do a VALU-bound wave and an MFMA-bound wave of the SAME SIMD overlap?
// One workgroup per CU; per SIMD NV waves run a v_fmac_f32 (VGPR operands) loop and NM waves a v_mfma_f32_16x16x32_bf16 loop.
// Times: VALU waves alone, MFMA waves alone, both together (cycles of the slowest wave of block 0).
//   build: hipcc -O3 --offload-arch=gfx950 valu_mfma_overlap.hip -o valu_mfma_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// mode bit 0: VALU waves work, bit 1: MFMA waves work.  Waves [0, 4 NM) are MFMA waves, the rest VALU waves
template <int NT, int NM, int VOP, bool VARY, int PRIO, int AG>
__global__ __launch_bounds__(NT) void k(uint32_t* out, uint64_t* cyc, int iters, int mode) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const bool is_mfma = wave < 4 * NM;
  uint64_t t0 = 0, t1 = 0;
  if (is_mfma) {
    if (mode & 2) {
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
      u32x4 a[8], b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = u32x4{0x3f803f80u + tid + i, 0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u}; b[i] = a[i]; asm volatile("" : "+v"(a[i]), "+v"(b[i])); }
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if constexpr (AG == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[VARY ? i : 0]), "v"(b[VARY ? (i + 3) & 7 : 0]));
            if constexpr (AG == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[VARY ? i : 0]), "v"(b[VARY ? (i + 3) & 7 : 0]));
            if constexpr (AG == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a[VARY ? i : 0]), "a"(b[VARY ? (i + 3) & 7 : 0]));
          }
      }
      t1 = __builtin_readcyclecounter();
      float s = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += acc[i][0];
      if (s == 1.2345f) out[tid] = 1;
    }
  } else if (mode & 1) {
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    float a0 = 1.f + tid * 1e-7f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 1e-6f;
    asm volatile("v_mov_b32 %0, %0" : "+v"(b));
    asm volatile("v_mov_b32 %0, %0" : "+v"(c));
    float bb[8], cc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bb[i] = b + i * 1e-6f; cc[i] = c * (i + 1); asm volatile("" : "+v"(bb[i]), "+v"(cc[i])); }
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (VOP == 0) {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(VARY ? bb[0] : b), "v"(VARY ? cc[5] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(VARY ? bb[1] : b), "v"(VARY ? cc[6] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(VARY ? bb[2] : b), "v"(VARY ? cc[7] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(VARY ? bb[3] : b), "v"(VARY ? cc[0] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a4) : "v"(VARY ? bb[4] : b), "v"(VARY ? cc[1] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a5) : "v"(VARY ? bb[5] : b), "v"(VARY ? cc[2] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a6) : "v"(VARY ? bb[6] : b), "v"(VARY ? cc[3] : c));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a7) : "v"(VARY ? bb[7] : b), "v"(VARY ? cc[4] : c));
        } else {
          asm volatile("v_sin_f32 %0, %0" : "+v"(a0));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a1));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a2));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a3));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a4));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a5));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a6));
          asm volatile("v_sin_f32 %0, %0" : "+v"(a7));
        }
      }
    }
    t1 = __builtin_readcyclecounter();
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 1.2345f) out[tid] = 1;
  }
  if ((tid & 63) == 0) cyc[blockIdx.x * 32 + wave] = t1 - t0;
}

template <int NT, int NM, int VOP, bool VARY, int PRIO = 0, int AG = 0>
void run(const char* name, uint32_t* out, uint64_t* cyc) {
  const int iters = 4000;
  double res[4] = {0, 0, 0, 0};
  double resm[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((k<NT, NM, VOP, VARY, PRIO, AG>), dim3(256), dim3(NT), 0, 0, out, cyc, 100, mode);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NT, NM, VOP, VARY, PRIO, AG>), dim3(256), dim3(NT), 0, 0, out, cyc, iters, mode);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(32);
    hipMemcpy(h.data(), cyc, 32 * 8, hipMemcpyDeviceToHost);
    uint64_t mv = 0, mm = 0;
    for (int w = 0; w < NT / 64; ++w) {
      if (w < 4 * NM) mm = h[w] > mm ? h[w] : mm;
      else mv = h[w] > mv ? h[w] : mv;
    }
    res[mode] = (double)mv / iters;
    resm[mode] = (double)mm / iters;
  }
  // per iteration: a VALU wave issues 64 instructions, an MFMA wave 16 MFMAs (= 256 cycles of its SIMD's matrix pipe)
  printf("%-66s VALU alone %7.1f  MFMA alone %7.1f  together: VALU %7.1f  MFMA %7.1f   cycles / iteration\n", name, res[1], resm[2],
         res[3], resm[3]);
}

int main() {
  uint32_t* out;
  uint64_t* cyc;
  hipMalloc(&out, 4096 * 4);
  hipMalloc(&cyc, 256 * 32 * 8);
  run<512, 1, 0, false>("1 MFMA + 1 fmac wave per SIMD, same operand registers", out, cyc);
  run<768, 1, 0, false>("1 MFMA + 2 fmac waves per SIMD, same", out, cyc);
  run<768, 2, 0, false>("2 MFMA + 1 fmac wave per SIMD, same", out, cyc);
  run<1024, 2, 0, false>("2 MFMA + 2 fmac waves per SIMD, same", out, cyc);
  run<512, 1, 0, true>("1 MFMA + 1 fmac wave per SIMD, varied operand registers", out, cyc);
  run<768, 1, 0, true>("1 MFMA + 2 fmac waves per SIMD, varied", out, cyc);
  run<768, 2, 0, true>("2 MFMA + 1 fmac wave per SIMD, varied", out, cyc);
  run<1024, 2, 0, true>("2 MFMA + 2 fmac waves per SIMD, varied", out, cyc);
  run<512, 1, 0, true, 3>("1 MFMA + 1 fmac wave per SIMD, varied, VALU waves s_setprio 3", out, cyc);
  run<768, 1, 0, true, 3>("1 MFMA + 2 fmac waves per SIMD, varied, prio 3", out, cyc);
  run<768, 2, 0, true, 3>("2 MFMA + 1 fmac wave per SIMD, varied, prio 3", out, cyc);
  run<1024, 2, 0, true, 3>("2 MFMA + 2 fmac waves per SIMD, varied, prio 3", out, cyc);
  run<1024, 2, 1, false, 3>("2 MFMA + 2 v_sin waves per SIMD, prio 3", out, cyc);
  run<1024, 2, 0, true, 0, 1>("2 MFMA + 2 fmac waves per SIMD, varied, accumulators in AGPRs", out, cyc);
  run<1024, 2, 0, true, 0, 2>("2 MFMA + 2 fmac waves per SIMD, varied, acc + A + B in AGPRs", out, cyc);
  run<768, 2, 0, true, 0, 2>("2 MFMA + 1 fmac wave per SIMD, varied, acc + A + B in AGPRs", out, cyc);
  run<1024, 2, 0, true, 3, 2>("2 MFMA + 2 fmac waves per SIMD, varied, all AGPR, VALU prio 3", out, cyc);
  run<512, 1, 1, false>("1 MFMA + 1 v_sin wave per SIMD", out, cyc);
  run<1024, 2, 1, false>("2 MFMA + 2 v_sin waves per SIMD", out, cyc);
  return 0;
}
