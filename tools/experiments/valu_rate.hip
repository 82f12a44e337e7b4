// Micro-benchmark (round 5): issue cost of the VALU instructions the anti-aliased Snake is made of, per wave64
// instruction, with 1 / 2 / 4 waves per SIMD (256 / 512 / 1024-thread workgroups, one per CU).  8 independent
// dependency chains per wave, 16 x 8 instructions per loop trip.
//   build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(2))) float f32x2;

#define OP1(ASM)                                                       \
  _Pragma("unroll") for (int u = 0; u < 16; ++u) {                     \
    asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c));                     \
    asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));                     \
  }

enum Op { FMA, PKFMA, PKMUL, PKADD, DOT2F16, DOT2BF16, DOT2CF16, DOT2CBF16, SIN, EXP, RCP, PKFMAF16, CVTPKBF16, CVTPKF16, PERM,
          CNDMASK, CNDMASK_S, CNDMASK_DEP, FMA_S, FMA_S2, FMAC, FMAAK, CMP, MULF32, ADDF32, LSHL, ANDOR, MAXMIN, FMAMIX, NOPS };

template <int OP, int NT>
__global__ __launch_bounds__(NT) void valu(uint32_t* out, uint64_t* cyc, int iters) {
  const int tid = threadIdx.x;
  if constexpr (OP == PKFMA || OP == PKMUL || OP == PKADD) {
    f32x2 a0 = {1.f, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    f32x2 b = {0.999f + tid * 1e-9f, 1.0001f}, c = {1e-6f, 1e-7f};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if constexpr (OP == PKFMA) { OP1("v_pk_fma_f32 %0, %0, %1, %2") }
      if constexpr (OP == PKMUL) { OP1("v_pk_mul_f32 %0, %0, %1") }
      if constexpr (OP == PKADD) { OP1("v_pk_add_f32 %0, %0, %2") }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const float s = a0[0] + a1[1] + a2[0] + a3[1] + a4[0] + a5[1] + a6[0] + a7[1];
    if (s == 1.2345f) out[blockIdx.x * NT + tid] = 1;
    if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
  } else {
    float a0 = 1.f + tid * 1e-7f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 1e-6f;
    asm volatile("s_mov_b32 s20, 0x3f7fbe77\n\ts_mov_b32 s21, 0x0000ffff" ::: "s20", "s21");
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      if constexpr (OP == FMA) { OP1("v_fma_f32 %0, %0, %1, %2") }
      if constexpr (OP == DOT2F16) { OP1("v_dot2_f32_f16 %0, %1, %2, %0") }
      if constexpr (OP == DOT2BF16) { OP1("v_dot2_f32_bf16 %0, %1, %2, %0") }
      if constexpr (OP == DOT2CF16) { OP1("v_dot2c_f32_f16 %0, %1, %2") }
      if constexpr (OP == DOT2CBF16) { OP1("v_dot2c_f32_bf16 %0, %1, %2") }
      if constexpr (OP == SIN) { OP1("v_sin_f32 %0, %0") }
      if constexpr (OP == EXP) { OP1("v_exp_f32 %0, %0") }
      if constexpr (OP == RCP) { OP1("v_rcp_f32 %0, %0") }
      if constexpr (OP == PKFMAF16) { OP1("v_pk_fma_f16 %0, %0, %1, %2") }
      if constexpr (OP == CVTPKBF16) { OP1("v_cvt_pk_bf16_f32 %0, %0, %1") }
      if constexpr (OP == CVTPKF16) { OP1("v_cvt_pkrtz_f16_f32 %0, %0, %1") }
      if constexpr (OP == PERM) { OP1("v_perm_b32 %0, %0, %1, %2") }
      if constexpr (OP == CNDMASK) { OP1("v_cndmask_b32 %0, %0, %1, vcc") }
      if constexpr (OP == CNDMASK_S) { OP1("v_cndmask_b32 %0, %0, %1, s[20:21]") }
      if constexpr (OP == CNDMASK_DEP) { OP1("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc") }
      if constexpr (OP == FMA_S) { OP1("v_fma_f32 %0, %0, s20, %2") }
      if constexpr (OP == FMA_S2) { OP1("v_fma_f32 %0, %1, s20, %0") }
      if constexpr (OP == FMAC) { OP1("v_fmac_f32 %0, %1, %2") }
      if constexpr (OP == FMAAK) { OP1("v_fmac_f32 %0, s20, %1") }
      if constexpr (OP == CMP) { OP1("v_cmp_lt_f32 vcc, %0, %1") }
      if constexpr (OP == MULF32) { OP1("v_mul_f32 %0, %0, %1") }
      if constexpr (OP == ADDF32) { OP1("v_add_f32 %0, %0, %2") }
      if constexpr (OP == LSHL) { OP1("v_lshlrev_b32 %0, 16, %0") }
      if constexpr (OP == ANDOR) { OP1("v_and_or_b32 %0, %0, %1, %2") }
      if constexpr (OP == MAXMIN) { OP1("v_med3_i32 %0, %0, %1, %2") }
      if constexpr (OP == FMAMIX) { OP1("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]") }
      if constexpr (OP == NOPS) { OP1("s_nop 0") }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 1.2345f) out[blockIdx.x * NT + tid] = 1;
    if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
  }
}

template <int OP, int NT>
double run(uint32_t* out, uint64_t* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((valu<OP, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, 100);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((valu<OP, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(16);
  hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < NT / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  // cycles per wave-instruction per SIMD: (NT / 256) waves share a SIMD
  return (double)mx / ((double)iters * 128 * (NT / 256));
}

#define ROW(OP)                                                                                         \
  printf("%-12s  %6.2f  %6.2f  %6.2f   cycles per wave64 instruction per SIMD (1 / 2 / 4 waves per SIMD)\n", #OP, \
         run<OP, 256>(out, cyc), run<OP, 512>(out, cyc), run<OP, 1024>(out, cyc));

int main() {
  uint32_t* out;
  uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&cyc, 256 * 16 * 8);
  ROW(NOPS) ROW(FMA) ROW(MULF32) ROW(ADDF32) ROW(PKFMA) ROW(PKMUL) ROW(PKADD) ROW(FMAMIX)
  ROW(DOT2F16) ROW(DOT2BF16) ROW(DOT2CF16) ROW(DOT2CBF16) ROW(PKFMAF16)
  ROW(SIN) ROW(EXP) ROW(RCP) ROW(CVTPKBF16) ROW(CVTPKF16) ROW(PERM) ROW(CNDMASK) ROW(CNDMASK_S) ROW(CNDMASK_DEP) ROW(FMA_S) ROW(FMA_S2) ROW(FMAC) ROW(FMAAK) ROW(CMP) ROW(LSHL) ROW(ANDOR) ROW(MAXMIN)
  return 0;
}
