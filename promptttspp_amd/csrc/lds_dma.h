// LDS-DMA helpers shared by the kernels that stream operands global -> LDS with global_load_lds_dwordx4
// (conv1d_glds.h, diffnet_layer.hip).  The DMA is issued from inline asm: the compiler neither counts nor drains it, so
// the plain C++ LDS reads keep their compiler-scheduled lgkmcnt ladders and the waits are counted by hand (glds_wait).
#pragma once
#include "ptpp_common.h"

namespace {

__device__ uint4 g_conv_zero_page[64];

// one wave-instruction: lane l's 16 bytes at gsrc -> LDS byte address lds_dst + 16 l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// the same with a wave-uniform base (SGPR pair) and a 32-bit byte offset per lane: no 64-bit lane pointers to keep
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void glds_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

}  // namespace
