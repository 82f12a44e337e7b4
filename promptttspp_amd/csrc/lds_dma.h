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
// N pieces of 1 KiB at consecutive LDS addresses lds_dst + 1024 q from one wave-uniform base and N lane offsets: M0 is saved
// and restored once
template <int N>
__device__ __forceinline__ void glds16_s_n(const void* sbase, const uint32_t (&voff)[N], uint32_t lds_dst) {
  static_assert(N >= 1 && N <= 4, "pieces per call");
  unsigned keep;
  if constexpr (N == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "s"(sbase), "s"(lds_dst) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "s"(sbase), "s"(lds_dst) : "memory", "scc");
  else if constexpr (N == 3)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "s"(sbase), "s"(lds_dst) : "memory", "scc");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(sbase), "s"(lds_dst) : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void glds_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

}  // namespace
