// Micro-benchmark (round 5): why do the Snake's v_fmac_f32 (VGPR operands only) average 4.2 cycles in the real kernel
// (SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU) when valu_rate.hip measures 2.3?  Candidates: dependency distance (4 chains per
// thread in the FIRs) and VGPR bank conflicts (operands of one instruction in the same bank = register number mod 4).
// Registers are pinned by hand: accumulators v[ACC0 + i], multiplicands v[B0 + ...], v[C0 + ...].
//   build: hipcc -O3 --offload-arch=gfx950 valu_bank.hip -o valu_bank.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// NCH chains; the multiplicand registers of chain i: v[40 + (i*SB) % 8], v[48 + (i*SC) % 8]; accumulators v[32 + i]
template <int NT, int NCH, int SB, int SC, int OB, int OC>
__global__ __launch_bounds__(NT) void k(uint32_t* out, uint64_t* cyc, int iters) {
  const int tid = threadIdx.x;
  asm volatile(
      "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n"
      "v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n"
      "v_mov_b32 v40, 0.5\n v_mov_b32 v41, 0.5\n v_mov_b32 v42, 0.5\n v_mov_b32 v43, 0.5\n"
      "v_mov_b32 v44, 0.5\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n"
      "v_mov_b32 v48, 0.5\n v_mov_b32 v49, 0.5\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.5\n"
      "v_mov_b32 v52, 0.5\n v_mov_b32 v53, 0.5\n v_mov_b32 v54, 0.5\n v_mov_b32 v55, 0.5\n" ::
          : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
            "v49", "v50", "v51", "v52", "v53", "v54", "v55");
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64 / NCH; ++u) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        constexpr int dummy = 0;
        // registers as immediates in the asm text
        switch ((i % 8)) {
#define ONE(I)                                                                                                              \
  case I:                                                                                                                   \
    asm volatile("v_fmac_f32 v%0, v%1, v%2" ::"n"(32 + I), "n"(40 + (I * SB + OB) % 8), "n"(48 + (I * SC + OC) % 8)          \
                 : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");                                                   \
    break;
          ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7)
#undef ONE
        }
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s;
  asm volatile("v_add_f32 %0, v32, v33" : "=v"(s)::"v32", "v33");
  if (s == 1.2345f) out[tid] = 1;
  if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}

template <int NT, int NCH, int SB, int SC, int OB, int OC>
double run(uint32_t* out, uint64_t* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NT, NCH, SB, SC, OB, OC>), dim3(256), dim3(NT), 0, 0, out, cyc, 100);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<NT, NCH, SB, SC, OB, OC>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(16);
  hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < NT / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  return (double)mx / ((double)iters * 64 * (NT / 256));
}

#define ROW(NAME, NCH, SB, SC, OB, OC)                                                              \
  printf("%-64s %6.2f %6.2f %6.2f\n", NAME, run<256, NCH, SB, SC, OB, OC>(out, cyc), run<512, NCH, SB, SC, OB, OC>(out, cyc), \
         run<1024, NCH, SB, SC, OB, OC>(out, cyc));

int main() {
  uint32_t* out;
  uint64_t* cyc;
  hipMalloc(&out, 4096 * 4);
  hipMalloc(&cyc, 256 * 16 * 8);
  printf("cycles per v_fmac_f32 per SIMD at 1 / 2 / 4 waves per SIMD; acc = v[32+i], b = v[40 + ..], c = v[48 + ..]\n");
  ROW("8 chains, b = v40, c = v48 for all (banks acc i%4, b 0, c 0)", 8, 0, 0, 0, 0)
  ROW("8 chains, b = v[40+i], c = v[48+i] (all three operands in bank i%4)", 8, 1, 1, 0, 0)
  ROW("8 chains, b = v[40+i+1], c = v[48+i+2] (three different banks)", 8, 1, 1, 1, 2)
  ROW("8 chains, b = v[40+i+1], c = v[48+i+1] (b, c same bank, acc other)", 8, 1, 1, 1, 1)
  ROW("8 chains, b = v[40+i], c = v[48+i+1] (acc, b same bank)", 8, 1, 1, 0, 1)
  ROW("4 chains, b = v40, c = v48", 4, 0, 0, 0, 0)
  ROW("4 chains, three different banks", 4, 1, 1, 1, 2)
  ROW("2 chains, b = v40, c = v48", 2, 0, 0, 0, 0)
  ROW("1 chain,  b = v40, c = v48", 1, 0, 0, 0, 0)
  return 0;
}
