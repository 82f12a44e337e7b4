// Micro-benchmark (round 5): does an MFMA stream whose A / B / C operands live in AGPRs leave the VALU waves of its SIMD their
// issue slots?  (valu_mfma_overlap.hip: with changing VGPR operands it does not.)  Registers are pinned by hand so that the
// compiler inserts no copies: MFMA waves use a[0:31] as accumulators and either v[64:127] or a[64:127] as A / B operands.
//   build: hipcc -O3 --offload-arch=gfx950 agpr_overlap.hip -o agpr_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// 16 MFMAs per iteration: accumulator i in a[4i .. 4i+3], operands from 8 distinct register quads
#define MF(ACC, A, B, PFX) "v_mfma_f32_16x16x32_bf16 a[" #ACC ":" #ACC "+3], " PFX "[" #A ":" #A "+3], " PFX "[" #B ":" #B "+3], a[" #ACC ":" #ACC "+3]\n"
#define MF8(PFX)                                                                                                  \
  MF(0, 32, 44, PFX) MF(4, 36, 48, PFX) MF(8, 40, 52, PFX) MF(12, 44, 56, PFX) MF(16, 48, 60, PFX) MF(20, 52, 32, PFX) \
  MF(24, 56, 36, PFX) MF(28, 60, 40, PFX)

template <int NT, int NM, int AG>
__global__ __launch_bounds__(NT) void k(uint32_t* out, uint64_t* cyc, int iters, int mode) {
  const int tid = threadIdx.x, wave = tid >> 6;
  const bool is_mfma = wave < 4 * NM;
  uint64_t t0 = 0, t1 = 0;
  if (is_mfma) {
    if (mode & 2) {
      // initialise operands (values irrelevant)
      for (int r = 0; r < 1; ++r) {
        asm volatile(
            "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n"
            "v_accvgpr_write_b32 a32, v32\n v_accvgpr_write_b32 a33, v32\n v_accvgpr_write_b32 a34, v32\n v_accvgpr_write_b32 a35, v32\n" ::
                : "v32", "v33", "v34", "v35", "a32", "a33", "a34", "a35");
      }
      t0 = __builtin_readcyclecounter();
      for (int it = 0; it < iters; ++it) {
        if constexpr (AG) asm volatile(MF8("a") MF8("a") ::: "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
        else asm volatile(MF8("v") MF8("v") ::: "memory", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
      }
      t1 = __builtin_readcyclecounter();
    }
  } else if (mode & 1) {
    float a0 = 1.f + tid * 1e-7f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float bb[8], cc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bb[i] = 0.999f + i * 1e-6f; cc[i] = 1e-6f * (i + 1); asm volatile("" : "+v"(bb[i]), "+v"(cc[i])); }
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(bb[0]), "v"(cc[5]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(bb[1]), "v"(cc[6]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(bb[2]), "v"(cc[7]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(bb[3]), "v"(cc[0]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a4) : "v"(bb[4]), "v"(cc[1]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a5) : "v"(bb[5]), "v"(cc[2]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a6) : "v"(bb[6]), "v"(cc[3]));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a7) : "v"(bb[7]), "v"(cc[4]));
      }
    }
    t1 = __builtin_readcyclecounter();
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 1.2345f) out[tid] = 1;
  }
  if ((tid & 63) == 0) cyc[blockIdx.x * 32 + wave] = t1 - t0;
}

template <int NT, int NM, int AG>
void run(const char* name, uint32_t* out, uint64_t* cyc) {
  const int iters = 4000;
  double res[4] = {0, 0, 0, 0}, resm[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((k<NT, NM, AG>), dim3(256), dim3(NT), 0, 0, out, cyc, 100, mode);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NT, NM, AG>), dim3(256), dim3(NT), 0, 0, out, cyc, iters, mode);
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
  printf("%-56s VALU alone %7.1f  MFMA alone %7.1f  together: VALU %7.1f  MFMA %7.1f   cycles / iteration\n", name, res[1], resm[2],
         res[3], resm[3]);
}

int main() {
  uint32_t* out;
  uint64_t* cyc;
  hipMalloc(&out, 4096 * 4);
  hipMalloc(&cyc, 256 * 32 * 8);
  run<1024, 2, 0>("2 MFMA + 2 fmac waves per SIMD, A / B in VGPRs (8 quads)", out, cyc);
  run<1024, 2, 1>("2 MFMA + 2 fmac waves per SIMD, A / B in AGPRs (8 quads)", out, cyc);
  run<768, 2, 0>("2 MFMA + 1 fmac wave per SIMD, A / B in VGPRs", out, cyc);
  run<768, 2, 1>("2 MFMA + 1 fmac wave per SIMD, A / B in AGPRs", out, cyc);
  run<512, 1, 0>("1 MFMA + 1 fmac wave per SIMD, A / B in VGPRs", out, cyc);
  run<512, 1, 1>("1 MFMA + 1 fmac wave per SIMD, A / B in AGPRs", out, cyc);
  return 0;
}
