// Micro-benchmark (round 5): the fragment-read mix of diffnet_layer_kernel's dilated-conv pass, alone -- 8 waves, per step 4 reads
// of a [256 rows][64 B] weight-stage image (rows wn * 64 + fn * 16 + lr, chunk lg ^ swz4) and, every other step, 4 reads of a
// [rows][128 B] activation window (rows wm * 64 + fm * 16 + lr + tap * dil, chunk (kh * 4 + lg) ^ swz8), s_waitcnt lgkmcnt(0) +
// s_barrier per step -- with the images at the bottom of the LDS allocation or above 64 KiB.  profiles/r04_diffnet_layer_phases.txt
// measured 0.33 us per step for these 48 KB (67 B/clk/CU); what does the LDS deliver for exactly this pattern?
//   build: hipcc -O3 --offload-arch=gfx950 lds_read_mimic.hip -o lds_read_mimic.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int BASE_KB, bool BARRIER>
__global__ __launch_bounds__(512) void k(uint32_t* out, uint64_t* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 160 * 1024 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int wm = wave >> 2, wn = wave & 3, lr = lane & 15, lg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + BASE_KB * 1024;
  // weight ring: 4 stages of 16 KiB at lds0; window: 2 x 18 KiB behind it
  const int q0 = wn * 64 + lr;
  uint32_t wa[4];
#pragma unroll
  for (int fn = 0; fn < 4; ++fn) {
    const int q = q0 + fn * 16;
    wa[fn] = lds0 + (q * 4 + (lg ^ ((-(q >> 2)) & 3))) * 16;
  }
  const uint32_t xs = lds0 + 64 * 1024;
  uint32_t x = 0;
  u32x4 w[4], xf[4];
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {  // 8 steps: stage s & 3; x fragments on even steps (tap = s / 2 % 3, kh = (s >> 1) & 1)
      const int st = s & 3;
      DSR(w[0], wa[0], 0); DSR(w[1], wa[1], 0); DSR(w[2], wa[2], 0); DSR(w[3], wa[3], 0);
      if (!(s & 1)) {
        const int tap = (s >> 1) % 3, kh = (s >> 2) & 1;
        const int r = wm * 64 + lr + tap * 8;
        const uint32_t xa = xs + (r * 8 + ((kh * 4 + lg) ^ ((r >> 1) & 7))) * 16;
        DSR(xf[0], xa, 0); DSR(xf[1], xa, 2048); DSR(xf[2], xa, 4096); DSR(xf[3], xa, 6144);
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) wa[fn] += (st == 3 ? -3 : 1) * 16384;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (BARRIER) asm volatile("s_barrier" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(w[i][0]));
      if (!(s & 1)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(xf[i][0]));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (x == 0x12345u) out[blockIdx.x * 512 + tid] = x;
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int BASE_KB, bool BARRIER>
void run(const char* name, uint32_t* out, uint64_t* cyc) {
  const int iters = 4000;
  auto kern = k<BASE_KB, BARRIER>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 160 * 1024, 0, out, cyc, 200);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 160 * 1024, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(16);
  hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
  const double bytes = 8.0 * iters * (8 * 4 + 4 * 4) * 1024.0;  // per iteration and wave: 32 W + 16 x fragment reads of 1 KiB
  printf("%-50s %7.1f B/clk/CU   %6.1f cycles per step (48 KiB)\n", name, bytes / (double)mx, (double)mx / (iters * 8.0));
}

int main() {
  uint32_t* out;
  uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 16 * 8);
  run<0, true>("images at LDS offset 0, barrier per step", out, cyc);
  run<0, false>("images at LDS offset 0, no barrier", out, cyc);
  run<64, true>("images at LDS offset 64 KiB, barrier per step", out, cyc);
  run<64, false>("images at LDS offset 64 KiB, no barrier", out, cyc);
  return 0;
}
