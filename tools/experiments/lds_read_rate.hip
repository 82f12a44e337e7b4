// Micro-benchmark (round 5): what does the LDS -> register path deliver on gfx950, as a function of
//   * ds_read_b128 wave-instructions in flight per s_waitcnt lgkmcnt(0)      R  in {4, 8, 16, 32}
//   * s_barrier every S stages of 8 reads                                    S  in {0 (never), 1, 2, 4}
//   * waves per CU                                                           W  in {4, 8, 16}
//   * address pattern: LIN (lane l reads 16 B at 16 l of a 1 KiB fragment), FRAG (the MFMA fragment read of the
//     conv kernels: 16 rows x 64-byte rows, lane (lr = l & 15, lg = l >> 4) reads chunk lg ^ swz(lr) of row lr)
//   * MFMAs: 0, or 2 v_mfma_f32_16x16x32_bf16 per read (the 64 x 64 wave tile's ratio: 8 fragment reads per 16 MFMAs),
//     issued behind the reads of the NEXT group (software pipeline) or behind their own wait (no pipeline)
// One workgroup per CU (96 KiB of dynamic LDS keeps a second one out); every wave walks a 4 x 16 KiB ring of stage
// images.  Output: bytes / clk / CU from s_memtime inside the kernel and from the event time.
//   build: hipcc -O3 --offload-arch=gfx950 lds_read_rate.hip -o lds_read_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int R>
struct Regs {
  u32x4 r[R];
};

// issue R reads: read i goes to fragment (i & 7) of the stage at `base` (1 KiB apart), stage advance every 8 reads
template <int R>
__device__ __forceinline__ void issue(Regs<R>& g, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int st = (i >> 3) & 3, f = i & 7;
    const uint32_t a = st == 0 ? a0 : st == 1 ? a1 : st == 2 ? a2 : a3;
    switch (f) {
      case 0: DSR(g.r[i], a, 0); break;
      case 1: DSR(g.r[i], a, 1024); break;
      case 2: DSR(g.r[i], a, 2048); break;
      case 3: DSR(g.r[i], a, 3072); break;
      case 4: DSR(g.r[i], a, 8192); break;
      case 5: DSR(g.r[i], a, 9216); break;
      case 6: DSR(g.r[i], a, 10240); break;
      default: DSR(g.r[i], a, 11264); break;
    }
  }
}

template <int R, int MF>
__device__ __forceinline__ void consume(const Regs<R>& g, f32x4 (&acc)[16], uint32_t& x) {
  if constexpr (MF == 0) {
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(g.r[i][0]));
  } else {
    // 8 reads = 4 "A" + 4 "B" fragments -> 16 MFMAs
#pragma unroll
    for (int q = 0; q < R / 8; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                       : "+v"(acc[i * 4 + j])
                       : "v"(g.r[q * 8 + i]), "v"(g.r[q * 8 + 4 + j]));
  }
}

// READS reads per iteration in total (= 32), in groups of R; barrier every S stages (S = 0: never)
template <int R, int S, int MF, bool PIPE, bool FRAG, int NT>
__global__ __launch_bounds__(NT) void lds_rate(uint32_t* out, uint64_t* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t off;
  if (FRAG) {
    const int lr = lane & 15, lg = lane >> 4;
    off = lr * 64 + ((lg ^ ((lr >> 1) & 3)) << 4);
  } else {
    off = lane * 16;
  }
  // waves of a block read DIFFERENT fragments of the same stage (as the n / m wave split of a tile does): the
  // half-stage each wave starts from differs by 4 KiB
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + off + (wave & 1) * 4096;
  const uint32_t a0 = base, a1 = base + 16384, a2 = base + 32768, a3 = base + 49152;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  uint32_t x = 0;
  constexpr int G = 32 / R;  // groups per iteration
  Regs<R> ga, gb;
  const uint64_t t0 = __builtin_readcyclecounter();
  if constexpr (PIPE) {
    issue<R>(ga, a0, a1, a2, a3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < G; g += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue<R>(gb, a0, a1, a2, a3);
        consume<R, MF>(ga, acc, x);
        if (S && ((g * R / 8) % S == 0)) asm volatile("s_barrier" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue<R>(ga, a0, a1, a2, a3);
        consume<R, MF>(gb, acc, x);
        if (S && (((g + 1) * R / 8) % S == 0)) asm volatile("s_barrier" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        issue<R>(ga, a0, a1, a2, a3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        consume<R, MF>(ga, acc, x);
        if (S && ((g * R / 8) % S == 0)) asm volatile("s_barrier" ::: "memory");
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (s == 123.456f || x == 0x12345u) out[blockIdx.x * blockDim.x + tid] = x;
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int R, int S, int MF, bool PIPE, bool FRAG, int NT>
void run(const char* name, uint32_t* out, uint64_t* cyc) {
  constexpr int waves = NT / 64;
  const int iters = 4000;
  auto kern = lds_rate<R, S, MF, PIPE, FRAG, NT>;
  const int smem = 96 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), smem, 0, out, cyc, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), smem, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(256 * 16);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  const double bytes_cu = (double)waves * iters * 32 * 1024.0;
  const double mfma_cyc = MF ? (double)(waves / 4.0) * iters * 64 * 16.0 : 0.0;  // per SIMD, 16 cyc per 16x16x32
  printf("%-34s W=%2d  %7.1f B/clk/CU (s_memtime)  %7.1f GB/s/CU (event)  clk %.2f GHz", name, waves, bytes_cu / (double)mx,
         bytes_cu / (ms * 1e6), (double)mx / (ms * 1e6));
  if (MF) printf("  MFMA busy %.0f %%", 100.0 * mfma_cyc / (double)mx);
  printf("\n");
}

#define FITS16(R, MF, PIPE) ((PIPE ? 2 : 1) * R * 4 + (MF ? 64 : 0) + 16 <= 128)
#define RUNW(R, S, MF, PIPE, FRAG, NAME)          \
  run<R, S, MF, PIPE, FRAG, 256>(NAME, out, cyc); \
  run<R, S, MF, PIPE, FRAG, 512>(NAME, out, cyc); \
  if (FITS16(R, MF, PIPE)) run<FITS16(R, MF, PIPE) ? R : 4, S, FITS16(R, MF, PIPE) ? MF : 0, PIPE, FRAG, 1024>(NAME, out, cyc);

int main() {
  uint32_t* out;
  uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&cyc, 256 * 16 * 8);
  printf("== reads only, LIN pattern, no barrier ==\n");
  RUNW(4, 0, 0, false, false, "R=4  nobar LIN");
  RUNW(8, 0, 0, false, false, "R=8  nobar LIN");
  RUNW(16, 0, 0, false, false, "R=16 nobar LIN");
  RUNW(32, 0, 0, false, false, "R=32 nobar LIN");
  printf("== reads only, FRAG pattern (64-byte rows, swizzled), no barrier ==\n");
  RUNW(4, 0, 0, false, true, "R=4  nobar FRAG");
  RUNW(8, 0, 0, false, true, "R=8  nobar FRAG");
  RUNW(16, 0, 0, false, true, "R=16 nobar FRAG");
  RUNW(32, 0, 0, false, true, "R=32 nobar FRAG");
  printf("== reads only, FRAG, barrier every S stages of 8 reads ==\n");
  RUNW(8, 1, 0, false, true, "R=8  bar/1 FRAG");
  RUNW(16, 2, 0, false, true, "R=16 bar/2 FRAG");
  RUNW(32, 4, 0, false, true, "R=32 bar/4 FRAG");
  RUNW(8, 2, 0, false, true, "R=8  bar/2 FRAG");
  RUNW(8, 4, 0, false, true, "R=8  bar/4 FRAG");
  printf("== reads pipelined one group ahead (reads of g+1 in flight during consume of g), FRAG ==\n");
  RUNW(8, 0, 0, true, true, "R=8  pipe nobar");
  RUNW(16, 0, 0, true, true, "R=16 pipe nobar");
  printf("== with MFMAs: 16 x 16x16x32 bf16 per 8 reads (64 x 64 wave tile), FRAG ==\n");
  RUNW(8, 1, 1, false, true, "R=8  bar/1 mfma nopipe");
  RUNW(8, 1, 1, true, true, "R=8  bar/1 mfma pipe");
  RUNW(16, 2, 1, false, true, "R=16 bar/2 mfma nopipe");
  RUNW(16, 2, 1, true, true, "R=16 bar/2 mfma pipe");
  RUNW(8, 0, 1, true, true, "R=8  nobar mfma pipe");
  RUNW(16, 0, 1, true, true, "R=16 nobar mfma pipe");
  RUNW(8, 4, 1, true, true, "R=8  bar/4 mfma pipe");
  return 0;
}
