// micro-benchmark: sustained v_mfma_f32_16x16x32_bf16 / 32x32x16 issue rate (no memory traffic)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    const int blocks = 256 * bpc;
    float ms = timeit([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    double fl = (double)blocks * 4 * iters * 16 * 16384.0;
    printf("16x16x32 bf16, 16 accs, %d blocks/CU: %.1f TFLOP/s\n", bpc, fl / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    fl = (double)blocks * 4 * iters * 4 * 16384.0;
    printf("16x16x32 bf16,  4 accs, %d blocks/CU: %.1f TFLOP/s\n", bpc, fl / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters); });
    fl = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("32x32x16 bf16,  4 accs, %d blocks/CU: %.1f TFLOP/s\n", bpc, fl / ms / 1e9);
  }
  return 0;
}
