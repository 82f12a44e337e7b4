// Host cost of one kernel launch through the HIP runtime as a function of the by-value argument block size
// (empty kernel, one stream): the conv kernels take a ~250-byte block.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blk { char b[N]; };
__global__ void k0() {}
template <int N> __global__ void kn(Blk<N> b, int* out) { if (out && b.b[0] == 77) *out = 1; }
template <int N> void run(hipStream_t st) {
  Blk<N> blk{};
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(kn<N>, dim3(1), dim3(64), 0, st, blk, nullptr);
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  auto t2 = std::chrono::steady_clock::now();
  printf("%4d-byte args: host %.2f us per launch, drained after %.2f us per launch\n", N,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / 20000, std::chrono::duration<double, std::micro>(t2 - t0).count() / 20000);
}
int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, st);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    printf("   no args: host %.2f us per launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 20000);
    run<16>(st); run<64>(st); run<120>(st); run<128>(st); run<136>(st); run<200>(st); run<256>(st); run<320>(st);
  }
  return 0;
}
