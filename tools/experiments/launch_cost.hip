// Host cost of one kernel launch through the HIP runtime (empty kernel, stream 0), and of one launch with a
// 200-byte by-value argument block like the conv kernels take.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[200]; };
__global__ void k0() {}
__global__ void k1(Big b, int* out) { if (out && b.b[0] == 77) *out = 1; }
int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  Big big{};
  for (int rep = 0; rep < 3; ++rep) {
    for (int which = 0; which < 2; ++which) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 20000; ++i) {
        if (which == 0) hipLaunchKernelGGL(k0, dim3(1), dim3(64), 0, st);
        else hipLaunchKernelGGL(k1, dim3(1), dim3(64), 0, st, big, nullptr);
      }
      auto t1 = std::chrono::steady_clock::now();
      hipDeviceSynchronize();
      auto t2 = std::chrono::steady_clock::now();
      printf("%s: host %.2f us per launch, drained after %.2f us per launch\n", which ? "200-byte args" : "no args",
             std::chrono::duration<double, std::micro>(t1 - t0).count() / 20000, std::chrono::duration<double, std::micro>(t2 - t0).count() / 20000);
    }
  }
  return 0;
}
