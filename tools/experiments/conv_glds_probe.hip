// Stand-alone probe of the LDS-DMA conv kernel (experiments): per-block clock stamps of the three phases
// (prologue until the first operands have landed, K loop, epilogue) for one shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/conv_glds_probe.hip -o gpurun_out/conv_probe
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../promptttspp_amd/csrc/conv1d_glds.h"

void ptpp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}

template <int FM, int FN, int WR, int WC, int D>
void run(const char* name, ConvP p) {
  constexpr int BM = WR * FM * 16, BN = WC * FN * 16;
  p.nMT = (p.T + BM - 1) / BM;
  p.nNT = (p.Cout + BN - 1) / BN;
  const int xrows = (BM + (p.ks - 1) * p.dil + 7) & ~7;
  const size_t smem = (size_t)(D * BN * 8 + 2 * xrows * 8) * 16;
  const int nblk = p.B * p.nMT * p.nNT;
  unsigned long long* stamps;
  hipMalloc(&stamps, (size_t)nblk * 32);
  hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &stamps, sizeof(stamps));
  auto kern = conv1d_glds_kernel<FM, FN, WR, WC, D, true>;
  auto kern0 = conv1d_glds_kernel<FM, FN, WR, WC, D, false>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t a, e;
  hipEventCreate(&a);
  hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern0, dim3(nblk), dim3(WR * WC * 64), smem, 0, p);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern0, dim3(nblk), dim3(WR * WC * 64), smem, 0, p);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, a, e);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(WR * WC * 64), smem, 0, p);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(WR * WC * 64), smem, 0, p);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float msd;
  hipEventElapsedTime(&msd, a, e);
  printf("   [stamped kernel: %.1f us per launch]\n", msd * 100.0);
  std::vector<unsigned long long> h((size_t)nblk * 4);
  hipMemcpy(h.data(), stamps, (size_t)nblk * 32, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < nblk; ++i) {
    tmin = std::min(tmin, h[4 * i]);
    tmax = std::max(tmax, h[4 * i + 3]);
  }
  std::vector<double> pro, loop, epi, tot, start;
  for (int i = 0; i < nblk; ++i) {
    pro.push_back((h[4 * i + 1] - h[4 * i]) * 0.01);
    loop.push_back((h[4 * i + 2] - h[4 * i + 1]) * 0.01);
    epi.push_back((h[4 * i + 3] - h[4 * i + 2]) * 0.01);
    tot.push_back((h[4 * i + 3] - h[4 * i]) * 0.01);
    start.push_back((h[4 * i] - tmin) * 0.01);
  }
  auto stat = [&](const char* n, std::vector<double> v) {
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    printf("   %-10s mean %7.2f us   p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f\n", n, s / v.size(), v[v.size() / 10], v[v.size() / 2],
           v[v.size() * 9 / 10], v.back());
  };
  const double flop = 2.0 * p.B * p.T * p.Cin * p.Cout * p.ks;
  printf("%s <%d,%d,%d,%d,D%d>: %d blocks, LDS %zu B, %.1f us per launch (%.0f TFLOP/s), stamped launch spans %.1f us\n", name, FM, FN, WR, WC, D,
         nblk, smem, ms * 100.0, flop / (ms * 1e-4) / 1e12, (tmax - tmin) * 0.01);
  stat("prologue", pro);
  stat("K loop", loop);
  stat("epilogue", epi);
  stat("block", tot);
  stat("start at", start);
  hipFree(stamps);
}

template <int FM, int FN, int WR, int WC, int D>
double time_only(ConvP p) {
  constexpr int BM = WR * FM * 16, BN = WC * FN * 16;
  p.nMT = (p.T + BM - 1) / BM;
  p.nNT = (p.Cout + BN - 1) / BN;
  const int xrows = (BM + (p.ks - 1) * p.dil + 7) & ~7;
  const size_t smem = (size_t)(D * BN * 8 + 2 * xrows * 8) * 16;
  if (smem > 160 * 1024) return -1.0;
  const int nblk = p.B * p.nMT * p.nNT;
  auto kern0 = conv1d_glds_kernel<FM, FN, WR, WC, D, false>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t a, e;
  hipEventCreate(&a);
  hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern0, dim3(nblk), dim3(WR * WC * 64), smem, 0, p);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern0, dim3(nblk), dim3(WR * WC * 64), smem, 0, p);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, a, e);
  return ms * 50.0;
}

int main(int argc, char** argv) {
  struct S { const char* name; int B, T, cin, cout, ks, dil; } shapes[] = {
      {"DiffNet dilated 256->512 k3 d2", 52, 576, 256, 512, 3, 2}, {"DiffNet 1x1 256->512", 52, 576, 256, 512, 1, 1},
      {"frame prior 256->256 k17", 52, 576, 256, 256, 17, 1},      {"pitch pred 256->256 k5", 52, 576, 256, 256, 5, 1},
      {"dgrad 512->256 k3 d2", 52, 576, 512, 256, 3, 2},           {"sampler 32x590 256->512 k3", 32, 590, 256, 512, 3, 1},
      {"BigVGAN C=128 k7 d3", 64, 30000, 128, 128, 7, 3},          {"BigVGAN C=256 k11 d5", 64, 6000, 256, 256, 11, 5},
      {"BigVGAN C=256 k3", 64, 6000, 256, 256, 3, 1},              {"BigVGAN C=512 k3 T=1000", 64, 1000, 512, 512, 3, 1}};
  const bool full = argc > 1;
  printf("%-32s %8s %8s %8s %8s %8s %8s %8s %8s %8s\n", "shape (us per launch)", "C 4224.2", "C3 4224.3", "E 2422.2", "E3 2422.3", "E8 2224.2",
         "E8.3", "K 1442.2", "A 4422.2", "G 2244.2");
  for (auto& s : shapes) {
    ConvP p{};
    const int B = s.B, T = s.T;
    void *x, *w, *y, *res;
    float* bias;
    hipMalloc(&x, (size_t)B * T * s.cin * 2);
    hipMalloc(&y, (size_t)B * T * s.cout * 2);
    hipMalloc(&res, (size_t)B * T * s.cout * 2);
    hipMalloc(&w, (size_t)s.cout * s.ks * s.cin * 2);
    hipMalloc(&bias, s.cout * 4);
    // bf16 values in [-1, 1): quote random-data numbers (zero-filled operands run faster)
    {
      std::vector<unsigned short> h((size_t)B * T * std::max(s.cin, s.cout));
      unsigned r = 12345u;
      for (auto& v : h) {
        r = r * 1664525u + 1013904223u;
        v = (unsigned short)(0x3C00u + ((r >> 16) & 0x3FFu)) | (unsigned short)((r >> 31) << 15);
      }
      hipMemcpy(x, h.data(), (size_t)B * T * s.cin * 2, hipMemcpyHostToDevice);
      hipMemcpy(res, h.data(), (size_t)B * T * s.cout * 2, hipMemcpyHostToDevice);
      hipMemcpy(w, h.data(), (size_t)s.cout * s.ks * s.cin * 2, hipMemcpyHostToDevice);
    }
    hipMemset(bias, 0, s.cout * 4);
    p.x = x; p.wp = w; p.bias = bias; p.res = res; p.res2 = nullptr; p.y = y; p.lengths = nullptr;
    p.B = B; p.T = T; p.Cin = s.cin; p.Cout = s.cout; p.ks = s.ks; p.dil = s.dil; p.pad = s.dil * (s.ks - 1) / 2;
    p.ldx = s.cin; p.ldy = s.cout; p.ldr = s.cout; p.ldr2 = 0; p.cinp = s.cin; p.act = 0; p.in_mask = 0; p.out_mask = 0;
    p.out_scale = 1.f; p.res_scale = 1.f; p.drop_thresh16 = 0; p.drop_inv_keep = 1.f; p.drop_seed = 0; p.ws = nullptr; p.nsplit = 1;
    printf("%-32s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", s.name, time_only<4, 2, 2, 4, 2>(p), time_only<4, 2, 2, 4, 3>(p),
           time_only<2, 4, 2, 2, 2>(p), time_only<2, 4, 2, 2, 3>(p), time_only<2, 2, 2, 4, 2>(p), time_only<2, 2, 2, 4, 3>(p),
           time_only<1, 4, 4, 2, 2>(p), time_only<4, 4, 2, 2, 2>(p), time_only<2, 2, 4, 4, 2>(p));
    if (full) {  // per-phase clock stamps of the two product configurations
      run<2, 4, 2, 2, 2>(s.name, p);
      run<4, 4, 2, 2, 2>(s.name, p);
    }
    hipFree(x); hipFree(y); hipFree(res); hipFree(w); hipFree(bias);
  }
  return 0;
}
