// GPU probe: semantics of ds_read_b64_tr_b16 (needed for the bf16 wgrad operand transposes).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short v4s;
__global__ void k(short* out, int mode) {
  __shared__ short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (short)i;
  __syncthreads();
  int lane = threadIdx.x;
  int off;  // in elements
  if (mode == 0) off = lane * 4;                                  // lane-linear 8 B each
  else if (mode == 1) off = (lane & 15) * 64 + (lane >> 4) * 4;   // row = lane&15 (stride 64 el), 4 el at col (lane>>4)*4
  else off = ((lane & 15) >> 2) * 64 + (lane & 3) * 4 + (lane >> 4) * 16;  // 4 rows x 16 cols block per 16-lane group
  v4s t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(sm + off));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = t[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
