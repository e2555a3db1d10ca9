#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../trafficbotsv1.5_amd/csrc/tbx_common.h"
__global__ void k(float* o, const float* in) {
  float v = in[threadIdx.x];
  o[threadIdx.x] = tbx::wave_sum(v);
  float b = v;
  for (int off = 1; off < 64; off <<= 1) b += __shfl_xor(b, off, 64);
  o[64 + threadIdx.x] = b;
  o[128 + threadIdx.x] = tbx::slot_sum(v);
  b = v;
  for (int off = 8; off < 64; off <<= 1) b += __shfl_xor(b, off, 64);
  o[192 + threadIdx.x] = b;
  float x, y; tbx::swap16(v, &x, &y); o[256 + threadIdx.x] = x; o[320 + threadIdx.x] = y;
  tbx::swap32(v, &x, &y); o[384 + threadIdx.x] = x; o[448 + threadIdx.x] = y;
  o[512 + threadIdx.x] = tbx::wave_max(v);
  o[576 + threadIdx.x] = tbx::group8_sum(v);
}
int main() {
  float h[64], *d, *o, r[640];
  for (int i = 0; i < 64; ++i) h[i] = i + 0.37f * (i % 7);
  hipMalloc(&d, 256); hipMalloc(&o, 640 * 4);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, d);
  hipMemcpy(r, o, 640 * 4, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 64; ++i) { bad1 += r[i] != r[64 + i]; bad2 += r[128 + i] != r[192 + i]; }
  printf("wave_sum mismatches %d slot_sum mismatches %d\n", bad1, bad2);
  printf("wave_sum %f ref %f max %f g8 %f %f\n", r[0], r[64], r[512], r[576], r[576+8]);
  printf("swap16 a:"); for (int i = 0; i < 64; i += 4) printf(" %g", r[256 + i]); printf("\nswap16 b:"); for (int i = 0; i < 64; i += 4) printf(" %g", r[320 + i]);
  printf("\nswap32 a:"); for (int i = 0; i < 64; i += 4) printf(" %g", r[384 + i]); printf("\nswap32 b:"); for (int i = 0; i < 64; i += 4) printf(" %g", r[448 + i]);
  printf("\n");
  return 0;
}
