// v_mfma_f32_4x4x4_16B_bf16 probe (gfx950): operand / result layout and the A-broadcast (cbsz, abid) the attention's stage 2 relies on.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma4x4_probe.hip -o /tmp/mfma4 && /tmp/mfma4
// Claim under test: 16 blocks b = lane >> 2; A: lane (b, i = lane & 3) holds A_b[i][k = 0..3]; B: lane (b, j = lane & 3) holds
// B_b[k = 0..3][j]; D: lane (b, j) holds D_b[i = 0..3][j] in its 4 result registers; with cbsz = 4, abid = q every block uses block
// q's A registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D, float* Dbc) {
  // A[16][4][4] (block, i, k), B[16][4][4] (block, k, j)
  const int l = threadIdx.x, b = l >> 2, r = l & 3;
  bf16x4 a, bb;
  for (int k = 0; k < 4; ++k) a[k] = (__bf16)A[(b * 4 + r) * 4 + k], bb[k] = (__bf16)B[(b * 4 + k) * 4 + r];
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, bb), z, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + r] = d[i];
  f32x4 e = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, bb), z, 4, 8, 0);  // every block uses block 8's A
  for (int i = 0; i < 4; ++i) Dbc[(b * 4 + i) * 4 + r] = e[i];
}

int main() {
  float hA[256], hB[256], hD[256], hE[256];
  for (int i = 0; i < 256; ++i) hA[i] = (float)((i * 7) % 13 - 6), hB[i] = (float)((i * 5) % 11 - 5);
  float *dA, *dB, *dD, *dE;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024); hipMalloc(&dE, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dE);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hE, dE, 1024, hipMemcpyDeviceToHost);
  int bad = 0, badbc = 0;
  for (int b = 0; b < 16; ++b)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        float s = 0.f, t = 0.f;
        for (int k = 0; k < 4; ++k) s += hA[(b * 4 + i) * 4 + k] * hB[(b * 4 + k) * 4 + j], t += hA[(8 * 4 + i) * 4 + k] * hB[(b * 4 + k) * 4 + j];
        bad += hD[(b * 4 + i) * 4 + j] != s;
        badbc += hE[(b * 4 + i) * 4 + j] != t;
      }
  printf("plain layout mismatches: %d of 256; broadcast (cbsz 4, abid 8) mismatches: %d of 256\n", bad, badbc);
  return 0;
}
