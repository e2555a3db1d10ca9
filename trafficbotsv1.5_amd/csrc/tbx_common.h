// Shared device helpers (wave64 reductions, error strings). gfx950 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace tbx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// sum over aligned groups of 8 lanes
__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

// 128-d (or 64-d) pe_xy_yaw channel c of pose (x, y, yaw): utils/pose_emb.py:50-55, utils/positional_emb.py:25,53.
// freqs_* are the reference's repeat-interleaved buffers; cos uses the even entries, sin the odd ones.
__device__ __forceinline__ float pose_emb_channel(int c, int pe_dim, float x, float y, float yaw,
                                                  const float* __restrict__ fxy, const float* __restrict__ fyaw) {
  const int nxy = pe_dim >> 3;  // frequencies per cos/sin block of x or y
  const int nyw = pe_dim >> 2;  // yaw harmonics
  if (c < nxy) return cosf(x * fxy[2 * c]);
  c -= nxy;
  if (c < nxy) return sinf(x * fxy[2 * c + 1]);
  c -= nxy;
  if (c < nxy) return cosf(y * fxy[2 * c]);
  c -= nxy;
  if (c < nxy) return sinf(y * fxy[2 * c + 1]);
  c -= nxy;
  if (c < nyw) return cosf(yaw * fyaw[2 * c]);
  c -= nyw;
  return sinf(yaw * fyaw[2 * c + 1]);
}

}  // namespace tbx
