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

// Writes the pe_dim-d pe_xy_yaw embedding of (x, y, yaw) to e[0:pe_dim], cooperatively over `nl` lanes (lane id `l`).
// The pe_dim channels are pe_dim/2 (cos, sin) pairs of the SAME argument (x f_i | y f_i | k yaw), so each lane evaluates
// one sincosf per argument and writes two channels: no divergent trig branches. The reference's `freqs` buffers are
// repeat-interleaved ([f0,f0,f1,f1,..], positional_emb.py:13,41); the even entry is used for both channels of a pair.
__device__ __forceinline__ void pose_emb_write(float* __restrict__ e, int pe_dim, float x, float y, float yaw,
                                               const float* __restrict__ fxy, const float* __restrict__ fyaw, int l, int nl) {
  const int nxy = pe_dim >> 3, nyw = pe_dim >> 2;
  for (int a = l; a < 2 * nxy + nyw; a += nl) {
    float arg;
    int cc, sc;
    if (a < nxy) {
      arg = x * fxy[2 * a];
      cc = a;
      sc = a + nxy;
    } else if (a < 2 * nxy) {
      const int i = a - nxy;
      arg = y * fxy[2 * i];
      cc = 2 * nxy + i;
      sc = 3 * nxy + i;
    } else {
      const int i = a - 2 * nxy;
      arg = yaw * fyaw[2 * i];
      cc = 4 * nxy + i;
      sc = 4 * nxy + nyw + i;
    }
    float sn, cs;
    sincosf(arg, &sn, &cs);
    e[cc] = cs;
    e[sc] = sn;
  }
}

}  // namespace tbx
