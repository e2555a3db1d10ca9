// Keyed elementwise dropout (tbx_keyed_dropout, include/tbx_hip.h): the training path's replacement of F.dropout
// (reference: modules/mlp.py:60-61, transformer_rpe.py:56-60,93-131). The mask is a counter-based hash of (seed, site,
// step, scene row, column), so a time-batched evaluation of T closed-loop steps draws exactly the masks of T per-step
// calls, the backward regenerates the forward's mask, and a captured graph draws fresh masks when the host rewrites the
// device-resident seed. HBM-bound: 4 B read + 4 B written per element, float4 accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

struct DropArgs {
  const float* x;
  float* y;
  int64_t rows;
  int cols, rows_per_scene, time_batch, time0;
  const uint64_t* seed;
  uint32_t site, thresh;
  float scale;
};

__device__ __forceinline__ uint32_t mix(uint32_t x, uint32_t lo, uint32_t hi) {
  x ^= lo;
  x *= 0x9E3779B1u;
  x ^= hi;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// One thread per float4 (VEC = 4, cols % 4 == 0, 16-byte aligned) or per element (VEC = 1).
template <int VEC>
__global__ __launch_bounds__(256) void keyed_dropout_kernel(const DropArgs a) {
  const uint64_t sd = *a.seed;
  const int cv = a.cols / VEC;
  const int64_t total = a.rows * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / cv;
    const int c = (int)(e - row * cv) * VEC;
    const int64_t b = row / a.rows_per_scene;
    const int64_t sc = b / a.time_batch;
    const uint32_t ts = (uint32_t)(a.time0 + (int)(b - sc * a.time_batch));
    const uint32_t krow = (uint32_t)(sc * a.rows_per_scene + (row - b * a.rows_per_scene));
    const uint32_t lo = (uint32_t)sd ^ (a.site * 0x85EBCA6Bu) ^ (ts * 0x27D4EB2Fu);
    const uint32_t hi = (uint32_t)(sd >> 32) + a.site * 0xC2B2AE35u + ts * 0x165667B1u;
    const uint32_t base = krow * (uint32_t)a.cols + (uint32_t)c;
    if constexpr (VEC == 4) {
      const float4 v = *(const float4*)(a.x + row * a.cols + c);
      float4 o;
      o.x = mix(base + 0u, lo, hi) >= a.thresh ? v.x * a.scale : 0.f;
      o.y = mix(base + 1u, lo, hi) >= a.thresh ? v.y * a.scale : 0.f;
      o.z = mix(base + 2u, lo, hi) >= a.thresh ? v.z * a.scale : 0.f;
      o.w = mix(base + 3u, lo, hi) >= a.thresh ? v.w * a.scale : 0.f;
      *(float4*)(a.y + row * a.cols + c) = o;
    } else {
      a.y[row * a.cols + c] = mix(base, lo, hi) >= a.thresh ? a.x[row * a.cols + c] * a.scale : 0.f;
    }
  }
}

}  // namespace

extern "C" int tbx_keyed_dropout(const float* x, float* y, int64_t rows, int cols, int rows_per_scene, float p_drop,
                                 const uint64_t* drop_seed, uint32_t site, int time_batch, int time0, void* stream) {
  if (!x || !y || !drop_seed || rows < 0 || cols <= 0 || rows_per_scene <= 0 || time_batch < 1 || time0 < 0) return TBX_ERR_ARG;
  if (p_drop <= 0.f || p_drop >= 1.f) return TBX_ERR_ARG;
  if (rows % rows_per_scene) return TBX_ERR_ARG;
  if (rows == 0) return TBX_OK;
  DropArgs a;
  a.x = x, a.y = y, a.rows = rows, a.cols = cols, a.rows_per_scene = rows_per_scene, a.time_batch = time_batch, a.time0 = time0;
  a.seed = drop_seed, a.site = site;
  const double th = (double)p_drop * 4294967296.0;
  a.thresh = th < 1.0 ? 1u : (uint32_t)th;
  a.scale = 1.0f / (1.0f - p_drop);
  const bool vec = (cols % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
  const int64_t total = rows * (vec ? cols / 4 : cols);
  const int64_t want = (total + 255) / 256;
  const int blocks = (int)(want < 16384 ? want : 16384);
  hipStream_t hs = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL(keyed_dropout_kernel<4>, dim3(blocks), dim3(256), 0, hs, a);
  else
    hipLaunchKernelGGL(keyed_dropout_kernel<1>, dim3(blocks), dim3(256), 0, hs, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
