// Training-side glue of a PointNet layer over very many groups (the time-batched windows: 10^5 groups of 11 rows), forward and
// backward (include/tbx_hip.h: tbx_pointnet_tail_fwd / _bwd, tbx_masked_maxpool_fwd / _bwd). Reference: polyline_encoder.py:49-61
// (per layer: Linear -> ReLU -> Dropout, then [h | max over the group's valid rows], invalid rows zeroed) and pooling.py:18-19,38
// (the closing masked max), differentiated by autograd there: masked_fill, amax, expand, cat, masked_fill and their backward
// kernels each stream the [groups, W, 64..128] tensors through HBM once or twice. Here a wavefront owns a group (lanes =
// channels), keeps its W rows in registers and reads / writes every tensor once. HBM-bound by design.
// The maximum's gradient is split evenly among tied rows, as aten's amax backward does.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

constexpr int PN_C = 64;      // channels of a layer's Linear (one per lane); the layer's output is 2 * PN_C wide
constexpr int PN_MAXW = 32;   // rows per group kept in registers: the kernels come in two sizes, <MAXW = 16> (the windows of 11) and <32> (the map's polylines of 20 nodes, the posterior's windows of 19)
constexpr int PN_WAVES = 4;

// tbx_keyed_dropout's hash (csrc/dropout.hip)
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

struct TailArgs {
  const float* z;        // [G, W, 64]
  const uint8_t* invalid;  // [G, W]
  float* out;            // [G, W, 128]
  int64_t G;
  int W;
  const uint64_t* seed;  // NULL: no dropout
  uint32_t site, thresh;
  float scale;
  int rows_per_scene, time_batch, time0;
};

// bit w of the result: row w of group g is invalid (wave-uniform)
__device__ __forceinline__ uint32_t invalid_bits(const uint8_t* invalid, int64_t g, int W, int lane) {
  const bool inv = lane < W ? invalid[g * W + lane] != 0 : true;
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__ballot(inv));
}

template <int MAXW>
__global__ __launch_bounds__(PN_WAVES * 64) void pointnet_tail_fwd_kernel(const TailArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * PN_WAVES + (threadIdx.x >> 6);
  if (g >= a.G) return;
  const int W = a.W;
  const uint32_t inv = invalid_bits(a.invalid, g, W, lane);
  const uint64_t sd = a.seed != nullptr ? *a.seed : 0;
  float h[MAXW];
  float m = -INFINITY;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W) {
      const int64_t row = g * W + w;
      float v = fmaxf(a.z[row * PN_C + lane], 0.f);
      if (a.seed != nullptr) {  // tbx_keyed_dropout on the [G * W, 64] view
        const int64_t b = row / a.rows_per_scene;
        const int64_t sc = b / a.time_batch;
        const uint32_t ts = (uint32_t)(a.time0 + (int)(b - sc * a.time_batch));
        const uint32_t krow = (uint32_t)(sc * a.rows_per_scene + (row - b * a.rows_per_scene));
        const uint32_t lo = (uint32_t)sd ^ (a.site * 0x85EBCA6Bu) ^ (ts * 0x27D4EB2Fu);
        const uint32_t hi = (uint32_t)(sd >> 32) + a.site * 0xC2B2AE35u + ts * 0x165667B1u;
        v = mix(krow * (uint32_t)PN_C + (uint32_t)lane, lo, hi) >= a.thresh ? v * a.scale : 0.f;
      }
      h[w] = v;
      if (!((inv >> w) & 1)) m = fmaxf(m, v);
    }
  }
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W) {
      const bool bad = (inv >> w) & 1;
      float* o = a.out + (g * W + w) * (2 * PN_C);
      o[lane] = bad ? 0.f : h[w];
      o[PN_C + lane] = bad ? 0.f : m;
    }
  }
}

struct TailBwdArgs {
  const float* dout;     // [G, W, 128]
  const float* out;      // [G, W, 128] of the forward
  const uint8_t* invalid;
  float* dz;             // [G, W, 64]
  int64_t G;
  int W;
  float scale;           // 1 / (1 - p), or 1 without dropout
};

template <int MAXW>
__global__ __launch_bounds__(PN_WAVES * 64) void pointnet_tail_bwd_kernel(const TailBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * PN_WAVES + (threadIdx.x >> 6);
  if (g >= a.G) return;
  const int W = a.W;
  const uint32_t inv = invalid_bits(a.invalid, g, W, lane);
  float h[MAXW], d[MAXW];
  float m = -INFINITY, gm = 0.f;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W) {
      const int64_t row = g * W + w;
      h[w] = a.out[row * (2 * PN_C) + lane];
      d[w] = a.dout[row * (2 * PN_C) + lane];
      const float dm = a.dout[row * (2 * PN_C) + PN_C + lane];
      if (!((inv >> w) & 1)) {
        m = fmaxf(m, h[w]);
        gm += dm;
      }
    }
  }
  int ties = 0;
#pragma unroll
  for (int w = 0; w < MAXW; ++w)
    if (w < W && !((inv >> w) & 1) && h[w] == m) ++ties;
  const float share = ties > 0 ? gm / (float)ties : 0.f;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W) {
      float v = 0.f;
      if (!((inv >> w) & 1) && h[w] > 0.f) v = (d[w] + (h[w] == m ? share : 0.f)) * a.scale;
      a.dz[(g * W + w) * PN_C + lane] = v;
    }
  }
}

struct PoolArgs {
  const float* x;        // [G, W, 128]
  const uint8_t* invalid;
  float* y;              // fwd: out [G, 128]; bwd: dy in
  float* dx;             // bwd only
  int64_t G;
  int W;
};

template <int MAXW>
__global__ __launch_bounds__(PN_WAVES * 64) void masked_maxpool_fwd_kernel(const PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * PN_WAVES + (threadIdx.x >> 6);
  if (g >= a.G) return;
  const int W = a.W;
  const uint32_t inv = invalid_bits(a.invalid, g, W, lane);
  float2 m = make_float2(-INFINITY, -INFINITY);
  bool any = false;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W && !((inv >> w) & 1)) {
      const float2 v = *(const float2*)(a.x + (g * W + w) * (2 * PN_C) + 2 * lane);
      m.x = fmaxf(m.x, v.x);
      m.y = fmaxf(m.y, v.y);
      any = true;
    }
  }
  if (!any) m = make_float2(0.f, 0.f);
  *(float2*)(a.y + g * (2 * PN_C) + 2 * lane) = m;
}

template <int MAXW>
__global__ __launch_bounds__(PN_WAVES * 64) void masked_maxpool_bwd_kernel(const PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * PN_WAVES + (threadIdx.x >> 6);
  if (g >= a.G) return;
  const int W = a.W;
  const uint32_t inv = invalid_bits(a.invalid, g, W, lane);
  float2 v[MAXW];
  float2 m = make_float2(-INFINITY, -INFINITY);
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W) {
      v[w] = *(const float2*)(a.x + (g * W + w) * (2 * PN_C) + 2 * lane);
      if (!((inv >> w) & 1)) {
        m.x = fmaxf(m.x, v[w].x);
        m.y = fmaxf(m.y, v[w].y);
      }
    }
  }
  int tx = 0, ty = 0;
#pragma unroll
  for (int w = 0; w < MAXW; ++w)
    if (w < W && !((inv >> w) & 1)) {
      tx += v[w].x == m.x;
      ty += v[w].y == m.y;
    }
  const float2 dy = *(const float2*)(a.y + g * (2 * PN_C) + 2 * lane);
  const float sx = tx > 0 ? dy.x / (float)tx : 0.f, sy = ty > 0 ? dy.y / (float)ty : 0.f;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    if (w < W) {
      float2 o = make_float2(0.f, 0.f);
      if (!((inv >> w) & 1)) {
        o.x = v[w].x == m.x ? sx : 0.f;
        o.y = v[w].y == m.y ? sy : 0.f;
      }
      *(float2*)(a.dx + (g * W + w) * (2 * PN_C) + 2 * lane) = o;
    }
  }
}

inline bool shape_ok(int64_t G, int W, int C) { return G > 0 && W > 0 && W <= PN_MAXW && C == PN_C; }
inline dim3 grid_of(int64_t G) { return dim3((unsigned)((G + PN_WAVES - 1) / PN_WAVES)); }

}  // namespace

extern "C" int tbx_pointnet_tail_fwd(const float* z, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols, float p_drop,
                                     const uint64_t* drop_seed, uint32_t site, int rows_per_scene, int time_batch, int time0, float* out,
                                     void* stream) {
  if (!z || !invalid || !out) return TBX_ERR_ARG;
  if (!shape_ok(n_groups, group_rows, cols)) return TBX_ERR_UNSUPPORTED;
  TailArgs a{z, invalid, out, n_groups, group_rows, nullptr, site, 0u, 1.0f, 1, 1, 0};
  if (p_drop > 0.f) {
    if (!drop_seed || p_drop >= 1.f || rows_per_scene <= 0 || time_batch < 1 || time0 < 0) return TBX_ERR_ARG;
    if ((n_groups * group_rows) % rows_per_scene) return TBX_ERR_ARG;
    const double th = (double)p_drop * 4294967296.0;
    a.seed = drop_seed, a.thresh = th < 1.0 ? 1u : (uint32_t)th, a.scale = 1.0f / (1.0f - p_drop);
    a.rows_per_scene = rows_per_scene, a.time_batch = time_batch, a.time0 = time0;
  }
  if (group_rows <= 16) hipLaunchKernelGGL(pointnet_tail_fwd_kernel<16>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(pointnet_tail_fwd_kernel<32>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_pointnet_tail_bwd(const float* dout, const float* out, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols,
                                     float p_drop, float* dz, void* stream) {
  if (!dout || !out || !invalid || !dz || p_drop < 0.f || p_drop >= 1.f) return TBX_ERR_ARG;
  if (!shape_ok(n_groups, group_rows, cols)) return TBX_ERR_UNSUPPORTED;
  TailBwdArgs a{dout, out, invalid, dz, n_groups, group_rows, p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f};
  if (group_rows <= 16) hipLaunchKernelGGL(pointnet_tail_bwd_kernel<16>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(pointnet_tail_bwd_kernel<32>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_masked_maxpool_fwd(const float* x, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols, float* y,
                                      void* stream) {
  if (!x || !invalid || !y) return TBX_ERR_ARG;
  if (!shape_ok(n_groups, group_rows, cols / 2) || (cols & 1)) return TBX_ERR_UNSUPPORTED;
  PoolArgs a{x, invalid, y, nullptr, n_groups, group_rows};
  if (group_rows <= 16) hipLaunchKernelGGL(masked_maxpool_fwd_kernel<16>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(masked_maxpool_fwd_kernel<32>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_masked_maxpool_bwd(const float* dy, const float* x, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols,
                                      float* dx, void* stream) {
  if (!dy || !x || !invalid || !dx) return TBX_ERR_ARG;
  if (!shape_ok(n_groups, group_rows, cols / 2) || (cols & 1)) return TBX_ERR_UNSUPPORTED;
  PoolArgs a{x, invalid, (float*)dy, dx, n_groups, group_rows};
  if (group_rows <= 16) hipLaunchKernelGGL(masked_maxpool_bwd_kernel<16>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(masked_maxpool_bwd_kernel<32>, grid_of(n_groups), dim3(PN_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
