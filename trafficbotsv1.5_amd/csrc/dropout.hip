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

// ---- the elementwise glue of a transformer layer in the time-batched training pass, one pass per tensor ------------------
// tbx_residual_drop_fwd / _bwd:  out = zero_out[row] ? 0 : x + dropout(zero_y[row] ? 0 : y)
//   (transformer_rpe.py:93-131: `x + dropout(attn_out.masked_fill(no_valid_target, 0))`, and the layer's closing
//   `(x + dropout(linear2(.))).masked_fill(invalid, 0)`) instead of masked_fill (clone + fill) / dropout / add [/ masked_fill].
// tbx_relu_drop_fwd / _bwd:      h = dropout(relu(z))   (the FFN's hidden activation, [rows, 512])
// Same keyed mask as tbx_keyed_dropout for the tensor's [rows, cols] view; thresh == 0: no dropout. float4 per thread.
struct GlueArgs {
  const float *x, *y;        // fwd: x, y (relu: y = z, x unused) | bwd: x = dout / dh, y = h (relu only)
  const uint8_t *zero_y, *zero_out;
  float *out, *out2;         // fwd: out | bwd: out = dy / dz, out2 = dx (residual with zero_out) or NULL
  int64_t rows;
  int cols, rows_per_scene, time_batch, time0;
  const uint64_t* seed;
  uint32_t site, thresh;
  float scale;
};

struct Keep4 {
  bool k[4];
};
__device__ __forceinline__ Keep4 keep4(const GlueArgs& a, uint64_t sd, int64_t row, int c) {
  Keep4 r;
  if (a.thresh == 0u) {
    r.k[0] = r.k[1] = r.k[2] = r.k[3] = true;
    return r;
  }
  const int64_t b = row / a.rows_per_scene;
  const int64_t sc = b / a.time_batch;
  const uint32_t ts = (uint32_t)(a.time0 + (int)(b - sc * a.time_batch));
  const uint32_t krow = (uint32_t)(sc * a.rows_per_scene + (row - b * a.rows_per_scene));
  const uint32_t lo = (uint32_t)sd ^ (a.site * 0x85EBCA6Bu) ^ (ts * 0x27D4EB2Fu);
  const uint32_t hi = (uint32_t)(sd >> 32) + a.site * 0xC2B2AE35u + ts * 0x165667B1u;
  const uint32_t base = krow * (uint32_t)a.cols + (uint32_t)c;
#pragma unroll
  for (int q = 0; q < 4; ++q) r.k[q] = mix(base + (uint32_t)q, lo, hi) >= a.thresh;
  return r;
}

// MODE 0: residual fwd, 1: residual bwd, 2: relu-drop fwd, 3: relu-drop bwd
template <int MODE>
__global__ __launch_bounds__(256) void glue_kernel(const GlueArgs a) {
  const uint64_t sd = a.thresh != 0u ? *a.seed : 0;
  const int cv = a.cols / 4;
  const int64_t total = a.rows * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / cv;
    const int c = (int)(e - row * cv) * 4;
    const int64_t at = row * a.cols + c;
    const Keep4 kp = keep4(a, sd, row, c);
    float o[4], o2[4];
    if constexpr (MODE == 0) {
      const bool zo = a.zero_out != nullptr && a.zero_out[row] != 0, zy = a.zero_y != nullptr && a.zero_y[row] != 0;
      const float4 xv = *(const float4*)(a.x + at), yv = *(const float4*)(a.y + at);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = zy ? 0.f : ys[q];
        o[q] = zo ? 0.f : xs[q] + (kp.k[q] ? v * a.scale : 0.f);
      }
    } else if constexpr (MODE == 1) {
      const bool zo = a.zero_out != nullptr && a.zero_out[row] != 0, zy = a.zero_y != nullptr && a.zero_y[row] != 0;
      const float4 dv = *(const float4*)(a.x + at);
      const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[q] = (zo || zy || !kp.k[q]) ? 0.f : ds[q] * a.scale;
        o2[q] = zo ? 0.f : ds[q];
      }
      if (a.out2 != nullptr) *(float4*)(a.out2 + at) = make_float4(o2[0], o2[1], o2[2], o2[3]);
    } else if constexpr (MODE == 2) {
      const float4 zv = *(const float4*)(a.y + at);
      const float zs[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = kp.k[q] ? fmaxf(zs[q], 0.f) * a.scale : 0.f;
    } else {
      const float4 dv = *(const float4*)(a.x + at), hv = *(const float4*)(a.y + at);
      const float ds[4] = {dv.x, dv.y, dv.z, dv.w}, hs[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = hs[q] > 0.f ? ds[q] * a.scale : 0.f;  // h > 0 <=> z > 0 and kept
    }
    *(float4*)(a.out + at) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

int glue_launch(int mode, GlueArgs a, float p_drop, const uint64_t* drop_seed, uint32_t site, int rows_per_scene, int time_batch,
                int time0, void* stream) {
  if (a.rows < 0 || a.cols <= 0 || (a.cols & 3)) return TBX_ERR_UNSUPPORTED;
  if (a.rows == 0) return TBX_OK;
  if (p_drop < 0.f || p_drop >= 1.f) return TBX_ERR_ARG;
  a.seed = drop_seed, a.site = site, a.thresh = 0u, a.scale = 1.0f;
  a.rows_per_scene = 1, a.time_batch = 1, a.time0 = 0;
  if (p_drop > 0.f) {
    if (!drop_seed || rows_per_scene <= 0 || time_batch < 1 || time0 < 0 || a.rows % rows_per_scene) return TBX_ERR_ARG;
    const double th = (double)p_drop * 4294967296.0;
    a.thresh = th < 1.0 ? 1u : (uint32_t)th;
    a.scale = 1.0f / (1.0f - p_drop);
    a.rows_per_scene = rows_per_scene, a.time_batch = time_batch, a.time0 = time0;
  }
  const int64_t total = a.rows * (a.cols / 4);
  const int64_t want = (total + 255) / 256;
  const dim3 grid((unsigned)(want < 16384 ? want : 16384));
  hipStream_t hs = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(glue_kernel<0>, grid, dim3(256), 0, hs, a); break;
    case 1: hipLaunchKernelGGL(glue_kernel<1>, grid, dim3(256), 0, hs, a); break;
    case 2: hipLaunchKernelGGL(glue_kernel<2>, grid, dim3(256), 0, hs, a); break;
    default: hipLaunchKernelGGL(glue_kernel<3>, grid, dim3(256), 0, hs, a); break;
  }
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int tbx_residual_drop_fwd(const float* x, const float* y, const uint8_t* zero_y, const uint8_t* zero_out, int64_t rows, int cols,
                                     float p_drop, const uint64_t* drop_seed, uint32_t site, int rows_per_scene, int time_batch, int time0,
                                     float* out, void* stream) {
  if (!x || !y || !out) return TBX_ERR_ARG;
  if (!aligned16(x) || !aligned16(y) || !aligned16(out)) return TBX_ERR_ALIGN;
  GlueArgs a{x, y, zero_y, zero_out, out, nullptr, rows, cols};
  return glue_launch(0, a, p_drop, drop_seed, site, rows_per_scene, time_batch, time0, stream);
}

extern "C" int tbx_residual_drop_bwd(const float* dout, const uint8_t* zero_y, const uint8_t* zero_out, int64_t rows, int cols, float p_drop,
                                     const uint64_t* drop_seed, uint32_t site, int rows_per_scene, int time_batch, int time0, float* dy,
                                     float* dx, void* stream) {
  if (!dout || !dy) return TBX_ERR_ARG;
  if (!aligned16(dout) || !aligned16(dy) || !aligned16(dx)) return TBX_ERR_ALIGN;
  GlueArgs a{dout, nullptr, zero_y, zero_out, dy, dx, rows, cols};
  return glue_launch(1, a, p_drop, drop_seed, site, rows_per_scene, time_batch, time0, stream);
}

extern "C" int tbx_relu_drop_fwd(const float* z, int64_t rows, int cols, float p_drop, const uint64_t* drop_seed, uint32_t site,
                                 int rows_per_scene, int time_batch, int time0, float* h, void* stream) {
  if (!z || !h) return TBX_ERR_ARG;
  if (!aligned16(z) || !aligned16(h)) return TBX_ERR_ALIGN;
  GlueArgs a{nullptr, z, nullptr, nullptr, h, nullptr, rows, cols};
  return glue_launch(2, a, p_drop, drop_seed, site, rows_per_scene, time_batch, time0, stream);
}

extern "C" int tbx_relu_drop_bwd(const float* dh, const float* h, int64_t rows, int cols, float p_drop, float* dz, void* stream) {
  if (!dh || !h || !dz) return TBX_ERR_ARG;
  if (!aligned16(dh) || !aligned16(h) || !aligned16(dz)) return TBX_ERR_ALIGN;
  if (cols <= 0 || (cols & 3) || rows < 0 || p_drop < 0.f || p_drop >= 1.f) return TBX_ERR_ARG;
  if (rows == 0) return TBX_OK;
  GlueArgs a{dh, h, nullptr, nullptr, dz, nullptr, rows, cols};
  // the mask is read off h: only the scale of the dropout is needed here
  a.seed = nullptr, a.site = 0, a.thresh = 0u, a.scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  a.rows_per_scene = 1, a.time_batch = 1, a.time0 = 0;
  const int64_t total = rows * (cols / 4);
  const int64_t want = (total + 255) / 256;
  hipLaunchKernelGGL(glue_kernel<3>, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

// h[n, a, m, :] = [relu](h[n, a, m, :] + (pa[n, a, :] + pm[n, m, :])) in place: the per-agent and per-polyline terms of NaviPredictor's first
// Linear broadcast onto the per-pair term (train_ops.NaviPairFirstLayer) - one pass over the [n, A, M, cols] tensor instead of three
// torch launches per scene (add, add_, copy_) and a relu over the whole of it.
struct PairBiasArgs {
  float* h;
  const float* pa;
  const float* pm;
  int64_t total;  // n * A * M * cols / 4
  int A, M, c4, relu;
};
__global__ __launch_bounds__(256) void pair_bias_kernel(const PairBiasArgs a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.total; i += stride) {
    const int64_t row = i / a.c4;
    const int c = (int)(i - row * a.c4);
    const int64_t na = row / a.M;
    const int m = (int)(row - na * a.M);
    const int64_t nn = na / a.A;
    const float4 u = *(const float4*)(a.pa + (na * a.c4 + c) * 4);
    const float4 w = *(const float4*)(a.pm + ((nn * a.M + m) * a.c4 + c) * 4);
    float4 v = *(const float4*)(a.h + i * 4);
    v.x += u.x + w.x, v.y += u.y + w.y, v.z += u.z + w.z, v.w += u.w + w.w;
    if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
    *(float4*)(a.h + i * 4) = v;
  }
}

extern "C" int tbx_pair_bias_relu(float* h, const float* pa, const float* pm, int64_t n_batch, int n_a, int n_m, int cols, int relu, void* stream) {
  if (!h || !pa || !pm || n_batch < 0 || n_a <= 0 || n_m <= 0 || cols <= 0 || (cols & 3)) return TBX_ERR_ARG;
  if (!aligned16(h) || !aligned16(pa) || !aligned16(pm)) return TBX_ERR_ALIGN;
  if (n_batch == 0) return TBX_OK;
  PairBiasArgs a{h, pa, pm, n_batch * n_a * n_m * (cols / 4), n_a, n_m, cols / 4, relu};
  const int64_t want = (a.total + 255) / 256;
  hipLaunchKernelGGL(pair_bias_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

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
