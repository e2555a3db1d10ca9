// tbx_layernorm_fwd / tbx_layernorm_bwd (include/tbx_hip.h): the 128-wide LayerNorms of the time-batched training pass (autograd of
// F.layer_norm at modules/transformer_rpe.py:207-245, attention_rpe.py:92-98 norm_tgt, in the reference's training_step).
// HBM-bound: x and dy read once, dx written once (1.5 KB per row); a wavefront per row (float2 per lane = one 512-byte row per
// load), four rows in flight per wavefront, dgamma / dbeta accumulated per lane across the wavefront's rows and combined in a
// fixed order (per workgroup, then over the workgroups' partials by a second small launch): deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

constexpr int LN_D = 128;
constexpr int LN_WAVES = 16;     // wavefronts per workgroup
constexpr int LN_MAX_WG = 512;   // 2 workgroups per CU: 8 wavefronts per SIMD
constexpr int LN_UNROLL = 4;

struct LnBwdArgs {
  const float *x, *dy, *gamma, *mean, *rstd;
  float *dx, *part;
  int64_t rows;
  const float* add;  // ADD: dx = add + (the LayerNorm's input gradient) - the gradient of the residual branch that forks off x
};

template <bool ADD>
__global__ __launch_bounds__(LN_WAVES * 64) void ln_bwd_kernel(const LnBwdArgs a) {
  __shared__ float red[LN_WAVES][2 * LN_D];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t w0 = (int64_t)blockIdx.x * LN_WAVES + wave, tw = (int64_t)gridDim.x * LN_WAVES;
  const float2 gm = *(const float2*)(a.gamma + 2 * lane);
  float2 dg = make_float2(0.f, 0.f), db = make_float2(0.f, 0.f);
  for (int64_t r = w0; r < a.rows; r += tw * LN_UNROLL) {
    float2 xv[LN_UNROLL], dv[LN_UNROLL], av[LN_UNROLL];
    float mu[LN_UNROLL], rs[LN_UNROLL];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int64_t rr = r + u * tw;
      const int64_t rc = rr < a.rows ? rr : r;  // (clamped: the loads of a row past the end are discarded below)
      xv[u] = *(const float2*)(a.x + rc * LN_D + 2 * lane);
      dv[u] = *(const float2*)(a.dy + rc * LN_D + 2 * lane);
      if (ADD) av[u] = *(const float2*)(a.add + rc * LN_D + 2 * lane);
      mu[u] = a.mean[rc];
      rs[u] = a.rstd[rc];
    }
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int64_t rr = r + u * tw;
      if (rr >= a.rows) break;
      const float hx = (xv[u].x - mu[u]) * rs[u], hy = (xv[u].y - mu[u]) * rs[u];
      const float gx = dv[u].x * gm.x, gy = dv[u].y * gm.y;
      const float s1 = tbx::wave_sum(gx + gy) * (1.0f / LN_D);
      const float s2 = tbx::wave_sum(gx * hx + gy * hy) * (1.0f / LN_D);
      float2 o;
      o.x = rs[u] * (gx - s1 - hx * s2);
      o.y = rs[u] * (gy - s1 - hy * s2);
      if (ADD) {  // (contraction off: o + add as autograd's own sum rounds it, not fused into the product above)
#pragma clang fp contract(off)
        o.x = o.x + av[u].x;
        o.y = o.y + av[u].y;
      }
      *(float2*)(a.dx + rr * LN_D + 2 * lane) = o;
      dg.x += dv[u].x * hx;
      dg.y += dv[u].y * hy;
      db.x += dv[u].x;
      db.y += dv[u].y;
    }
  }
  red[wave][2 * lane] = dg.x;
  red[wave][2 * lane + 1] = dg.y;
  red[wave][LN_D + 2 * lane] = db.x;
  red[wave][LN_D + 2 * lane + 1] = db.y;
  __syncthreads();
  if (threadIdx.x < 2 * LN_D) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LN_WAVES; ++w) s += red[w][threadIdx.x];
    a.part[(int64_t)blockIdx.x * (2 * LN_D) + threadIdx.x] = s;
  }
}

struct LnFwdArgs {
  const float *x, *gamma, *beta;
  float *y, *mean, *rstd;
  int64_t rows;
  float eps;
};

// Forward: the arithmetic of the row chains' LAYERNORM stage (csrc/rowchain.hip ln_row: two-pass mean / variance, 1 / sqrtf), so the
// time-batched pass normalises a row exactly as the stepping pass did.
__global__ __launch_bounds__(LN_WAVES * 64) void ln_fwd_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t w0 = (int64_t)blockIdx.x * LN_WAVES + wave, tw = (int64_t)gridDim.x * LN_WAVES;
  const float2 gm = *(const float2*)(a.gamma + 2 * lane), bt = *(const float2*)(a.beta + 2 * lane);
  for (int64_t r = w0; r < a.rows; r += tw * LN_UNROLL) {
    float2 xv[LN_UNROLL];
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int64_t rr = r + u * tw;
      xv[u] = *(const float2*)(a.x + (rr < a.rows ? rr : r) * LN_D + 2 * lane);
    }
#pragma unroll
    for (int u = 0; u < LN_UNROLL; ++u) {
      const int64_t rr = r + u * tw;
      if (rr >= a.rows) break;
      const float mean = tbx::wave_sum(xv[u].x + xv[u].y) / (float)LN_D;
      const float dx = xv[u].x - mean, dy = xv[u].y - mean;
      const float var = tbx::wave_sum(dx * dx + dy * dy) / (float)LN_D;
      const float rstd = 1.0f / sqrtf(var + a.eps);
      float2 o;
      o.x = dx * rstd * gm.x + bt.x;
      o.y = dy * rstd * gm.y + bt.y;
      *(float2*)(a.y + rr * LN_D + 2 * lane) = o;
      if (lane == 0) {
        a.mean[rr] = mean;
        a.rstd[rr] = rstd;
      }
    }
  }
}

// dgamma / dbeta = the column sums of the workgroups' partials [n, 256]. 16 workgroups of 16 columns (one workgroup read the 512 KiB of
// partials through ONE CU's port, a chain of 128 dependent-latency loads per thread: 19 us per call x 90 LayerNorms per training
// step): thread = (slice s of 64, column c of 16) sums partials s, s + 64, .. in order, then the 64 slices are summed in a fixed
// tree - deterministic, 64-byte segments per 16 lanes.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
  __shared__ float red[64][17];
  const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c;
  float acc = 0.f;
  for (int j = s; j < n; j += 64) acc += part[(int64_t)j * (2 * LN_D) + col];
  red[s][c] = acc;
  __syncthreads();
  for (int h = 32; h >= 1; h >>= 1) {
    if (s < h) red[s][c] += red[s + h][c];
    __syncthreads();
  }
  if (s == 0) {
    const float v = red[0][c];
    if (col < LN_D)
      dgamma[col] = v;
    else
      dbeta[col - LN_D] = v;
  }
}

int ln_workgroups(int64_t rows) {
  const int64_t want = (rows + LN_WAVES * LN_UNROLL - 1) / (LN_WAVES * LN_UNROLL);
  return (int)(want < 1 ? 1 : (want > LN_MAX_WG ? LN_MAX_WG : want));
}

}  // namespace

extern "C" int tbx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int cols, float* y,
                                 float* mean, float* rstd, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return TBX_ERR_ARG;
  if (cols != LN_D) return TBX_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 7) return TBX_ERR_ALIGN;
  LnFwdArgs a{x, gamma, beta, y, mean, rstd, rows, eps};
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_workgroups(rows)), dim3(LN_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_layernorm_bwd_partials(int64_t rows) { return ln_workgroups(rows); }

extern "C" int tbx_layernorm_bwd_add(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                                     int64_t rows, int cols, const float* add, float* dx, float* dgamma, float* dbeta, float* scratch,
                                     void* stream) {
  if (!x || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !scratch || rows <= 0) return TBX_ERR_ARG;
  if (cols != LN_D) return TBX_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)gamma) | ((uintptr_t)add)) & 7) return TBX_ERR_ALIGN;
  LnBwdArgs a{x, dy, gamma, mean, rstd, dx, scratch, rows, add};
  const int n = ln_workgroups(rows);
  hipStream_t s = (hipStream_t)stream;
  if (add != nullptr) hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(n), dim3(LN_WAVES * 64), 0, s, a);
  else hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(n), dim3(LN_WAVES * 64), 0, s, a);
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(2 * LN_D / 16), dim3(1024), 0, s, (const float*)scratch, n, dgamma, dbeta);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                                 int64_t rows, int cols, float* dx, float* dgamma, float* dbeta, float* scratch, void* stream) {
  return tbx_layernorm_bwd_add(x, dy, gamma, mean, rstd, rows, cols, nullptr, dx, dgamma, dbeta, scratch, stream);
}
