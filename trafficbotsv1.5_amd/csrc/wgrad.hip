// Weight gradient of a tall-skinny LINEAR (tbx_linear_wgrad, include/tbx_hip.h): dW[N,K] = dY[R,N]^T X[R,K], db[N] = sum_r dY,
// with R = 10^5..10^6 rows (the time-batched training pass: every closed-loop step of every scene is a row block) and
// N, K <= a few hundred. The library GEMM for this shape (one or two output tiles, reduction depth R) ran at 2.5-6.7 ms per call;
// the shape is HBM-bound (both operands are read once: 2 GB at R = 2 M, N = K = 128) with the exact-fp32 MFMA as second bound.
//
// A wavefront accumulates a 64 x 64 block of dW over a range of rows. Per 4 rows every lane loads ONE float4 of dY and ONE of X
// (lane l: row l >> 4, columns 4 (l & 15) .. +3 of the block: 256 B contiguous per row) and issues 16 v_mfma_f32_16x16x4_f32:
// MFMA (t, u) takes component t of the dY float4 as A[i = l & 15][kk = l >> 4] and component u of the X float4 as B, i.e. it
// accumulates the strided 16 x 16 tile dW[n0 + 4 i + t][k0 + 4 j + u] - 16 independent accumulators (64 VGPRs), 2 loads per
// 16 MFMAs, no LDS in the loop. A workgroup = one block over one row range (its 4 waves a quarter of the rows each, summed through
// LDS at the end); partial blocks land in a scratch buffer and a second kernel sums them (deterministic: no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define TBX_GLOBAL __attribute__((address_space(1)))

struct WgradArgs {
  const float* dy;
  const float* x;
  float* part;  // [splits][n*k + n]
  int64_t rows, rows_per_split;
  int ld_dy, ld_x, n, k, tiles_k, tiles, with_db;
};

constexpr int P = 4;  // 4-row groups loaded ahead per pipeline stage (16 rows)
constexpr int N_XCD = 8;

// (Non-temporal operand loads were measured and dropped in round 6: a read-only pass over 1 GB streams at 0.90 of the HBM peak with nt
// against 0.80 without (profiles/r06_stream_probe.txt), but inside the training step these operands were just WRITTEN by their
// producers - they come out of the 256 MiB Infinity Cache / the XCD's L2, which the tiles of one row range share - and nt gives that
// up: tbx_linear_wgrad_bf16 13.7 -> 16.4 ms per step, tbx_tall_linear_bf16 17.5 -> 18.2, ln_fwd 1.8 -> 2.1, the step 145.8 -> 148.2 ms.)
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const TBX_GLOBAL f32x4*)p; }

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
  const bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return __builtin_bit_cast(s16x4, v);
}

// Round 6: ONE 64 x 64 block of dW per WORKGROUP - its 4 waves take a quarter of the workgroup's rows each and are summed through LDS -
// instead of a 2 x 2 group of blocks (one per wave) over the same rows:
//   * partial sums: a workgroup hands 16 KiB to the scratch buffer instead of 64 - at 768 workgroups 12.6 MB instead of 50 MB written
//     and read back (a [92,160 x 128]^T [92,160 x 128] call reads 94 MB of operands: the partials were half its traffic);
//   * operand re-reads: the tiles of ONE row range - (n / 64) (k / 64) workgroups, 20 for a 640 x 128 gradient - read the same dY / X
//     rows. The workgroup index is re-mapped so that they sit on ONE XCD, dispatched back to back (hardware deals consecutive
//     workgroup ids round-robin over the 8 XCDs, each with its own L2): the re-reads are L2 hits instead of HBM reads on 5 different
//     XCDs (measured before: 2.4-3.3 TB/s of algorithmic bytes on the 640-wide shapes = ~1.7 x that in HBM traffic).
// BF16 = tbx_linear_wgrad_bf16: ONE bf16 product per term - dY and X rounded to bfloat16 in registers (v_cvt_pk_bf16_f32), fp32
// accumulation over the rows - what torch's autocast(bfloat16) gives a weight gradient (the reference trains at precision 16,
// configs/trainer/default.yaml:16). The contraction index of an MFMA is the ROW, and a lane already holds what the instruction wants
// from it: lane (rr = l >> 4, cq = l & 15) has loaded, for the 4 row groups q = 0..3 of a 16-row stage, rows 4 q + rr of its 4 columns -
// component t of its 4 dY float4s IS the A operand A[i = cq][kk = 4 rr + q] of v_mfma_f32_16x16x16_bf16 for the strided tile
// n = n0 + 4 i + t (the order of the 16 rows inside the contraction is free, and the same for both operands), component u of its 4 X
// float4s the B operand: 16 MFMAs + 16 packed conversions per 16 rows instead of 64 exact-fp32 MFMAs.
// (A split-bf16 form - three products with the ROW as the reduction index - was built in round 3 and measured slower than the
// exact-fp32 kernel: git history, profiles/MEASUREMENT_LOG.md.)
template <bool BF16>
__global__ __launch_bounds__(256, 2) void wgrad_partial_kernel(const WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float red[3][68][64];  // waves 1..3: 64 accumulator registers + 4 of the bias sum, per lane
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // workgroup id -> (row range, tile): ids congruent mod 8 share an XCD; virtual index v runs through one XCD's ids first
  const unsigned G = gridDim.x, L = blockIdx.x;
  const unsigned xcd = L % N_XCD, slot = L / N_XCD, per = G / N_XCD, rem = G % N_XCD;
  const unsigned v = slot + xcd * per + (xcd < rem ? xcd : rem);
  const int split = (int)(v / (unsigned)a.tiles), tile = (int)(v % (unsigned)a.tiles);
  const int n0 = (tile / a.tiles_k) * 64, k0 = (tile % a.tiles_k) * 64;
  const int64_t rq = a.rows_per_split >> 2;  // rows per wave (a multiple of 16)
  const int64_t r0 = (int64_t)split * a.rows_per_split + wave * rq;
  const int64_t r1 = r0 + rq < a.rows ? r0 + rq : a.rows;
  const int rr = lane >> 4, c = (lane & 15) * 4;
  const bool n_ok = n0 + c < a.n, k_ok = k0 + c < a.k;  // n, k are multiples of 4: a float4 is wholly inside or outside
  const float* py = a.dy + n0 + c;
  const float* px = a.x + k0 + c;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = zero;
  f32x4 bsum = zero;
  const bool want_db = a.with_db && k0 == 0;
  f32x4 ya[P], xa[P], yb[P], xb[P];
  auto load = [&](f32x4(&y)[P], f32x4(&x)[P], int64_t r) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int64_t row = r + q * 4 + rr;
      const bool live = row < r1;
      y[q] = (live && n_ok) ? ldg4(py + row * a.ld_dy) : zero;
      x[q] = (live && k_ok) ? ldg4(px + row * a.ld_x) : zero;
    }
  };
  load(ya, xa, r0);
  for (int64_t r = r0; r < r1; r += 4 * P) {
    load(yb, xb, r + 4 * P);
    if constexpr (BF16) {
      if (want_db) bsum += (ya[0] + ya[1]) + (ya[2] + ya[3]);  // (the bias gradient stays an exact fp32 sum)
      s16x4 ay[4], bx[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ay[t] = pack4(ya[0][t], ya[1][t], ya[2][t], ya[3][t]);
        bx[t] = pack4(xa[0][t], xa[1][t], xa[2][t], xa[3][t]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ay[t], bx[u], acc[t][u], 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < P; ++q) {
        if (want_db) bsum += ya[q];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[q][t], xa[q][u], acc[t][u], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) ya[q] = yb[q], xa[q] = xb[q];
  }
  // ---- the four waves' blocks summed in a fixed order (wave 0 + 1 + 2 + 3: deterministic); register i of lane l at red[w][i][l]
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wave - 1][(t * 4 + u) * 4 + reg][lane] = acc[t][u][reg];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave - 1][64 + e][lane] = bsum[e];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll 1
  for (int w = 0; w < 3; ++w) {  // (not unrolled: the three blocks' 204 LDS reads per lane would all be hoisted into registers)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[t][u][reg] += red[w][(t * 4 + u) * 4 + reg][lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[e] += red[w][64 + e][lane];
  }
  // acc[t][u][reg] at lane l = dW[n0 + 4 ((l >> 4) * 4 + reg) + t][k0 + 4 (l & 15) + u]
  float* part = a.part + (int64_t)split * ((int64_t)a.n * a.k + a.n);
  if (k_ok) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = n0 + 4 * (rr * 4 + reg) + t;
        if (n < a.n) {
          f32x4 vv = {acc[t][0][reg], acc[t][1][reg], acc[t][2][reg], acc[t][3][reg]};
          *(TBX_GLOBAL f32x4*)(part + (int64_t)n * a.k + k0 + c) = vv;
        }
      }
  }
  if (want_db) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float vv = bsum[e];
      vv += __shfl_xor(vv, 16);
      vv += __shfl_xor(vv, 32);
      bsum[e] = vv;
    }
    if (rr == 0 && n_ok) *(TBX_GLOBAL f32x4*)(part + (int64_t)a.n * a.k + n0 + c) = bsum;
  }
}

// out[e] = sum_g part[g][e], e < total (dW then db): a workgroup per 64 outputs, its 4 waves take a quarter of the splits each
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int64_t total, int nk,
                                                           float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float s[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float v = 0.f;
  if (e < total) {
    float v2 = 0.f, v3 = 0.f, v4 = 0.f;
    int g = wave;
    for (; g + 12 < splits; g += 16) {
      v += part[(int64_t)g * total + e];
      v2 += part[(int64_t)(g + 4) * total + e];
      v3 += part[(int64_t)(g + 8) * total + e];
      v4 += part[(int64_t)(g + 12) * total + e];
    }
    for (; g < splits; g += 4) v += part[(int64_t)g * total + e];
    v = (v + v2) + (v3 + v4);
  }
  s[wave][lane] = v;
  __syncthreads();
  if (wave == 0 && e < total) {
    const float r = (s[0][lane] + s[1][lane]) + (s[2][lane] + s[3][lane]);
    if (e < nk)
      dw[e] = r;
    else if (db != nullptr)
      db[e - nk] = r;
  }
}

}  // namespace

extern "C" int tbx_linear_wgrad_splits(int64_t rows, int n, int k) {
  if (rows <= 0 || n <= 0 || k <= 0) return TBX_ERR_ARG;
  const int tiles = ((n + 63) / 64) * ((k + 63) / 64);
  static const int target = [] {
    const char* e = getenv("TBX_WGRAD_WGS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 768;
  }();
  int64_t s = target / tiles;  // 768 workgroups = 3 per CU = the 3 wavefronts per SIMD the kernel's registers allow: one full wave of workgroups, no tail
  if (s < 1) s = 1;
  const int64_t cap = (rows + 63) / 64;  // at least 64 rows per split (16 per wave)
  if (s > cap) s = cap;
  return (int)s;
}

static int wgrad_launch(bool bf16, const float* dy, int ld_dy, const float* x, int ld_x, int64_t rows, int n, int k, float* dw, float* db,
                        float* scratch, int splits, void* stream) {
  if (!dy || !x || !dw || !scratch || rows <= 0 || n <= 0 || k <= 0 || splits <= 0) return TBX_ERR_ARG;
  if ((n & 3) || (k & 3) || (ld_dy & 3) || (ld_x & 3) || ld_dy < n || ld_x < k) return TBX_ERR_UNSUPPORTED;
  if ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)scratch)) & 15) return TBX_ERR_ALIGN;
  WgradArgs a;
  a.dy = dy, a.x = x, a.part = scratch, a.rows = rows, a.ld_dy = ld_dy, a.ld_x = ld_x, a.n = n, a.k = k;
  const int rstep = 4 * 4 * P;  // 4 waves x 16 rows
  a.rows_per_split = (((rows + splits - 1) / splits + rstep - 1) / rstep) * rstep;
  a.tiles_k = (k + 63) / 64;
  a.tiles = ((n + 63) / 64) * a.tiles_k;
  a.with_db = db != nullptr;
  const int64_t grid = (int64_t)a.tiles * splits;
  if (grid > 0x7fffffff) return TBX_ERR_UNSUPPORTED;
  hipStream_t hs = (hipStream_t)stream;
  if (bf16) hipLaunchKernelGGL(wgrad_partial_kernel<true>, dim3((unsigned)grid), dim3(256), 0, hs, a);
  else hipLaunchKernelGGL(wgrad_partial_kernel<false>, dim3((unsigned)grid), dim3(256), 0, hs, a);
  if (hipGetLastError() != hipSuccess) return TBX_ERR_LAUNCH;
  const int64_t total = (int64_t)n * k + n;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, hs, scratch, splits, total, n * k, dw, db);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_linear_wgrad(const float* dy, int ld_dy, const float* x, int ld_x, int64_t rows, int n, int k, float* dw, float* db,
                                float* scratch, int splits, void* stream) {
  return wgrad_launch(false, dy, ld_dy, x, ld_x, rows, n, k, dw, db, scratch, splits, stream);
}

extern "C" int tbx_linear_wgrad_bf16(const float* dy, int ld_dy, const float* x, int ld_x, int64_t rows, int n, int k, float* dw, float* db,
                                     float* scratch, int splits, void* stream) {
  return wgrad_launch(true, dy, ld_dy, x, ld_x, rows, n, k, dw, db, scratch, splits, stream);
}
