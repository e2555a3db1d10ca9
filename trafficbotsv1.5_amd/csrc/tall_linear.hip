// tbx_tall_linear: y = x W^T (+ b) over VERY many rows (training's time-batched pass: 10^5 .. 2 x 10^6 rows, K and N multiples of
// 128 up to 640) - the forward and input-gradient products of every nn.Linear of the differentiated pass, which the library ran at
// 50-105 TF/s of exact-fp32 MFMA (tools/train_gemm_shapes.py: 33 ms of a 200 ms step). These products are BYTE-bound if the
// arithmetic is cheap enough - (K + N) x 4 B per row against 2 K N flops - so this kernel does the arithmetic on the split-bf16
// matrix path of the tile kernels (tile_core.h: x = x_hi + x_lo, w = w_hi + w_lo in bf16, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation: < 3e-5 of sum |x||w| per output, ~5 x the fp32 MFMA rate) and streams the rows:
//   * a workgroup (8 waves) owns 64 rows = 4 row tiles of 16; per 128-wide K chunk the rows are loaded once (coalesced float4),
//     split once and parked in LDS as bf16 hi / lo planes (the conflict-free layout of tile_core.h), double-buffered: the next
//     chunk's loads are in flight while this chunk multiplies;
//   * wave w owns output tile w (16 channels) of the current 128-wide N block and holds its weights as a register unit
//     (tbx_pack_weight_mfma32: 8 KiB per [16 channels x 128 k]), the next unit in flight; a unit serves the 4 row tiles (48 MFMAs);
//   * products are formed transposed (D = W x^T): a lane ends with 4 consecutive channels of one row: one 16-byte store.
// Loop nest: N blocks outside, K chunks inside (every shape of the pass has min(K, N) = 128 or 256: the rows are read once when
// K = 128 - their planes stay resident across the N blocks - and once per N block otherwise).
#include <stdlib.h>

#include "tile_core.h"

using namespace tbx_tile;

namespace {

constexpr int ROWS = 64;
typedef Planes<ROWS, 4> PL;
constexpr int PLANE = PL::PLANE;
constexpr size_t LDS_BYTES = 2 * 2 * PLANE;  // two buffers of (hi, lo)

struct TallArgs {
  const float* x;
  const float* img;  // tbx_pack_weight_mfma32 image of W [n x k] (+ bias)
  float* y;
  int64_t m;
  int ldx, ldy, k, n, has_bias, relu;
};

__global__ __launch_bounds__(NT) void tall_linear_kernel(const TallArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_c[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int KC = a.k >> 7, NB = a.n >> 7, T = a.n >> 4;  // 128-wide K chunks, N blocks; 16-channel tiles of the image
  const int64_t n_rb = (a.m + ROWS - 1) / ROWS;          // row blocks: this workgroup takes blockIdx.x, + gridDim.x, ...
  const int64_t my_rb = (n_rb - (int64_t)blockIdx.x + gridDim.x - 1) / gridDim.x;
  // the thread's share of a 64 x 128 chunk of rows: 4 float4 (row r = q * 16 + (tid >> 5), columns (tid & 31) * 4)
  const int lr = tid >> 5, lc = (tid & 31) * 4;
  f32x4 xin[4];
  auto request_x = [&](int64_t rb, int kc) {
    const int64_t r0 = rb * ROWS;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = r0 + q * 16 + lr;
      xin[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (r < a.m) xin[q] = gld4(a.x + r * (int64_t)a.ldx + kc * 128 + lc);
    }
  };
  auto park_x = [&](char* P) {
#pragma unroll
    for (int q = 0; q < 4; ++q) planes_write4<PL>(P, q * 16 + lr, lc, xin[q]);
  };
  W wb[2];
  int buf = 0;
  request_x(blockIdx.x, 0);
  load_unit(wb[0], a.img, wave, lane);  // (N block 0, K chunk 0): unit = kc * T + tile
  park_x(lds_c);
  __syncthreads();
  Acc acc[4];
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  // iterations of this workgroup: (its row blocks) x (N blocks) x (K chunks), K innermost; two per loop trip: the register slots stay
  // compile-time. A chunk of rows (row block, K chunk) is parked in LDS when the iteration before its first use ends - with K = 128
  // the planes serve all N blocks of the row block, and the next row block's rows are requested under the last N block.
  const int per_rb = NB * KC;
  const int64_t I = my_rb * per_rb;
  auto body = [&](const int64_t it, const W& cur, W& nxt) {
    const int64_t ib = it / per_rb;
    const int in = (int)(it - ib * per_rb);
    const int nb = in / KC, kc = in - nb * KC;
    const int64_t rb = (int64_t)blockIdx.x + ib * gridDim.x;
    if (kc == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q].zero();
      bias = cur.bias;
    }
    const bool more = it + 1 < I;
    const int64_t ib2 = (it + 1) / per_rb;
    const int in2 = (int)((it + 1) - ib2 * per_rb);
    const int nb2 = in2 / KC, kc2 = in2 - nb2 * KC;
    const bool new_x = more && (KC > 1 || ib2 != ib);  // the next iteration reads another chunk of rows
    if (new_x) request_x((int64_t)blockIdx.x + ib2 * gridDim.x, kc2);
    if (more) load_unit(nxt, a.img, kc2 * T + nb2 * 8 + wave, lane);
    const char* P = lds_c + buf * 2 * PLANE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int aoff = PL::lane_off(lane, q * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc[q], cur.hi[s], cur.lo[s], P + aoff, s);
    }
    if (new_x) {
      park_x(lds_c + (buf ^ 1) * 2 * PLANE);  // (the other buffer: last read before the previous barrier)
      __syncthreads();
      buf ^= 1;
    }
    if (kc + 1 == KC) {  // the N block's 64 x 128 outputs: lane = (row tile q, row j, channels 16 * wave + 4 g ..)
      const int c = nb * 128 + 16 * wave + 4 * g;
      const int64_t r0 = rb * ROWS;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t r = r0 + q * 16 + j;
        f32x4 v = acc[q].sum();
        if (a.has_bias) v += bias;
        if (a.relu) v = relu4(v);
        if (r < a.m) gst4(a.y + r * (int64_t)a.ldy + c, v);
      }
    }
  };
  for (int64_t it = 0; it < I; it += 2) {
    body(it, wb[0], wb[1]);
    if (it + 1 < I) body(it + 1, wb[1], wb[0]);
  }
}

}  // namespace

extern "C" int tbx_tall_linear(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                               void* stream) {
  if (x == nullptr || image == nullptr || y == nullptr || m <= 0) return TBX_ERR_ARG;
  if (k <= 0 || n <= 0 || (k % 128) || (n % 128) || k > 1024 || n > 1024) return TBX_ERR_UNSUPPORTED;
  if (ldx < k || ldy < n || (ldx % 4) || (ldy % 4)) return TBX_ERR_ARG;
  if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)image)) & 15) return TBX_ERR_ALIGN;
  TallArgs a{x, image, y, m, ldx, ldy, k, n, has_bias, relu};
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)tall_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return TBX_ERR_LAUNCH;
    attr_set = true;
  }
  const int64_t n_rb = (m + ROWS - 1) / ROWS;
  static const int grid_max = [] { const char* e = getenv("TBX_TALL_GRID"); return e ? atoi(e) : 256; }();  // one workgroup per CU, striding over the row blocks
  hipLaunchKernelGGL(tall_linear_kernel, dim3((unsigned)(n_rb < grid_max ? n_rb : grid_max)), dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
