// tbx_tall_linear: y = x W^T (+ b) over VERY many rows (training's time-batched pass: 10^5 .. 2 x 10^6 rows, K and N multiples of
// 128 up to 640 - or of 64: the image is then that of the weight zero-padded to the next multiple of 128) - the forward and input-gradient products of every nn.Linear of the differentiated pass, which the library ran at
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
// (TBX_TILE_SINGLE build = tbx_tall_linear_bf16: one product, hi planes and the hi halves of the weight units only)
constexpr int NPL = TBX_TILE_SINGLE ? 1 : 2;        // planes per buffer: hi (, lo)
constexpr size_t LDS_BYTES = 2 * NPL * PLANE;       // two buffers

struct TallArgs {
  const float* x;
  const float* img;  // tbx_pack_weight_mfma32 image of W [n x k] (+ bias)
  float* y;
  int64_t m;
  int ldx, ldy, k, n, has_bias, relu;
  uint16_t* y16;  // optional second output: the same rows as bfloat16 [m, ldy16] (the K/V tables the matrix-core attention gathers)
  int ldy16;
  // optional keyed dropout behind the relu (tbx_tall_linear_relu_drop: the FFN's hidden activation / an MLP layer's output as ONE
  // launch instead of LINEAR + tbx_relu_drop_fwd): tbx_keyed_dropout's mask for the [m, n] tensor; drop_thresh == 0: none
  const uint64_t* drop_seed;
  uint32_t drop_site, drop_thresh;
  float drop_scale;
  int rows_per_scene, time_batch, time0;
};

// tbx_keyed_dropout's mask (csrc/dropout.hip `mix`): the per-row part of the key (step of the row's batch entry, its row inside the scene)
// once per row block, the per-element hash in the epilogue
struct TallRowKey {
  uint32_t lo, hi, base;  // base = key row * n
};
__device__ __forceinline__ TallRowKey tall_row_key(const TallArgs& a, const uint64_t sd, const uint32_t row) {
  const uint32_t b = row / (uint32_t)a.rows_per_scene;  // (m < 2^31: 32-bit divisions)
  const uint32_t sc = b / (uint32_t)a.time_batch;
  const uint32_t ts = (uint32_t)a.time0 + (b - sc * (uint32_t)a.time_batch);
  const uint32_t krow = sc * (uint32_t)a.rows_per_scene + (row - b * (uint32_t)a.rows_per_scene);
  TallRowKey k;
  k.lo = (uint32_t)sd ^ (a.drop_site * 0x85EBCA6Bu) ^ (ts * 0x27D4EB2Fu);
  k.hi = (uint32_t)(sd >> 32) + a.drop_site * 0xC2B2AE35u + ts * 0x165667B1u;
  k.base = krow * (uint32_t)a.n;
  return k;
}
__device__ __forceinline__ f32x4 tall_drop4(const TallArgs& a, const TallRowKey& k, const int c, f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uint32_t x = (k.base + (uint32_t)(c + r)) ^ k.lo;
    x *= 0x9E3779B1u;
    x ^= k.hi;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    v[r] = x >= a.drop_thresh ? v[r] * a.drop_scale : 0.f;
  }
  return v;
}

__global__ __launch_bounds__(NT) void tall_linear_kernel(const TallArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_c[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  // 128-wide K chunks, N blocks; 16-channel tiles of the image. k, n are multiples of 64: a trailing half chunk / half block (the 64-wide
  // PointNet layers) runs on the image of the weight zero-padded to the next multiple of 128, its loads / stores masked by column
  const int KC = (a.k + 127) >> 7, NB = (a.n + 127) >> 7, T = NB << 3;
  const int n_rb = (int)((a.m + ROWS - 1) / ROWS);        // row blocks: this workgroup takes blockIdx.x, + gridDim.x, ...
  const int my_rb = (n_rb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  // Index arithmetic is what this loop must NOT spend its issue slots on (a matrix instruction leaves ~4 of them): everything below
  // is pointers stepped by precomputed strides and 32-bit counters.
  // the thread's share of a 64 x 128 chunk of rows: 4 float4 (row q * 16 + lr, columns lc ..)
  const int lr = tid >> 5, lc = (tid & 31) * 4;
  const int64_t x16 = 16 * (int64_t)a.ldx, y16 = 16 * (int64_t)a.ldy;
  const int64_t x_rb = (int64_t)gridDim.x * ROWS * a.ldx, y_rb = (int64_t)gridDim.x * ROWS * a.ldy;  // to this workgroup's next row block
  const float* xp = a.x + ((int64_t)blockIdx.x * ROWS + lr) * a.ldx + lc;                // the row block whose rows are requested next
  float* yp = a.y + ((int64_t)blockIdx.x * ROWS + j) * a.ldy + 16 * wave + 4 * g;        // the row block being multiplied
  uint16_t* yh = a.y16 == nullptr ? nullptr : a.y16 + ((int64_t)blockIdx.x * ROWS + j) * a.ldy16 + 16 * wave + 4 * g;
  const int64_t h16 = 16 * (int64_t)a.ldy16, h_rb = (int64_t)gridDim.x * ROWS * a.ldy16;
  const TBX_GLOBAL float* wq = (const TBX_GLOBAL float*)a.img + (int64_t)wave * UNIT + lane * 4;  // the wave's units; the lane's 16 bytes
  int rows_req = (int)(a.m - (int64_t)blockIdx.x * ROWS);  // rows left from the requested row block on (may exceed 64)
  int rows_cur = rows_req;
  int64_t row_cur = (int64_t)blockIdx.x * ROWS;  // first global row of the row block being multiplied
  const uint64_t drop_sd = a.drop_thresh != 0u ? *(const TBX_GLOBAL uint64_t*)a.drop_seed : 0ull;
  TallRowKey rkey[4];  // the lane's 4 output rows (q * 16 + j) of the current row block
#pragma unroll
  for (int q = 0; q < 4; ++q) rkey[q] = a.drop_thresh != 0u ? tall_row_key(a, drop_sd, (uint32_t)(row_cur + q * 16 + j)) : TallRowKey{0u, 0u, 0u};
  f32x4 xin[4];
  auto request_x = [&](int kc) {  // rows of the row block at xp, K chunk kc
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      xin[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (q * 16 + lr < rows_req && kc * 128 + lc < a.k) xin[q] = *(const TBX_GLOBAL f32x4*)(xp + q * x16 + kc * 128);  // (default cache policy: nt measured slower, see wgrad.hip)
    }
  };
  auto park_x = [&](char* P) {
#pragma unroll
    for (int q = 0; q < 4; ++q) planes_write4<PL>(P, q * 16 + lr, lc, xin[q]);
  };
  auto load_w = [&](W& w, int unit) {  // unit = kc * T + nb * 8 (+ wave: in wq)
    const TBX_GLOBAL float* base = wq + (uint32_t)unit * (uint32_t)UNIT;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      w.hi[s] = *(const TBX_GLOBAL bf16x8*)(base + s * 512);
#if !TBX_TILE_SINGLE
      w.lo[s] = *(const TBX_GLOBAL bf16x8*)(base + s * 512 + 256);
#endif
    }
    w.bias = *(const TBX_GLOBAL f32x4*)(base - lane * 4 + 2048 + g * 4);
  };
  W wb[2];
  int buf = 0;
  request_x(0);
  load_w(wb[0], 0);
  park_x(lds_c);
  __syncthreads();
  Acc acc[4];
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  // iterations of this workgroup: (its row blocks) x (N blocks) x (K chunks), K innermost; two per loop trip: the register slots stay
  // compile-time. A chunk of rows (row block, K chunk) is parked in LDS when the iteration before its first use ends - with K = 128
  // the planes serve all N blocks of the row block, and the next row block's rows are requested under the last N block.
  const int I = my_rb * NB * KC;
  int nb = 0, kc = 0;
  auto body = [&](const int it, const W& cur, W& nxt) {
    if (kc == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q].zero();
      bias = cur.bias;
    }
    const bool more = it + 1 < I;
    int nb2 = nb, kc2 = kc + 1;
    bool next_rb = false;
    if (kc2 == KC) {
      kc2 = 0;
      if (++nb2 == NB) nb2 = 0, next_rb = true;
    }
    const bool new_x = more && (KC > 1 || next_rb);  // the next iteration reads another chunk of rows
    if (more && next_rb) xp += x_rb, rows_req -= (int)gridDim.x * ROWS;
    if (new_x) request_x(kc2);
    if (more) load_w(nxt, kc2 * T + nb2 * 8);
    const char* P = lds_c + buf * NPL * PLANE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int aoff = PL::lane_off(lane, q * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc[q], cur.hi[s], cur.lo[s], P + aoff, s);
    }
    if (new_x) {
      park_x(lds_c + (buf ^ 1) * NPL * PLANE);  // (the other buffer: last read before the previous barrier)
      __syncthreads();
      buf ^= 1;
    }
    if (kc + 1 == KC) {  // the N block's 64 x 128 outputs: lane = (row tile q, row j, channels 16 * wave + 4 g ..)
      // (Through an LDS tile as whole 512-byte rows - instead of 16 rows x 64 bytes per wave instruction - was measured in round 6:
      // 184,320 x 128 -> 640 152 -> 147 us, but 92,160 x 128 x 128 25.7 -> 31.1 and the 16,384-row calls + 30 %: two more barriers per N
      // block; the class + 1 ms per step. Dropped.)
      float* yo = yp + nb * 128;
      const int c0 = nb * 128 + 16 * wave + 4 * g;
      if (c0 < a.n) {  // (a trailing half block: the 64-wide layers)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = acc[q].sum();
          if (a.has_bias) v += bias;
          if (a.relu) v = relu4(v);
          if (a.drop_thresh != 0u) v = tall_drop4(a, rkey[q], c0, v);
          if (q * 16 + j < rows_cur) {
            *(TBX_GLOBAL f32x4*)(yo + q * y16) = v;
            if (yh != nullptr) *(TBX_GLOBAL u32x2*)(yh + nb * 128 + q * h16) = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
          }
        }
      }
      if (next_rb) {
        yp += y_rb, rows_cur -= (int)gridDim.x * ROWS, row_cur += (int64_t)gridDim.x * ROWS;
        if (yh != nullptr) yh += h_rb;
        if (a.drop_thresh != 0u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) rkey[q] = tall_row_key(a, drop_sd, (uint32_t)(row_cur + q * 16 + j));
        }
      }
    }
    nb = nb2, kc = kc2;
  };
  for (int it = 0; it < I; it += 2) {
    body(it, wb[0], wb[1]);
    if (it + 1 < I) body(it + 1, wb[1], wb[0]);
  }
}

}  // namespace

// the device's CU count, cached per device ordinal (as tile_layer.hip's cu_count)
static int tall_cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

struct TallDrop {
  float p;
  const uint64_t* seed;
  uint32_t site;
  int rows_per_scene, time_batch, time0;
};

static int tall_launch(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                       uint16_t* y16, int ldy16, void* stream, const TallDrop* drop = nullptr) {
  if (x == nullptr || image == nullptr || y == nullptr || m <= 0) return TBX_ERR_ARG;
  if (k <= 0 || n <= 0 || (k % 64) || (n % 64) || k > 1024 || n > 1024) return TBX_ERR_UNSUPPORTED;
  if (ldx < k || ldy < n || (ldx % 4) || (ldy % 4)) return TBX_ERR_ARG;
  if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)image)) & 15) return TBX_ERR_ALIGN;
  if (y16 != nullptr && (ldy16 < n || (ldy16 % 4) || (((uintptr_t)y16) & 7))) return TBX_ERR_ALIGN;
  TallArgs a{x, image, y, m, ldx, ldy, k, n, has_bias, relu, y16, ldy16, nullptr, 0u, 0u, 1.0f, 1, 1, 0};
  if (drop != nullptr && drop->p > 0.f) {
    if (drop->p >= 1.f || !drop->seed || drop->rows_per_scene <= 0 || drop->time_batch < 1 || drop->time0 < 0 || m % drop->rows_per_scene) return TBX_ERR_ARG;
    if (m > 0x7fffffff) return TBX_ERR_UNSUPPORTED;  // (the mask's row arithmetic is 32-bit)
    const double th = (double)drop->p * 4294967296.0;  // (tbx_keyed_dropout's threshold and scale)
    a.drop_seed = drop->seed, a.drop_site = drop->site, a.drop_thresh = th < 1.0 ? 1u : (uint32_t)th, a.drop_scale = 1.0f / (1.0f - drop->p);
    a.rows_per_scene = drop->rows_per_scene, a.time_batch = drop->time_batch, a.time0 = drop->time0;
  }
  static tbx::PerDeviceOnce lds_attr;  // (per device, thread-safe: tbx_common.h)
  if (!lds_attr([&] { return !(hipFuncSetAttribute((const void*)tall_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess); })) return TBX_ERR_LAUNCH;
  const int64_t n_rb = (m + ROWS - 1) / ROWS;
  // one workgroup per CU, striding over the row blocks (TBX_TALL_GRID: an explicit grid size for A/B runs, clamped to >= 1)
  static const int grid_env = [] { const char* e = getenv("TBX_TALL_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const int grid_max = grid_env > 0 ? grid_env : tall_cu_count();
  hipLaunchKernelGGL(tall_linear_kernel, dim3((unsigned)(n_rb < grid_max ? n_rb : grid_max)), dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int TBX_TILE_ENTRY(tbx_tall_linear)(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                                               void* stream) {
  return tall_launch(x, m, k, ldx, image, n, has_bias, relu, y, ldy, nullptr, 0, stream);
}

extern "C" int TBX_TILE_ENTRY(tbx_tall_linear_dual)(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y,
                                                    int ldy, uint16_t* y16, int ldy16, void* stream) {
  if (y16 == nullptr) return TBX_ERR_ARG;
  return tall_launch(x, m, k, ldx, image, n, has_bias, relu, y, ldy, y16, ldy16, stream);
}

extern "C" int TBX_TILE_ENTRY(tbx_tall_linear_relu_drop)(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, float* y,
                                                         int ldy, float p_drop, const uint64_t* drop_seed, uint32_t site, int rows_per_scene,
                                                         int time_batch, int time0, void* stream) {
  if (p_drop < 0.f) return TBX_ERR_ARG;
  const TallDrop d{p_drop, drop_seed, site, rows_per_scene, time_batch, time0};
  return tall_launch(x, m, k, ldx, image, n, has_bias, 1, y, ldy, nullptr, 0, stream, &d);
}
