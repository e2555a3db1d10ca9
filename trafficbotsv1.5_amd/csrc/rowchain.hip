// Row-tile chain interpreter (tbx_rowchain): see include/tbx_hip.h for the stage semantics.
//
// One workgroup of 16 wavefronts (16-row tiles) or 8 wavefronts (32-row tiles) owns TILE_ROWS = 16*MT rows. Activations stay in LDS between stages
// (two ping-pong buffers of `ldw` floats per row + a 260-float auxiliary buffer for residuals); weights are streamed
// straight from L2/HBM into VGPRs once per tile (GEMV-like regime: no reuse across waves, so no LDS staging) and fed
// to v_mfma_f32_16x16x4_f32, which is an exact fp32 fma chain. Wave w owns output column tiles w, w+16, ...
//
// MFMA operand mapping (guide: cdna_hip_programming.md §3): A lane l supplies A[i=l&15][k=l>>4], B lane l supplies
// B[k=l>>4][j=l&15], C/D: col = l&15, row = (l>>4)*4 + reg. K is walked in blocks of 16 with the 4 MFMAs of a block
// taking k = kb*16 + (l>>4)*4 + t, so every lane reads ONE float4 of activations (ds_read_b128) and ONE float4 of
// weights (global_load_dwordx4) per 4 MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

struct RowchainArgs {
  tbx_stage_t st[TBX_MAX_STAGES];
  int32_t n_stages;
  int32_t group_rows;
  int32_t ldw0, ldw1, ld_aux;
  int64_t n_rows;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, bool EXT>
struct Tile {
  static constexpr int ROWS = 16 * MT;
  float* base;    // LDS base; BUF0 = ROWS*ldw0 floats, then BUF1 = ROWS*ldw1, then AUX = ROWS*ld_aux
  int ldw0, ldw1, ld_aux;
  // computed, not indexed: a runtime-indexed member array would live in scratch memory
  // EXT = per-buffer widths + LINEAR-to-global (tbx_rowchain_ex); the plain layout (two ldw0-wide buffers + a 260-wide
  // auxiliary) keeps the address arithmetic of the common small-grid programs minimal (measured: 7 % on a C2 step)
  __device__ __forceinline__ float* b(int i) const {
    if constexpr (EXT) return base + (i == 0 ? 0 : (i == 1 ? ROWS * ldw0 : ROWS * (ldw0 + ldw1)));
    return base + (i == TBX_BUF_AUX ? 2 : i) * ROWS * ldw0;
  }
  __device__ __forceinline__ int l(int i) const {
    if constexpr (EXT) return i == 0 ? ldw0 : (i == 1 ? ldw1 : ld_aux);
    return i == TBX_BUF_AUX ? TBX_AUX_LD : ldw0;
  }
  int64_t g0;      // first global row of the tile
  int n_valid;     // rows r < n_valid map to a global row
  int64_t group;   // group index (grouped mode) or tile index
};

__device__ __forceinline__ int64_t row_of(const tbx_stage_t& s, int64_t g) {
  if (s.flags & TBX_F_ROW_DIV) return g / s.div;
  if (s.flags & TBX_F_ROW_MOD) return g % s.div;
  if (s.flags & TBX_F_ROW_BATCH_MOD) return (g / s.k) * s.div + g % s.div;
  if (s.flags & TBX_F_ROW_IDX) return (int64_t)((const int32_t*)s.p1)[g];
  return g;
}

template <int MT, bool EXT>
__device__ void op_load(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = 16 * MT;
  float* dst = t.b(s.dst) + s.dst_col;
  const int ld = t.l(s.dst);
  const float* src = (const float*)s.p0;
  const int width = (s.flags & TBX_F_ROW_BATCH_MOD) ? s.n : (s.k > s.n ? s.k : s.n);
  const bool accum = (s.flags & TBX_F_ACCUM) != 0;
  // fast path: whole float4s, 16-byte aligned on both sides (the common case: 128 / 640 wide activations)
  if (src != nullptr && !accum && width == s.n && (s.n & 3) == 0 && (s.ld & 3) == 0 && (s.dst_col & 3) == 0 &&
      ((((uintptr_t)src) & 15) == 0)) {
    const int w4 = s.n >> 2;
    for (int e = threadIdx.x; e < ROWS * w4; e += blockDim.x) {
      const int r = e / w4, c4 = e - r * w4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < t.n_valid) v = *(const float4*)(src + row_of(s, t.g0 + r) * (int64_t)s.ld + c4 * 4);
      *(float4*)(dst + r * ld + c4 * 4) = v;
    }
    return;
  }
  for (int e = threadIdx.x; e < ROWS * width; e += blockDim.x) {
    const int r = e / width, c = e - r * width;
    float v = 0.f;
    if (r < t.n_valid && c < s.n && src != nullptr) v = src[row_of(s, t.g0 + r) * (int64_t)s.ld + c];
    if (accum && c < s.n)
      dst[r * ld + c] += v;
    else if (!accum)
      dst[r * ld + c] = v;
  }
}

constexpr int CH = 8;  // k-blocks (of 16) whose weight fragments are in flight per wave

// LINEAR (optionally grouped: `reserved` = G groups, group g reads src columns src_col + g*src_stride, writes
// dst_col + g*dst_stride, with src_stride / dst_stride packed in `div` as (src << 16 | dst); its weight block is the next
// n rows (or k rows if TBX_F_WT) after the previous group's, its bias the next n entries).
template <int MT, bool EXT>
__device__ void op_linear(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwave = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const float* src0 = t.b(s.src) + s.src_col;
  const bool to_global = EXT && s.dst == TBX_BUF_GLOBAL;  // output tile written straight to global memory (never accumulates)
  // LDS and global destinations are kept in separate variables: one pointer that may be either would turn every LDS
  // access of the epilogue into a FLAT access (measured 4x slower chains)
  float* dst0 = t.b(to_global ? 0 : s.dst) + s.dst_col;
  float* __restrict__ gout0 = to_global ? (float*)s.p2 + t.g0 * (int64_t)s.ld2 + s.dst_col : nullptr;
  const int lds_s = t.l(s.src), lds_d = t.l(to_global ? 0 : s.dst);
  const float* __restrict__ W0 = (const float*)s.p0;
  const float* __restrict__ bias0 = (const float*)s.p1;
  const int K = s.k, N = s.n, ldw = s.ld;
  const int G = s.reserved > 0 ? s.reserved : 1;
  const int gs_src = (s.div >> 16) & 0xffff, gs_dst = s.div & 0xffff;
  const bool wt = (s.flags & TBX_F_WT) != 0;
  const bool accum = (s.flags & TBX_F_ACCUM) != 0;
  const bool fast = (K % 16 == 0) && (wt || ((ldw % 4 == 0) && ((((uintptr_t)W0) & 15) == 0)));
  const int n_tiles = (N + 15) / 16;
  const int kblocks = (K + 15) / 16;
  const int tiles_total = G * n_tiles;

  // TBX_F_WPACK: p0 holds the MFMA B fragments in load order (tbx_pack_weight): every wave-wide float4 load is one
  // contiguous 1 KiB. A row-major [n,k] weight makes the 16 lanes of a fragment row read 16 different 512-byte rows,
  // i.e. 64 separate 16-byte requests per load - measured 2.6 us of a 5.4 us 128x128 stage, L2-resident or not.
  const bool packed = (s.flags & TBX_F_WPACK) != 0;
  const bool fast_wt = !packed && fast && wt && kblocks <= 4;
  const bool fast_n = packed || (fast && !wt);
  const int kstride = packed ? 256 : 16;
  // pointer (already offset to this lane's float4) to the first k-block of tile ti; tiles past the end and, in the
  // row-major layout, columns past n are clamped to valid memory (their products are zeroed at use / never stored)
  auto wrow_of = [&](int ti) -> const float* {
    ti = ti < tiles_total ? ti : tiles_total - 1;
    if (packed) return W0 + ((int64_t)ti * kblocks) * 256 + lane * 4;
    const int grp = ti / n_tiles;
    int col = (ti - grp * n_tiles) * 16 + j;
    col = col < N ? col : N - 1;
    return W0 + ((int64_t)grp * N + col) * ldw + g * 4;
  };

  float4 cur[CH], nxt[CH];
  const float* wrow = nullptr;
  if (fast_n) {
    wrow = wrow_of(wave);
#pragma unroll
    for (int q = 0; q < CH; ++q) cur[q] = (q < kblocks) ? *(const float4*)(wrow + q * kstride) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int ti = wave; ti < tiles_total; ti += nwave) {
    const int grp = ti / n_tiles;
    const int n0 = (ti - grp * n_tiles) * 16;
    const int col = n0 + j;
    const bool col_ok = col < N;
    const float* src = src0 + grp * gs_src;
    float* dst = dst0 + grp * gs_dst;
    float* gout = gout0 + grp * gs_dst;
    const float* bias = bias0 != nullptr ? bias0 + grp * N : nullptr;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float b = (bias != nullptr && col_ok) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float c0 = b;
        if (accum && col_ok) c0 += dst[(m * 16 + g * 4 + r) * lds_d + col];
        acc[m][r] = c0;
      }
    }
    if (fast_n) {
      // weights go straight to VGPRs, CH k-blocks ahead and across tile boundaries: the L2/HBM latency is paid once
      // per chunk and overlaps the previous chunk's MFMAs
      const float* wnext = wrow_of(ti + nwave);
      for (int c0 = 0; c0 < kblocks; c0 += CH) {
        const bool last = c0 + CH >= kblocks;
        const float* pn = last ? wnext : wrow + (c0 + CH) * kstride;
        const int kb_left = last ? kblocks : kblocks - (c0 + CH);
#pragma unroll
        for (int q = 0; q < CH; ++q) nxt[q] = (q < kb_left) ? *(const float4*)(pn + q * kstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          if (c0 + q < kblocks) {
            const int k0 = (c0 + q) * 16 + g * 4;
            float4 bv = cur[q];
            if (!col_ok) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float4 av = *(const float4*)(src + (m * 16 + j) * lds_s + k0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[m], 0, 0, 0);
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[m], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) cur[q] = nxt[q];
      }
      wrow = wnext;
    } else if (fast_wt) {
      // [k, n] layout with a short K (the per-head rpe fold, K = 32): every k-block's 4 rows are fetched up front
      const float* w = W0 + ((int64_t)grp * K + g * 4) * ldw + (col_ok ? col : 0);
      float4 bw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bw[q] = (q < kblocks) ? make_float4(w[(q * 16) * ldw], w[(q * 16 + 1) * ldw], w[(q * 16 + 2) * ldw], w[(q * 16 + 3) * ldw])
                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < kblocks) {
          const int k0 = q * 16 + g * 4;
          float4 bv = bw[q];
          if (!col_ok) bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float4 av = *(const float4*)(src + (m * 16 + j) * lds_s + k0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[m], 0, 0, 0);
          }
        }
      }
    } else {
      const float* W = W0 + (int64_t)grp * (wt ? K : N) * ldw;
      for (int kb = 0; kb < kblocks; ++kb) {
        const int k0 = kb * 16 + g * 4;
        float tmp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kk = k0 + q;
          float w = 0.f;
          if (col_ok && kk < K) w = wt ? W[(int64_t)kk * ldw + col] : W[(int64_t)col * ldw + kk];
          tmp[q] = w;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4 av = *(const float4*)(src + (m * 16 + j) * lds_s + k0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, tmp[0], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, tmp[1], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, tmp[2], acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, tmp[3], acc[m], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][r];
        if (s.act == TBX_ACT_RELU) v = fmaxf(v, 0.f);
        // columns of the last partial tile beyond n are zero-filled (unless accumulating or grouped) so that the next
        // stage may read a K padded to 16
        if (to_global) {
          if (col_ok && m * 16 + g * 4 + r < t.n_valid) gout[(int64_t)(m * 16 + g * 4 + r) * s.ld2 + col] = v;
        } else if (col_ok)
          dst[(m * 16 + g * 4 + r) * lds_d + col] = v;
        else if (!accum && G == 1 && col < lds_d - s.dst_col)
          dst[(m * 16 + g * 4 + r) * lds_d + col] = 0.f;
      }
    }
  }
}

template <int MT, bool EXT>
__device__ void op_layernorm(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = 16 * MT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const float* src = t.b(s.src) + s.src_col;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_s = t.l(s.src), lds_d = t.l(s.dst);
  const float* gamma = (const float*)s.p0;
  const float* beta = (const float*)s.p1;
  const int n = s.n;
  for (int r = wave; r < ROWS; r += nwave) {
    float v[8];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = lane + 64 * q;
      v[q] = (c < n) ? src[r * lds_s + c] : 0.f;
      sum += v[q];
    }
    sum = tbx::wave_sum(sum);
    const float mean = sum / (float)n;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = lane + 64 * q;
      const float d = (c < n) ? v[q] - mean : 0.f;
      var += d * d;
    }
    var = tbx::wave_sum(var) / (float)n;
    const float rstd = 1.0f / sqrtf(var + s.f0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = lane + 64 * q;
      if (c < n) dst[r * lds_d + c] = (v[q] - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

template <int MT, bool EXT>
__device__ void op_elementwise(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = 16 * MT;
  const float* src = t.b(s.src) + s.src_col;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_s = t.l(s.src), lds_d = t.l(s.dst);
  const int n = s.n;
  for (int e = threadIdx.x; e < ROWS * n; e += blockDim.x) {
    const int r = e / n, c = e - r * n;
    if (s.op == TBX_OP_ADD)
      dst[r * lds_d + c] += src[r * lds_s + c];
    else if (s.op == TBX_OP_COPY)
      dst[r * lds_d + c] = src[r * lds_s + c];
    else  // CLAMP
      dst[r * lds_d + c] = fminf(fmaxf(dst[r * lds_d + c], s.f0), s.f1);
  }
}

template <int MT, bool EXT>
__device__ void op_rowmask(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = 16 * MT;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_d = t.l(s.dst);
  const uint8_t* mask = (const uint8_t*)s.p0;
  const int n = s.n;
  for (int e = threadIdx.x; e < ROWS * n; e += blockDim.x) {
    const int r = e / n, c = e - r * n;
    bool m = r >= t.n_valid;
    if (!m && mask != nullptr) m = mask[row_of(s, t.g0 + r)] != 0;
    if (m) dst[r * lds_d + c] = s.f0;
  }
}

template <int MT, bool EXT>
__device__ void op_groupmax(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = 16 * MT;
  const float* src = t.b(s.src) + s.src_col;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_s = t.l(s.src), lds_d = t.l(s.dst);
  for (int c = threadIdx.x; c < s.n; c += blockDim.x) {
    float m = -INFINITY;
    for (int r = 0; r < ROWS; ++r) m = fmaxf(m, src[r * lds_s + c]);
    for (int r = 0; r < ROWS; ++r) dst[r * lds_d + c] = m;
  }
}

template <int MT, bool EXT>
__device__ void op_poolmax(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  const float* src = t.b(s.src) + s.src_col;
  const int lds_s = t.l(s.src);
  const uint8_t* mask = (const uint8_t*)s.p1;
  float* out = (float*)s.p0;
  for (int c = threadIdx.x; c < s.n; c += blockDim.x) {
    float m = -INFINITY;
    bool any = false;
    for (int r = 0; r < t.n_valid; ++r) {
      if (mask != nullptr && mask[t.g0 + r] != 0) continue;
      any = true;
      m = fmaxf(m, src[r * lds_s + c]);
    }
    out[t.group * (int64_t)s.ld + s.dst_col + c] = any ? m : 0.f;
  }
}

template <int MT, bool EXT>
__device__ void op_store(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = 16 * MT;
  const float* src = t.b(s.src) + s.src_col;
  const int lds_s = t.l(s.src);
  float* out = (float*)s.p0;
  const int n = s.n;
  if ((n & 3) == 0 && (s.ld & 3) == 0 && (s.dst_col & 3) == 0 && (s.src_col & 3) == 0 && ((((uintptr_t)out) & 15) == 0)) {
    const int w4 = n >> 2;
    for (int e = threadIdx.x; e < ROWS * w4; e += blockDim.x) {
      const int r = e / w4, c4 = e - r * w4;
      if (r < t.n_valid) *(float4*)(out + (t.g0 + r) * (int64_t)s.ld + s.dst_col + c4 * 4) = *(const float4*)(src + r * lds_s + c4 * 4);
    }
    return;
  }
  for (int e = threadIdx.x; e < ROWS * n; e += blockDim.x) {
    const int r = e / n, c = e - r * n;
    if (r < t.n_valid) out[(t.g0 + r) * (int64_t)s.ld + s.dst_col + c] = src[r * lds_s + c];
  }
}

#ifdef TBX_STAGE_CLOCK
// Profiling build only (libtbx_hip_clk.so, tools/stage_clock.py): workgroup 0 of every launch stamps the 100 MHz wall clock
// at each stage boundary. Never compiled into libtbx_hip.so.
constexpr int CLK_LAUNCHES = 2048, CLK_SLOTS = TBX_MAX_STAGES + 4;
__device__ unsigned long long g_clk[CLK_LAUNCHES * CLK_SLOTS];
__device__ unsigned int g_clk_launch;
#define TBX_CLK(i)                                                                          \
  do {                                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk_slot < (unsigned)CLK_LAUNCHES)           \
      g_clk[clk_slot * CLK_SLOTS + (i)] = wall_clock64();                                   \
  } while (0)
#else
#define TBX_CLK(i)
#endif

template <int MT, bool EXT>
__global__ __launch_bounds__(MT == 1 ? 1024 : 512) void rowchain_kernel(const RowchainArgs a) {
  constexpr int ROWS = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef TBX_STAGE_CLOCK
  unsigned clk_slot = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk_slot = atomicAdd(&g_clk_launch, 1u);
    if (clk_slot < (unsigned)CLK_LAUNCHES) g_clk[clk_slot * CLK_SLOTS + CLK_SLOTS - 1] = ((unsigned long long)gridDim.x << 32) | a.n_stages;
  }
  TBX_CLK(0);
  if (blockIdx.x == 0 && threadIdx.x == 0 && clk_slot < (unsigned)CLK_LAUNCHES) g_clk[clk_slot * CLK_SLOTS + CLK_SLOTS - 3] = clock64();
#endif
  Tile<MT, EXT> t;
  t.base = lds;
  t.ldw0 = a.ldw0;
  t.ldw1 = a.ldw1;
  t.ld_aux = a.ld_aux;
  t.group = blockIdx.x;
  if (a.group_rows > 0) {
    t.g0 = (int64_t)blockIdx.x * a.group_rows;
    t.n_valid = a.group_rows;
  } else {
    t.g0 = (int64_t)blockIdx.x * ROWS;
    const int64_t left = a.n_rows - t.g0;
    t.n_valid = left < ROWS ? (int)left : ROWS;
  }
  // The program is copied kernarg -> LDS once (one parallel read) and each stage descriptor is decoded from LDS into
  // SGPRs (readfirstlane): a scalar load per stage from the kernarg segment costs ~0.6 us of exposed latency per stage.
  constexpr int SW = (int)(sizeof(tbx_stage_t) / 4);
  uint32_t* prog = (uint32_t*)(lds + (size_t)ROWS * (a.ldw0 + a.ldw1 + a.ld_aux));
  {
    const uint32_t* ka = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int e = threadIdx.x; e < a.n_stages * SW; e += blockDim.x) prog[e] = ka[e];
  }
  __syncthreads();
  for (int i = 0; i < a.n_stages; ++i) {
    const uint32_t* ps = prog + i * SW;
    auto rd = [&](int w) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)ps[w]); };
    auto rdp = [&](int w) -> const void* { return (const void*)(((uint64_t)rd(w + 1) << 32) | rd(w)); };
    tbx_stage_t s;
    s.op = (int32_t)rd(0), s.src = (int32_t)rd(1), s.dst = (int32_t)rd(2), s.src_col = (int32_t)rd(3);
    s.dst_col = (int32_t)rd(4), s.k = (int32_t)rd(5), s.n = (int32_t)rd(6), s.act = (int32_t)rd(7);
    s.flags = (int32_t)rd(8), s.ld = (int32_t)rd(9), s.div = (int32_t)rd(10), s.reserved = (int32_t)rd(11);
    s.ld2 = (int32_t)rd(12), s.pad = 0;
    s.f0 = __uint_as_float(rd(14)), s.f1 = __uint_as_float(rd(15));
    s.p0 = rdp(16), s.p1 = rdp(18), s.p2 = rdp(20);
    switch (s.op) {
      case TBX_OP_LOAD: op_load<MT, EXT>(s, t); break;
      case TBX_OP_LINEAR: op_linear<MT, EXT>(s, t); break;
      case TBX_OP_LAYERNORM: op_layernorm<MT, EXT>(s, t); break;
      case TBX_OP_ADD:
      case TBX_OP_COPY:
      case TBX_OP_CLAMP: op_elementwise<MT, EXT>(s, t); break;
      case TBX_OP_ROWMASK: op_rowmask<MT, EXT>(s, t); break;
      case TBX_OP_GROUPMAX: op_groupmax<MT, EXT>(s, t); break;
      case TBX_OP_POOLMAX: op_poolmax<MT, EXT>(s, t); break;
      case TBX_OP_STORE: op_store<MT, EXT>(s, t); break;
      default: break;
    }
    __syncthreads();
    TBX_CLK(i + 1);
  }
#ifdef TBX_STAGE_CLOCK
  if (blockIdx.x == 0 && threadIdx.x == 0 && clk_slot < (unsigned)CLK_LAUNCHES) g_clk[clk_slot * CLK_SLOTS + CLK_SLOTS - 2] = clock64();
#endif
}

// out[(((grp*n_tiles + tile)*kblocks + kb)*64 + lane)*4 + t] = W_grp[col = tile*16 + (lane & 15)][kk = kb*16 + (lane >> 4)*4 + t]
__global__ void pack_weight_kernel(const float* __restrict__ w, int n, int k, int ld, int groups, int wt, float* __restrict__ out,
                                   int64_t total) {
  const int n_tiles = (n + 15) / 16, kblocks = (k + 15) / 16;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e & 3), lane = (int)((e >> 2) & 63);
    int64_t r = e >> 8;
    const int kb = (int)(r % kblocks);
    r /= kblocks;
    const int tile = (int)(r % n_tiles), grp = (int)(r / n_tiles);
    const int col = tile * 16 + (lane & 15), kk = kb * 16 + (lane >> 4) * 4 + t;
    float v = 0.f;
    if (col < n && kk < k) v = wt ? w[((int64_t)grp * k + kk) * ld + col] : w[((int64_t)grp * n + col) * ld + kk];
    out[e] = v;
  }
}

int check_stage(const tbx_stage_t& s, int ldw0, int ldw1, int ld_aux, int tile_rows) {
  auto buf_ld = [&](int b) { return b == 0 ? ldw0 : (b == 1 ? ldw1 : (b == 2 ? ld_aux : (1 << 30))); };
  if (s.op < TBX_OP_LOAD || s.op > TBX_OP_CLAMP) return TBX_ERR_ARG;
  const bool gdst = s.op == TBX_OP_LINEAR && s.dst == TBX_BUF_GLOBAL;
  if (s.src < 0 || s.src > 2 || s.dst < 0 || (s.dst > 2 && !gdst)) return TBX_ERR_ARG;
  if (gdst && (s.p2 == nullptr || s.ld2 <= 0 || (s.flags & TBX_F_ACCUM))) return TBX_ERR_ARG;
  if (s.n <= 0) return TBX_ERR_ARG;
  const bool reads_src = s.op == TBX_OP_LINEAR || s.op == TBX_OP_LAYERNORM || s.op == TBX_OP_ADD || s.op == TBX_OP_COPY ||
                         s.op == TBX_OP_GROUPMAX || s.op == TBX_OP_POOLMAX || s.op == TBX_OP_STORE;
  const bool writes_dst = s.op != TBX_OP_POOLMAX && s.op != TBX_OP_STORE;
  if (writes_dst) {
    int w = s.n;
    if (s.op == TBX_OP_LOAD && !(s.flags & TBX_F_ROW_BATCH_MOD) && s.k > w) w = s.k;
    if (s.dst_col < 0 || s.dst_col + w > buf_ld(s.dst)) return TBX_ERR_UNSUPPORTED;
  }
  if (reads_src) {
    const int w = s.op == TBX_OP_LINEAR ? ((s.k + 15) / 16) * 16 : s.n;
    if (s.src_col < 0 || s.src_col + w > buf_ld(s.src)) return TBX_ERR_UNSUPPORTED;
  }
  if (s.op == TBX_OP_LINEAR) {
    if (s.k <= 0 || s.p0 == nullptr || s.ld <= 0) return TBX_ERR_ARG;
    const int G = s.reserved > 0 ? s.reserved : 1;
    const int gs_src = (s.div >> 16) & 0xffff, gs_dst = s.div & 0xffff;
    if (s.src_col % 4 != 0 || (G > 1 && gs_src % 4 != 0)) return TBX_ERR_ALIGN;
    const int kw = ((s.k + 15) / 16) * 16;
    const int nw = ((s.n + 15) / 16) * 16;
    const int src_hi = s.src_col + (G - 1) * gs_src + kw, dst_hi = s.dst_col + (G - 1) * gs_dst + (G > 1 ? s.n : nw);
    if (G > 1 && (s.n % 16 != 0)) return TBX_ERR_UNSUPPORTED;
    if (src_hi > buf_ld(s.src) || s.dst_col + (G - 1) * gs_dst + s.n > buf_ld(s.dst)) return TBX_ERR_UNSUPPORTED;
    if (s.src == s.dst && !(s.dst_col >= src_hi || s.src_col >= dst_hi)) return TBX_ERR_UNSUPPORTED;  // in place: disjoint only
  }
  if (s.op == TBX_OP_LAYERNORM && (s.n > 512 || s.p0 == nullptr || s.p1 == nullptr)) return TBX_ERR_UNSUPPORTED;
  if ((s.op == TBX_OP_POOLMAX || s.op == TBX_OP_STORE) && (s.p0 == nullptr || s.ld <= 0)) return TBX_ERR_ARG;
  if (s.op != TBX_OP_LINEAR && (s.flags & (TBX_F_ROW_DIV | TBX_F_ROW_MOD | TBX_F_ROW_BATCH_MOD)) && s.div <= 0) return TBX_ERR_ARG;
  if ((s.flags & TBX_F_ROW_BATCH_MOD) && s.k <= 0) return TBX_ERR_ARG;
  (void)tile_rows;
  return TBX_OK;
}

}  // namespace

extern "C" int tbx_rowchain_ex(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                               int ldw0, int ldw1, int ld_aux, void* stream);

#ifdef TBX_STAGE_CLOCK
extern "C" int tbx_debug_clock_reset() {
  unsigned int z = 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_clk_launch), &z, sizeof(z)) == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
// host_out: [max_launches][TBX_MAX_STAGES + 4] u64; returns the number of launches recorded (or a negative error)
extern "C" int tbx_debug_clock_dump(unsigned long long* host_out, int max_launches) {
  unsigned int n = 0;
  if (hipDeviceSynchronize() != hipSuccess) return TBX_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_clk_launch), sizeof(n)) != hipSuccess) return TBX_ERR_LAUNCH;
  int m = (int)n < max_launches ? (int)n : max_launches;
  m = m < CLK_LAUNCHES ? m : CLK_LAUNCHES;
  if (m > 0 && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clk), (size_t)m * CLK_SLOTS * sizeof(unsigned long long)) != hipSuccess)
    return TBX_ERR_LAUNCH;
  return m;
}
#endif

extern "C" int64_t tbx_pack_weight_size(int n, int k, int groups) {
  if (n <= 0 || k <= 0 || groups <= 0) return TBX_ERR_ARG;
  return (int64_t)groups * ((n + 15) / 16) * ((k + 15) / 16) * 256;
}

extern "C" int tbx_pack_weight(const float* w, int n, int k, int ld, int groups, int wt, float* out, void* stream) {
  if (w == nullptr || out == nullptr || n <= 0 || k <= 0 || ld <= 0 || groups <= 0) return TBX_ERR_ARG;
  if (groups > 1 && n % 16 != 0) return TBX_ERR_UNSUPPORTED;
  const int64_t total = tbx_pack_weight_size(n, k, groups);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, n, k, ld, groups, wt, out, total);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_rowchain(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                            int ldw, void* stream) {
  return tbx_rowchain_ex(stages, n_stages, n_rows, group_rows, tile_rows, ldw, ldw, TBX_AUX_LD, stream);
}

extern "C" int tbx_rowchain_ex(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                               int ldw0, int ldw1, int ld_aux, void* stream) {
  if (stages == nullptr || n_stages <= 0 || n_rows <= 0) return TBX_ERR_ARG;
  if (n_stages > TBX_MAX_STAGES) return TBX_ERR_UNSUPPORTED;
  if (tile_rows != 16 && tile_rows != 32) return TBX_ERR_UNSUPPORTED;
  if (ldw0 <= 0 || ldw1 <= 0 || ld_aux <= 0 || ldw0 % 4 != 0 || ldw1 % 4 != 0 || ld_aux % 4 != 0) return TBX_ERR_ALIGN;
  if (group_rows < 0 || group_rows > tile_rows) return TBX_ERR_UNSUPPORTED;
  if (group_rows > 0 && n_rows % group_rows != 0) return TBX_ERR_ARG;
  // activation buffers + the program itself (decoded from LDS by the kernel)
  const size_t lds_bytes = (size_t)(ldw0 + ldw1 + ld_aux) * tile_rows * sizeof(float) + (size_t)n_stages * sizeof(tbx_stage_t);
  if (lds_bytes > 160 * 1024) return TBX_ERR_UNSUPPORTED;
  RowchainArgs a;
  for (int i = 0; i < n_stages; ++i) {
    const int rc = check_stage(stages[i], ldw0, ldw1, ld_aux, tile_rows);
    if (rc != TBX_OK) return rc;
    a.st[i] = stages[i];
  }
  a.n_stages = n_stages;
  a.group_rows = group_rows;
  a.ldw0 = ldw0;
  a.ldw1 = ldw1;
  a.ld_aux = ld_aux;
  a.n_rows = n_rows;
  const int64_t n_tiles = group_rows > 0 ? n_rows / group_rows : (n_rows + tile_rows - 1) / tile_rows;
  hipStream_t s = (hipStream_t)stream;
  bool ext = !(ldw0 == ldw1 && ld_aux == TBX_AUX_LD);
  for (int i = 0; i < n_stages; ++i) ext = ext || (stages[i].op == TBX_OP_LINEAR && stages[i].dst == TBX_BUF_GLOBAL);
#define TBX_RC_LAUNCH(MT, EXTF, NT)                                                                                        \
  do {                                                                                                                     \
    if (lds_bytes > 64 * 1024)                                                                                             \
      (void)hipFuncSetAttribute((const void*)rowchain_kernel<MT, EXTF>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                (int)lds_bytes);                                                                           \
    hipLaunchKernelGGL((rowchain_kernel<MT, EXTF>), dim3((unsigned)n_tiles), dim3(NT), lds_bytes, s, a);                   \
  } while (0)
  if (tile_rows == 16) {
    if (ext)
      TBX_RC_LAUNCH(1, true, 1024);
    else
      TBX_RC_LAUNCH(1, false, 1024);
  } else {
    if (ext)
      TBX_RC_LAUNCH(2, true, 512);
    else
      TBX_RC_LAUNCH(2, false, 512);
  }
#undef TBX_RC_LAUNCH
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
