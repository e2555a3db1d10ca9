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
  int32_t first_packed;  // first TBX_F_WPACK LINEAR stage (-1: none): its weights are requested before stage 0 runs
  int32_t live_rows;     // tbx_rowchain_live: a tile holds this many rows (1, 2 or 4) of a 16-row LDS tile; 0: whole tiles
  int64_t n_rows;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// Pointers decoded from the LDS-resident program are generic to the compiler; loads through them would be FLAT
// instructions, which also count against lgkmcnt and so serialise with every LDS wait. These casts state what the ABI
// guarantees: stage pointers are device global memory.
#define TBX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 gld4(const float* p) {
  const f32x4 v = *(const TBX_GLOBAL f32x4*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
// uniform base (SGPR pair) + 32-bit per-lane byte offset: the saddr addressing form, no 64-bit VGPR address per load
__device__ __forceinline__ float4 gld4(const float* sbase, uint32_t lane_bytes) {
  const f32x4 v = *(const TBX_GLOBAL f32x4*)((const TBX_GLOBAL char*)sbase + lane_bytes);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float gld1(const float* sbase, uint32_t lane_bytes) {
  return *(const TBX_GLOBAL float*)((const TBX_GLOBAL char*)sbase + lane_bytes);
}
__device__ __forceinline__ float gld1(const float* p) { return *(const TBX_GLOBAL float*)p; }
__device__ __forceinline__ uint8_t gld1(const uint8_t* p) { return *(const TBX_GLOBAL uint8_t*)p; }
__device__ __forceinline__ int32_t gld1(const int32_t* p) { return *(const TBX_GLOBAL int32_t*)p; }
__device__ __forceinline__ void gst4(float* p, float4 v) {
  f32x4 x;
  x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
  *(TBX_GLOBAL f32x4*)p = x;
}
__device__ __forceinline__ void gst1(float* p, float v) { *(TBX_GLOBAL float*)p = v; }
// TBX_F_OUT_BF16 destinations: element e of a bfloat16 buffer (round to nearest even)
__device__ __forceinline__ uint16_t to_bf16(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ void gst1_bf16(float* base, int64_t e, float v) { *((TBX_GLOBAL uint16_t*)base + e) = to_bf16(v); }

template <int MT, bool EXT>
struct Tile {
  static constexpr int ROWS = MT > 0 ? 16 * MT : 4;  // MT = 0: the 4-row tiles of tbx_rowchain_live
  float* base;    // LDS base; BUF0 = ROWS*ldw0 floats, then BUF1 = ROWS*ldw1, then AUX = ROWS*ld_aux
  int ldw0, ldw1, ld_aux;
  // computed, not indexed: a runtime-indexed member array would live in scratch memory - and so does the whole struct when
  // the choice is written as nested ?: on the members (the optimiser turns it into a load through a selected ADDRESS, which
  // pins the struct in scratch: a scratch round trip per stage in every EXT kernel). Mask arithmetic on the three values keeps
  // them in SGPRs (ScratchSize 72 -> 0 in the ISA of all EXT variants but the two that spill VGPRs).
  // EXT = per-buffer widths + LINEAR-to-global (tbx_rowchain_ex); the plain layout (two ldw0-wide buffers + a 260-wide
  // auxiliary) keeps the address arithmetic of the common small-grid programs minimal (measured: 7 % on a C2 step)
  __device__ __forceinline__ float* b(int i) const {
    if constexpr (EXT) return base + ROWS * ((ldw0 & -(int)(i >= 1)) + (ldw1 & -(int)(i >= 2)));
    return base + (i == TBX_BUF_AUX ? 2 : i) * ROWS * ldw0;
  }
  __device__ __forceinline__ int l(int i) const {
    if constexpr (EXT) {
      const int m1 = -(int)(i == 1), m2 = -(int)(i >= 2);
      return (ldw0 & ~(m1 | m2)) | (ldw1 & m1) | (ld_aux & m2);
    }
    return i == TBX_BUF_AUX ? TBX_AUX_LD : ldw0;
  }
  int64_t g0;      // first global row of the tile
  int n_valid;     // rows r < n_valid map to a global row
  int64_t group;   // index of the tile's first group (grouped mode) or tile index
  int gw, ng;      // grouped mode: rows per group and groups in this tile (rows [j*gw, (j+1)*gw) = group `group + j`);
                   // flat mode: gw = ROWS, ng = 1
  int rows_live;   // rows the row-wise ops walk: ROWS, or the live rows of a tbx_rowchain_live tile (the rest is padding)
};

__device__ __forceinline__ int64_t row_of(const tbx_stage_t& s, int64_t g) {
  if (s.flags & TBX_F_ROW_DIV) return g / s.div;
  if (s.flags & TBX_F_ROW_MOD) return g % s.div;
  if (s.flags & TBX_F_ROW_BATCH_MOD) return (g / s.k) * s.div + g % s.div;
  if (s.flags & TBX_F_ROW_IDX) return (int64_t)gld1((const int32_t*)s.p1 + g);
  return g;
}

// The row-wise ops below give every wavefront whole rows (row = wave, wave + nwave, ...; lanes walk the columns): with 16
// wavefronts sharing one CU a VALU instruction costs 16 issue cycles per workgroup, and a per-element e / n, e % n index
// was most of a trivial stage's time.
#define TBX_WAVE_ROWS                                                            \
  const int lane = threadIdx.x & 63;                                             \
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      \
  const int nwave = (int)(blockDim.x >> 6)

template <int MT, bool EXT>
__device__ void op_load(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = Tile<MT, EXT>::ROWS;
  TBX_WAVE_ROWS;
  float* dst = t.b(s.dst) + s.dst_col;
  const int ld = t.l(s.dst);
  const float* src = (const float*)s.p0;
  const int width = (s.flags & TBX_F_ROW_BATCH_MOD) ? s.n : (s.k > s.n ? s.k : s.n);
  const bool accum = (s.flags & TBX_F_ACCUM) != 0;
  // fast path: whole float4s, 16-byte aligned on both sides (the common case: 128 / 640 wide activations)
  if (src != nullptr && !accum && width == s.n && (s.n & 3) == 0 && (s.ld & 3) == 0 && (s.dst_col & 3) == 0 &&
      ((((uintptr_t)src) & 15) == 0)) {
    const int w4 = s.n >> 2;
    // TBX_F_LOAD2: a second source rides in the same round trip (the token row x beside the attention output)
    const bool two = (s.flags & TBX_F_LOAD2) != 0;
    const float* src2 = (const float*)s.p2;
    float* dst2 = t.b(s.src) + s.src_col;
    const int ldb = t.l(s.src), w4b = two ? (s.reserved >> 2) : 0;
    // all of a wave's requests first, then the LDS writes: the rows were written by the previous launch and come from L2 /
    // HBM at ~1 us per dependent round trip; a 640-wide load was 6 of them in a row (2.7 us -> one round trip)
    constexpr int RPW = ROWS >= 8 ? ROWS / 8 : 1;  // rows per wave at the 8-wave workgroups every launch uses
    if (ROWS >= 8 && RPW <= 4 && nwave == 8 && w4 <= 256) {  // (48-row tiles would hold 96 registers here: they take the loop below)
      float4 v[RPW][4], v2[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave + rr * 8;
        const bool live = r < t.n_valid;
        const int64_t grow = live ? row_of(s, t.g0 + r) : 0;
        const float* srow = src + grow * (int64_t)s.ld;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c4 = lane + 64 * it;
          v[rr][it] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (live && c4 < w4) v[rr][it] = gld4(srow + c4 * 4);
        }
        v2[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (two && live && lane < w4b) v2[rr] = gld4(src2 + grow * (int64_t)s.ld2 + lane * 4);
      }
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave + rr * 8;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c4 = lane + 64 * it;
          if (c4 < w4) *(float4*)(dst + r * ld + c4 * 4) = v[rr][it];
        }
        if (two && lane < w4b) *(float4*)(dst2 + r * ldb + lane * 4) = v2[rr];
      }
      return;
    }
    for (int r = wave; r < ROWS; r += nwave) {
      const bool live = r < t.n_valid;
      const int64_t grow = live ? row_of(s, t.g0 + r) : 0;
      const float* srow = src + grow * (int64_t)s.ld;
      for (int c4 = lane; c4 < w4; c4 += 64) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = gld4(srow + c4 * 4);
        *(float4*)(dst + r * ld + c4 * 4) = v;
      }
      if (two && lane < w4b) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = gld4(src2 + grow * (int64_t)s.ld2 + lane * 4);
        *(float4*)(dst2 + r * ldb + lane * 4) = v;
      }
    }
    return;
  }
  for (int r = wave; r < ROWS; r += nwave) {
    const bool live = r < t.n_valid && src != nullptr;
    const float* srow = src + (live ? row_of(s, t.g0 + r) : 0) * (int64_t)s.ld;
    for (int c = lane; c < width; c += 64) {
      float v = 0.f;
      if (live && c < s.n) v = gld1(srow + c);
      if (accum && c < s.n)
        dst[r * ld + c] += v;
      else if (!accum)
        dst[r * ld + c] = v;
    }
  }
}

constexpr int CH = 8;  // k-blocks (of 16) whose weight fragments are in flight per wave
constexpr int SW = (int)(sizeof(tbx_stage_t) / 4);  // stage descriptor, dwords (the program lives in LDS)

// Packed weight image (tbx_pack_weight): per 16-column tile, kblocks x (64 lanes x float4) B fragments followed by 64
// floats of bias (lane l holds bias[col(l & 15)]). One wave-wide load = one contiguous 1 KiB / 256 B.
__device__ __forceinline__ int wp_tile_stride(int kblocks) { return kblocks * 256 + 64; }

// Weight fragments a wave holds ahead of their use: the first CH k-blocks (+ bias) of its first tile of packed LINEAR
// stage `stage`. Filled at kernel start and by the previous packed LINEAR while it works on its last tile, so the
// L2/HBM latency of a stage's first weights overlaps the stages before it. The stage barrier therefore only orders LDS.
struct WeightAhead {
  float4 w[CH];
  float bias;
  int stage;  // -1: nothing held
};

__device__ __forceinline__ uint32_t prog_word(const uint32_t* prog, int stage, int w) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)prog[stage * SW + w]);
}

// shape of packed LINEAR stage j, read from the LDS program
struct PackedShape {
  const float* W;
  int kblocks, tiles_total;
};
__device__ __forceinline__ PackedShape packed_shape(const uint32_t* prog, int j) {
  PackedShape ps;
  const int k = (int)prog_word(prog, j, 5), n = (int)prog_word(prog, j, 6), G = (int)prog_word(prog, j, 11);
  ps.W = (const float*)(((uint64_t)prog_word(prog, j, 17) << 32) | prog_word(prog, j, 16));
  ps.kblocks = (k + 15) / 16;
  ps.tiles_total = (G > 0 ? G : 1) * ((n + 15) / 16);
  return ps;
}

__device__ __forceinline__ void weights_ahead(const uint32_t* prog, int j, int lane, int wave, WeightAhead& wa) {
  const PackedShape ps = packed_shape(prog, j);
  const int ti = wave < ps.tiles_total ? wave : ps.tiles_total - 1;
  const float* base = ps.W + (int64_t)ti * wp_tile_stride(ps.kblocks);
#pragma unroll
  for (int q = 0; q < CH; ++q)
    wa.w[q] = gld4(base + (q < ps.kblocks ? q : 0) * 256, (uint32_t)lane * 16u);
  wa.bias = gld1(base + ps.kblocks * 256, (uint32_t)lane * 4u);
  wa.stage = j;
}

// Packed-weight tiles of one LINEAR stage for one wave. `cur` (= the wave's WeightAhead registers) holds the CH k-blocks
// being multiplied, `nxt` the CH after them in the wave's stream: this tile's next chunk, then the wave's next tile, then
// its first tile of the next packed LINEAR stage (whose fragments are thus in flight across the stage barrier). Loads
// are unconditional with scalar-selected addresses (a load under a branch is waited for on the spot and degrades the
// vmcnt bookkeeping to vmcnt(0)); with nothing left to fetch they re-read the tile's first block, an L2 hit.
template <int MT, bool EXT, bool FULL>
__device__ __forceinline__ void linear_packed(const tbx_stage_t& s, const Tile<MT, EXT>& t, const uint32_t* prog, WeightAhead& wa,
                                              int lane, int wave, int nwave, bool to_global) {
  const int j = lane & 15, g = lane >> 4;
  const float* src0 = t.b(s.src) + s.src_col;
  float* dst0 = t.b(to_global ? 0 : s.dst) + s.dst_col;
  float* __restrict__ gout0 = to_global ? (float*)s.p2 + t.g0 * (int64_t)s.ld2 + s.dst_col : nullptr;
  const int lds_s = t.l(s.src), lds_d = t.l(to_global ? 0 : s.dst);
  const float* __restrict__ W0 = (const float*)s.p0;
  const int N = s.n;
  const int G = s.reserved > 0 ? s.reserved : 1;
  const int gs_src = (s.div >> 16) & 0xffff, gs_dst = s.div & 0xffff;
  const bool accum = (s.flags & TBX_F_ACCUM) != 0;
  const int n_tiles = (N + 15) / 16;
  const int kblocks = (s.k + 15) / 16;
  const int tiles_total = G * n_tiles;
  const int tstride = wp_tile_stride(kblocks);
  const int nj = s.pad;
  PackedShape nx;
  nx.W = W0, nx.kblocks = 0, nx.tiles_total = 1;
  if (nj > 0) nx = packed_shape(prog, nj);
  float4(&cur)[CH] = wa.w;
  float4 nxt[CH];
  float cb = wa.bias;
  const bool split = FULL && (s.flags & TBX_F_WSPLIT) != 0;  // the lean kernel carries no split-bf16 code
  const bool rowskip = (s.flags & TBX_F_ROWSKIP) != 0 && !to_global;
  for (int ti = wave; ti < tiles_total; ti += nwave) {
    const int grp = ti / n_tiles;
    const int n0 = (ti - grp * n_tiles) * 16;
    const int col = n0 + j;
    const bool col_ok = col < N;
    const float* src = src0 + grp * gs_src;
    float* dst = dst0 + grp * gs_dst;
    float* gout = gout0 + grp * gs_dst;
    const float* tbase = W0 + (int64_t)ti * tstride;
    // the tile after this one in the wave's stream
    const float* nbase = tbase;
    int kb_next = 0;
    if (ti + nwave < tiles_total) {
      nbase = tbase + (int64_t)nwave * tstride;
      kb_next = kblocks;
    } else if (nj > 0) {
      const int tn = wave < nx.tiles_total ? wave : nx.tiles_total - 1;
      nbase = nx.W + (int64_t)tn * wp_tile_stride(nx.kblocks);
      kb_next = nx.kblocks;
    }
    const float nb = gld1(nbase + (kb_next > 0 ? kb_next : kblocks) * 256, (uint32_t)lane * 4u);
    // TBX_F_ROWSKIP: bit (m * 4 + r) set = the lane's row m*16 + g*4 + r keeps its old content (requested now, used after the MFMAs)
    uint32_t skip = 0u;
    if (rowskip) {
      const uint8_t* mask = (const uint8_t*)s.p1;
      const bool inv = (s.flags & TBX_F_MASK_INV) != 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m * 16 + g * 4 + r;
          bool sk = row >= t.n_valid;
          if (!sk) sk = (gld1(mask + t.g0 + row) != 0) != inv;
          skip |= sk ? (1u << (m * 4 + r)) : 0u;
        }
    }
    f32x4 acc[MT], acc_x[MT], acc_y[MT];  // acc_x / acc_y: the two cross products of the split path
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float c0 = cb;
        if (accum && col_ok) c0 += dst[(m * 16 + g * 4 + r) * lds_d + col];
        acc[m][r] = c0;
        acc_x[m][r] = 0.f;
        acc_y[m][r] = 0.f;
      }
    }
    for (int c0 = 0; c0 < kblocks; c0 += CH) {
      const bool last = c0 + CH >= kblocks;
      const float* pn = last ? nbase : tbase + (c0 + CH) * 256;
      const int kb_left = last ? kb_next : kblocks - (c0 + CH);
#pragma unroll
      for (int q = 0; q < CH; ++q) nxt[q] = gld4(pn + (q < kb_left ? q : 0) * 256, (uint32_t)lane * 16u);
      if (FULL && split) {
        // three-product split-bf16 (TBX_F_WSPLIT): a = a_hi + a_lo, w = w_hi + w_lo in bf16 (RNE, ~2^-17 relative), the
        // products hi*hi + hi*lo + lo*hi on the 16x-faster v_mfma_f32_16x16x16_bf16 with fp32 accumulation (lo*lo, ~2^-18
        // of a product on average, is dropped). The A fragment of a k-block - the lane's four k values - is exactly the
        // bf16 MFMA's A operand, the packed image's 16 bytes exactly its B operands.
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          if (c0 + q < kblocks) {
            const int k0 = (c0 + q) * 16 + g * 4;
            const float4 bw = cur[q];
            const s16x4_t bh = __builtin_bit_cast(s16x4_t, make_float2(bw.x, bw.y));
            const s16x4_t bl = __builtin_bit_cast(s16x4_t, make_float2(bw.z, bw.w));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const float4 av = *(const float4*)(src + (m * 16 + j) * lds_s + k0);
              const f32x2_t a01 = {av.x, av.y}, a23 = {av.z, av.w};
              const bf16x2_t h01 = __builtin_convertvector(a01, bf16x2_t), h23 = __builtin_convertvector(a23, bf16x2_t);
              const f32x2_t r01 = a01 - __builtin_convertvector(h01, f32x2_t), r23 = a23 - __builtin_convertvector(h23, f32x2_t);
              const bf16x2_t l01 = __builtin_convertvector(r01, bf16x2_t), l23 = __builtin_convertvector(r23, bf16x2_t);
              struct P2 { bf16x2_t a, b; };
              const s16x4_t ah = __builtin_bit_cast(s16x4_t, (P2{h01, h23}));
              const s16x4_t al = __builtin_bit_cast(s16x4_t, (P2{l01, l23}));
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc[m], 0, 0, 0);
              acc_x[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc_x[m], 0, 0, 0);
              acc_y[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc_y[m], 0, 0, 0);
            }
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          if (c0 + q < kblocks) {
            const int k0 = (c0 + q) * 16 + g * 4;
            const float4 bv = cur[q];
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
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) cur[q] = nxt[q];
    }
    cb = nb;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][r] + (acc_x[m][r] + acc_y[m][r]);
        if (s.act == TBX_ACT_RELU) v = fmaxf(v, 0.f);
        // columns of the last partial tile beyond n are zero-filled (unless accumulating or grouped) so that the next
        // stage may read a K padded to 16
        if (to_global) {
          if (col_ok && m * 16 + g * 4 + r < t.n_valid) {
            if (s.flags & TBX_F_OUT_BF16)
              gst1_bf16((float*)s.p2, (t.g0 + m * 16 + g * 4 + r) * (int64_t)s.ld2 + s.dst_col + grp * gs_dst + col, v);
            else
              gst1(gout + (int64_t)(m * 16 + g * 4 + r) * s.ld2 + col, v);
          }
        } else if (col_ok) {
          if (!(skip & (1u << (m * 4 + r))))
            dst[(m * 16 + g * 4 + r) * lds_d + col] = v;
          else if (s.flags & TBX_F_ROWZERO)
            dst[(m * 16 + g * 4 + r) * lds_d + col] = 0.f;
        }
        else if (!accum && G == 1 && col < lds_d - s.dst_col)
          dst[(m * 16 + g * 4 + r) * lds_d + col] = 0.f;
      }
    }
  }
  wa.bias = cb;
}

// ---------------------------------------------------------------------------------------------------------------------
// TBX_F_WGEMV: LINEAR for the 1 / 2 / 4-row tiles of tbx_rowchain_live (the closed loop at a few scenes, where a launch has
// 64-128 rows in all and every stage is latency-, not throughput-bound). A 16-row MFMA tile would spend 3/4 .. 15/16 of the
// exact-fp32 matrix pipe (which runs at the VALU rate anyway) on padding rows and string 32 dependent 40-cycle MFMAs per 128 k.
// Here a THREAD owns one output element (row r = tid / 128, column cl = tid % 128 of the current 128-column block) and runs the
// k-ordered v_fma chain the 16x16x4 MFMA sequence is equivalent to - per k-block of 16: step t = 0..3, lane group g = 0..3,
// k = kb*16 + g*4 + t, starting from bias (+ destination) - so results are bit-identical to the packed MFMA path
// (cdna_hip_programming.md 3: an f32 MFMA is a k-ordered fmaf chain).
// Weights come as the tbx_pack_weight_gemv image: per 128-column block a row of (bias, 0, 0, 0) per column, then float4 rows
// q = kb*4 + t of [128 columns][4] = (W[c][kb*16 + t], W[c][kb*16+4 + t], W[c][kb*16+8 + t], W[c][kb*16+12 + t]). They stream
// through two 66 KiB LDS slots in chunks of <= 8 k-blocks x 128 columns by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction, no VGPRs, no ds_write pass): chunk i+1 - or chunk 0 of the NEXT LINEAR stage - is in flight while chunk i is
// multiplied; every thread then reads its column conflict-free (ds_read_b128, consecutive lanes = consecutive 16 B) and the row's
// activations as broadcast reads.
// Measured alternatives (tools/live_micro.py, 128x128 stage, 4 live rows, weights L2-hot; this form: 2.1 k cycles in the multiply
// loop): v_mfma_f32_4x4x1_16b_f32 (4 rows x 64 columns x one k per issue, no padding): 3.2 k - ~25 cycles per dependent issue;
// activations as SGPR operands through v_readlane instead of broadcast LDS reads: 5.2 k; the LDS reads two k-blocks deep in
// software: 2.4 k. At 8 waves the loop is LDS-issue bound (a broadcast ds_read_b128 still costs 4 LDS cycles per wave
// instruction), so 1- or 2-row tiles (2 / 4 multiplying waves) are the fastest per stage.
#ifdef TBX_STAGE_CLOCK
__device__ unsigned long long g_sub[8];  // shader-clock stamps inside the last linear_gemv call of workgroup 0 (tools/live_micro.py)
#define TBX_SUB(i)                                                       \
  do {                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_sub[i] = clock64();       \
  } while (0)
#else
#define TBX_SUB(i)
#endif
constexpr int GCOLS = 128;                   // columns per block = threads per live row
constexpr int GKC = 8;                       // k-blocks per chunk
constexpr int GROW = GCOLS * 4;              // floats per float4 row of a column block (2 KiB)
constexpr int GSLOT = (1 + GKC * 4) * GROW;  // floats per LDS weight slot: a bias row + 32 weight rows (66 KiB)

struct WeightStream {
  int stage;  // stage whose chunk 0 is (being) loaded into slot `slot`; -1: nothing in flight
  int slot;
};

__device__ __forceinline__ int gemv_kblocks(const uint32_t* prog, int j) { return ((int)prog_word(prog, j, 5) + 15) / 16; }

// LDS-DMA of chunk (cb, kc) of the image at `img` (kblocks k-blocks per column block; a column block = one row of [128][4] whose
// .x is the bias, then kblocks*4 weight rows) into `slot`: chunk 0 of a block brings the bias row along. Pieces of 1 KiB, piece p
// by wave p % nwave; completion = every issuing wave's vmcnt(0) + a workgroup barrier.
// Inline asm, not __builtin_amdgcn_global_load_lds: hipcc orders every later ds_read behind a DMA it knows about (s_waitcnt
// vmcnt(0) in front of the FMA loop's LDS reads - measured: no overlap at all, a stage = DMA latency + multiply time). The asm
// form is invisible to its counters (which can then only over-wait: VMEM returns in order); the LDS hazard is handled by hand:
// every consumer sits behind "s_waitcnt vmcnt(0); s_barrier" (cdna_hip_programming.md 5.7: M0 is written in the statement that
// reads it and restored).
// The waves that multiply (the first 2 * live rows: a wave is 64 columns of one row) issue NO pieces when other waves exist: a
// wave's later LDS reads wait behind its own DMA (measured in isolation, tools/scratch/l2_stream_probe.hip: a chunk's DMA and its
// multiply added up - 3074 cycles per 66 KiB chunk - instead of overlapping; 2212 with the pieces left to the idle waves).
__device__ __forceinline__ void gemv_dma(const float* img, int kblocks, int cb, int kc, float* slot, int lane, int wave, int nwave, int w0) {
  if (wave < w0) return;
  const int kbc = (kblocks - kc * GKC) < GKC ? (kblocks - kc * GKC) : GKC;
  const int rows = kbc * 4 + (kc == 0 ? 1 : 0);
  const float* base = img + ((int64_t)cb * (1 + kblocks * 4) + (kc == 0 ? 0 : 1 + kc * GKC * 4)) * GROW;
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)slot);
  for (int p = wave - w0; p < rows * 2; p += nwave - w0) {
    const float* gsrc = base + p * 256 + lane * 4;
    const uint32_t dst = lds0 + (uint32_t)p * 1024u;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  }
}

template <int MT, bool EXT>
__device__ __forceinline__ void linear_gemv(const tbx_stage_t& s, const Tile<MT, EXT>& t, const uint32_t* prog, float* wslots,
                                            const uint32_t* skipflags, WeightStream& ws, int stage) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nwave = (int)(blockDim.x >> 6);
  const bool to_global = EXT && s.dst == TBX_BUF_GLOBAL;
  const float* src0 = t.b(s.src) + s.src_col;
  float* dst0 = t.b(to_global ? 0 : s.dst) + s.dst_col;
  float* __restrict__ gout0 = to_global ? (float*)s.p2 + t.g0 * (int64_t)s.ld2 + s.dst_col : nullptr;
  const int lds_s = t.l(s.src), lds_d = t.l(to_global ? 0 : s.dst);
  const float* __restrict__ img = (const float*)s.p0;
  const int N = s.n;
  const int G = s.reserved > 0 ? s.reserved : 1;
  const int gs_src = (s.div >> 16) & 0xffff, gs_dst = s.div & 0xffff;
  const bool accum = (s.flags & TBX_F_ACCUM) != 0;
  const bool rowskip = (s.flags & TBX_F_ROWSKIP) != 0 && !to_global;
  const int kblocks = (s.k + 15) / 16;
  const int NO = G * N;
  const int ncb = (NO + GCOLS - 1) / GCOLS;
  const int nkc = (kblocks + GKC - 1) / GKC;
  const int r = (int)(threadIdx.x >> 7);  // the thread's row (wave-uniform: a wave is 64 consecutive columns of one row)
  const int cl = (int)(threadIdx.x & (GCOLS - 1));
  const bool row_on = r < t.rows_live;
  // row-skip flags were gathered into LDS by the kernel prologue: an ordinary global load here would make the compiler drain
  // the weight DMAs in flight (vmcnt is in order) - one exposed memory round trip per stage
  const bool skip = r >= t.n_valid || (rowskip && ((skipflags[stage] >> r) & 1u) != 0u);
  TBX_SUB(0);
  // chunk 0 is already on its way if the previous LINEAR stage (or the kernel prologue) requested it
  int slot = ws.stage == stage ? ws.slot : 0;
  const int w0 = 2 * t.rows_live < nwave ? 2 * t.rows_live : 0;  // waves that multiply and therefore do not fetch
  if (ws.stage != stage) gemv_dma(img, kblocks, 0, 0, wslots + slot * GSLOT, lane, wave, nwave, w0);
  const int nj = s.pad;  // next TBX_F_WGEMV stage (0: none)
  const int n_chunks = ncb * nkc;
  const bool n_pow2 = (N & (N - 1)) == 0;
  const int n_shift = __builtin_ctz((unsigned)N);
  float acc = 0.f;
  int cb = 0, kc = 0;  // chunk i = (cb, kc), walked without divisions
  for (int i = 0; i < n_chunks; ++i) {
    // chunk i has landed (every wave waits for its own pieces, the barrier collects all waves'); everybody is also done with the
    // other slot (chunk i - 1), so the next chunk may go there
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    TBX_SUB(1);
    const float* w = wslots + slot * GSLOT;
    if (i + 1 < n_chunks) {
      const bool wrap = kc + 1 == nkc;
      gemv_dma(img, kblocks, wrap ? cb + 1 : cb, wrap ? 0 : kc + 1, wslots + (1 - slot) * GSLOT, lane, wave, nwave, w0);
    } else if (nj > 0) {
      const float* nimg = (const float*)(((uint64_t)prog_word(prog, nj, 17) << 32) | prog_word(prog, nj, 16));
      gemv_dma(nimg, gemv_kblocks(prog, nj), 0, 0, wslots + (1 - slot) * GSLOT, lane, wave, nwave, w0);
    }
    TBX_SUB(2);
    const int o = cb * GCOLS + cl;
    const bool live = o < NO;
    const int oc = live ? o : NO - 1;
    const int grp = G > 1 ? (n_pow2 ? oc >> n_shift : oc / N) : 0;
    const int c = oc - grp * N;
    if (row_on) {
      const float* wc = w + cl * 4;
      if (kc == 0) {
        acc = wc[0];  // the block's bias row
        if (accum && live) acc += dst0[r * lds_d + grp * gs_dst + c];
        wc += GROW;
      }
      const float* xr = src0 + grp * gs_src + r * lds_s + kc * GKC * 16;
      const int kbc = (kblocks - kc * GKC) < GKC ? (kblocks - kc * GKC) : GKC;
      const float4* xr4 = (const float4*)__builtin_assume_aligned(xr, 16);  // src_col, the row widths and the chunk offset are multiples of 4 floats
      const float4* wc4 = (const float4*)__builtin_assume_aligned(wc, 16);
      float a = acc;
      // software pipeline: the 8 operand reads of k-block kb + 1 go out before the 16 dependent fmas of k-block kb (the compiler's
      // own schedule re-used one register set: reads and fmas alternated; 2807 -> 1876 cycles per 128 k in isolation together
      // with the DMA change above)
      // ... and the activations through DPP: a lane reads ONE of the k-block's four float4s (lane & 3 picks which), the fma takes
      // its x operand from the quad lane that holds it (v_fmac_f32_dpp quad_perm:[g,g,g,g]; as inline asm - the compiler does not
      // fold a v_mov_dpp into the fma). 5 KiB instead of 8 KiB of LDS returns per wave and k-block: the loop is bound by the LDS
      // return path (1829 -> 1636 cycles per 128 k in isolation). Same products in the same order: bit-identical.
#define TBX_RD(XQ, W, KB)                                                                                           \
  XQ = xr4[(KB) * 4 + (lane & 3)];                                                                                  \
  W[0] = wc4[((KB) * 4) * (GROW / 4)], W[1] = wc4[((KB) * 4 + 1) * (GROW / 4)], W[2] = wc4[((KB) * 4 + 2) * (GROW / 4)], \
  W[3] = wc4[((KB) * 4 + 3) * (GROW / 4)]
#define TBX_FD(XC, WC, G) \
  asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #G "," #G "," #G "," #G "] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(XC), "v"(WC))
#define TBX_FM(XQ, W)                                                                                  \
  TBX_FD(XQ.x, W[0].x, 0); TBX_FD(XQ.x, W[0].y, 1); TBX_FD(XQ.x, W[0].z, 2); TBX_FD(XQ.x, W[0].w, 3);      \
  TBX_FD(XQ.y, W[1].x, 0); TBX_FD(XQ.y, W[1].y, 1); TBX_FD(XQ.y, W[1].z, 2); TBX_FD(XQ.y, W[1].w, 3);      \
  TBX_FD(XQ.z, W[2].x, 0); TBX_FD(XQ.z, W[2].y, 1); TBX_FD(XQ.z, W[2].z, 2); TBX_FD(XQ.z, W[2].w, 3);      \
  TBX_FD(XQ.w, W[3].x, 0); TBX_FD(XQ.w, W[3].y, 1); TBX_FD(XQ.w, W[3].z, 2); TBX_FD(XQ.w, W[3].w, 3)
      float4 xa, xb, wa[4], wb[4];
      TBX_RD(xa, wa, 0);
      if (kbc == GKC) {  // the full chunk: straight-line
#pragma unroll
        for (int kb = 0; kb < GKC; kb += 2) {
          TBX_RD(xb, wb, kb + 1);
          TBX_FM(xa, wa);
          if (kb + 2 < GKC) { TBX_RD(xa, wa, kb + 2); }
          TBX_FM(xb, wb);
        }
      } else {
        for (int kb = 0; kb < kbc; kb += 2) {
          const bool more = kb + 1 < kbc;  // wave-uniform
          if (more) { TBX_RD(xb, wb, kb + 1); }
          TBX_FM(xa, wa);
          if (more) {
            if (kb + 2 < kbc) { TBX_RD(xa, wa, kb + 2); }
            TBX_FM(xb, wb);
          }
        }
      }
#undef TBX_RD
#undef TBX_FD
#undef TBX_FM
      acc = a;
      TBX_SUB(3);
      if (kc == nkc - 1) {
        float v = acc;
        if (s.act == TBX_ACT_RELU) v = fmaxf(v, 0.f);
        if (to_global) {
          if (live && r < t.n_valid) {
            if (s.flags & TBX_F_OUT_BF16)
              gst1_bf16((float*)s.p2, (t.g0 + r) * (int64_t)s.ld2 + s.dst_col + grp * gs_dst + c, v);
            else
              gst1(gout0 + (int64_t)r * s.ld2 + grp * gs_dst + c, v);
          }
        } else if (live) {
          if (!rowskip || !skip)
            dst0[r * lds_d + grp * gs_dst + c] = v;
          else if (s.flags & TBX_F_ROWZERO)
            dst0[r * lds_d + grp * gs_dst + c] = 0.f;
        } else if (!accum && G == 1 && o < (N + 15) / 16 * 16 && o < lds_d - s.dst_col) {
          dst0[r * lds_d + o] = 0.f;  // K padding of the next stage, as the MFMA path leaves it
        }
      }
    }
    slot = 1 - slot;
    if (++kc == nkc) kc = 0, ++cb;
    TBX_SUB(4);
  }
  // `slot` now names the buffer the look-ahead request went to
  ws.stage = nj > 0 ? nj : -1;
  ws.slot = slot;
}

// LINEAR (optionally grouped: `reserved` = G groups, group g reads src columns src_col + g*src_stride, writes
// dst_col + g*dst_stride, with src_stride / dst_stride packed in `div` as (src << 16 | dst); its weight block is the next
// n rows (or k rows if TBX_F_WT) after the previous group's, its bias the next n entries).
template <int MT, bool EXT, bool FULL, bool LIVE>
__device__ void op_linear(const tbx_stage_t& s, const Tile<MT, EXT>& t, const uint32_t* prog, int stage, WeightAhead& wa) {
  if constexpr (LIVE) return;  // tbx_rowchain_live programs run linear_gemv (called by the kernel: it owns the LDS weight slots)
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
  const int n_tiles = (N + 15) / 16;
  const int kblocks = (K + 15) / 16;
  const int tiles_total = G * n_tiles;
  // TBX_F_WPACK: p0 holds the MFMA B fragments in load order (tbx_pack_weight): every wave-wide float4 load is one
  // contiguous 1 KiB. A row-major [n,k] weight makes the 16 lanes of a fragment row read 16 different 512-byte rows,
  // i.e. 64 separate 16-byte requests per load - measured 2.6 us of a 5.4 us 128x128 stage, L2-resident or not.
  const bool packed = (s.flags & TBX_F_WPACK) != 0;

  if (packed) {
    // wa holds this stage's first fragments by construction: requested at kernel start for the first packed LINEAR
    // (RowchainArgs::first_packed) and by every packed LINEAR for the next one (stage.pad, set by tbx_rowchain_ex)
    const int nj = s.pad;
    if (wave < tiles_total)
      linear_packed<MT, EXT, FULL>(s, t, prog, wa, lane, wave, nwave, to_global);
    else if (nj > 0)  // no tile here: only keep the pipeline of the next stage fed
      weights_ahead(prog, nj, lane, wave, wa);
    wa.stage = nj > 0 ? nj : -1;
    return;
  }

  // row-major weights ([n,k], or [k,n] with TBX_F_WT): reference path of the ABI, one k-block at a time (not in the lean kernel)
  if constexpr (FULL)
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
      const float b = (bias != nullptr && col_ok) ? gld1(bias + col) : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float c0 = b;
        if (accum && col_ok) c0 += dst[(m * 16 + g * 4 + r) * lds_d + col];
        acc[m][r] = c0;
      }
    }
    const float* W = W0 + (int64_t)grp * (wt ? K : N) * ldw;
    for (int kb = 0; kb < kblocks; ++kb) {
      const int k0 = kb * 16 + g * 4;
      float tmp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = k0 + q;
        float w = 0.f;
        if (col_ok && kk < K) w = gld1(wt ? W + (int64_t)kk * ldw + col : W + (int64_t)col * ldw + kk);
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
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][r];
        if (s.act == TBX_ACT_RELU) v = fmaxf(v, 0.f);
        if (to_global) {
          if (col_ok && m * 16 + g * 4 + r < t.n_valid) gst1(gout + (int64_t)(m * 16 + g * 4 + r) * s.ld2 + col, v);
        } else if (col_ok)
          dst[(m * 16 + g * 4 + r) * lds_d + col] = v;
        else if (!accum && G == 1 && col < lds_d - s.dst_col)
          dst[(m * 16 + g * 4 + r) * lds_d + col] = 0.f;
      }
    }
  }
}

template <int NQ>
__device__ __forceinline__ void ln_row(const float* __restrict__ src, float* __restrict__ dst, int n, int lane, float eps,
                                       const float* __restrict__ gamma, const float* __restrict__ beta) {
  float v[NQ], gm[NQ], bt[NQ];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q;
    const bool ok = c < n;
    v[q] = ok ? src[c] : 0.f;
    gm[q] = ok ? gld1(gamma + c) : 0.f;  // issued before the reductions: their latency overlaps them
    bt[q] = ok ? gld1(beta + c) : 0.f;
    sum += v[q];
  }
  sum = tbx::wave_sum(sum);
  const float mean = sum / (float)n;
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float d = (lane + 64 * q < n) ? v[q] - mean : 0.f;
    var += d * d;
  }
  var = tbx::wave_sum(var) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = lane + 64 * q;
    if (c < n) dst[c] = (v[q] - mean) * rstd * gm[q] + bt[q];
  }
}

template <int MT, bool EXT>
__device__ void op_layernorm(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  TBX_WAVE_ROWS;
  const float* src = t.b(s.src) + s.src_col;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_s = t.l(s.src), lds_d = t.l(s.dst);
  const float* gamma = (const float*)s.p0;
  const float* beta = (const float*)s.p1;
  const int n = s.n;
  for (int r = wave; r < t.rows_live; r += nwave) {
    if (n <= 128)
      ln_row<2>(src + r * lds_s, dst + r * lds_d, n, lane, s.f0, gamma, beta);
    else if (n <= 256)
      ln_row<4>(src + r * lds_s, dst + r * lds_d, n, lane, s.f0, gamma, beta);
    else
      ln_row<8>(src + r * lds_s, dst + r * lds_d, n, lane, s.f0, gamma, beta);
  }
}

template <int MT, bool EXT>
__device__ void op_elementwise(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  TBX_WAVE_ROWS;
  const float* src = t.b(s.src) + s.src_col;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_s = t.l(s.src), lds_d = t.l(s.dst);
  const int n = s.n;
  if (s.op != TBX_OP_CLAMP && (n & 3) == 0 && ((s.src_col | s.dst_col) & 3) == 0) {
    const int w4 = n >> 2;
    for (int r = wave; r < t.rows_live; r += nwave)
      for (int c4 = lane; c4 < w4; c4 += 64) {
        float4 v = *(const float4*)(src + r * lds_s + c4 * 4);
        float4* d = (float4*)(dst + r * lds_d + c4 * 4);
        if (s.op == TBX_OP_ADD) {
          const float4 o = *d;
          v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
        }
        *d = v;
      }
    return;
  }
  for (int r = wave; r < t.rows_live; r += nwave)
    for (int c = lane; c < n; c += 64) {
      if (s.op == TBX_OP_ADD)
        dst[r * lds_d + c] += src[r * lds_s + c];
      else if (s.op == TBX_OP_COPY)
        dst[r * lds_d + c] = src[r * lds_s + c];
      else  // CLAMP
        dst[r * lds_d + c] = fminf(fmaxf(dst[r * lds_d + c], s.f0), s.f1);
    }
}

// tbx_keyed_dropout's mask (csrc/dropout.hip) applied in place to an LDS-resident block: time_batch = 1, so the key row is the
// chain's global row.
template <int MT, bool EXT>
__device__ void op_dropout(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = Tile<MT, EXT>::ROWS;
  TBX_WAVE_ROWS;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_d = t.l(s.dst);
  const int n = s.n;
  const uint64_t sd = *(const TBX_GLOBAL uint64_t*)s.p0;
  const uint32_t site = (uint32_t)s.div, ts = (uint32_t)s.k, thresh = (uint32_t)s.reserved;
  const uint32_t lo = (uint32_t)sd ^ (site * 0x85EBCA6Bu) ^ (ts * 0x27D4EB2Fu);
  const uint32_t hi = (uint32_t)(sd >> 32) + site * 0xC2B2AE35u + ts * 0x165667B1u;
  for (int r = wave; r < ROWS; r += nwave) {
    const uint32_t base = (uint32_t)(t.g0 + r) * (uint32_t)n;
    for (int c = lane; c < n; c += 64) {
      uint32_t x = (base + (uint32_t)c) ^ lo;
      x *= 0x9E3779B1u;
      x ^= hi;
      x ^= x >> 16;
      x *= 0x7feb352du;
      x ^= x >> 15;
      x *= 0x846ca68bu;
      x ^= x >> 16;
      const float v = dst[r * lds_d + c];
      dst[r * lds_d + c] = x >= thresh ? v * s.f0 : 0.f;
    }
  }
}

template <int MT, bool EXT>
__device__ void op_rowmask(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  TBX_WAVE_ROWS;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_d = t.l(s.dst);
  const uint8_t* mask = (const uint8_t*)s.p0;
  const int n = s.n;
  for (int r = wave; r < t.rows_live; r += nwave) {
    bool m = r >= t.n_valid;
    if (!m && mask != nullptr) m = (gld1(mask + row_of(s, t.g0 + r)) != 0) != ((s.flags & TBX_F_MASK_INV) != 0);
    if (m)
      for (int c = lane; c < n; c += 64) dst[r * lds_d + c] = s.f0;
  }
}

template <int MT, bool EXT>
__device__ void op_groupmax(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = Tile<MT, EXT>::ROWS;
  float* src = t.b(s.src) + s.src_col;
  float* dst = t.b(s.dst) + s.dst_col;
  const int lds_s = t.l(s.src), lds_d = t.l(s.dst);
  // p1 != NULL: the masked form - one stage for [ROWMASK(-inf) -> GROUPMAX -> ROWMASK(0) over both halves] of a PointNet layer
  // (polyline_encoder.py:52-58): rows whose byte p1[g] is set (and padding rows) are left out of the maximum and come out as 0 in
  // src and dst columns alike. The tile's row bytes are read once per wave (one ballot), not once per row and thread.
  const uint8_t* mask = (const uint8_t*)s.p1;
  unsigned long long inv = 0ull;
  if (mask != nullptr) {
    const int lane = threadIdx.x & 63;
    bool m = true;
    if (lane < ROWS && lane < t.n_valid) m = gld1(mask + t.g0 + lane) != 0;
    inv = __ballot(m);
  }
  auto off = [&](int r) { return ((inv >> r) & 1ull) != 0ull; };
  if (t.ng == 1) {
    for (int c = threadIdx.x; c < s.n; c += blockDim.x) {
      float m = -INFINITY;
      for (int r = 0; r < ROWS; ++r)
        if (!off(r)) m = fmaxf(m, src[r * lds_s + c]);
      for (int r = 0; r < ROWS; ++r) {
        dst[r * lds_d + c] = off(r) ? 0.f : m;
        if (off(r)) src[r * lds_s + c] = 0.f;
      }
    }
    return;
  }
  // several groups in the tile: one (group, column) per thread; padding rows past the last group are left alone (their
  // content is never read across rows and the program masks them)
  for (int e = threadIdx.x; e < s.n * t.ng; e += blockDim.x) {
    const int j = e / s.n, c = e - j * s.n;
    float m = -INFINITY;
    for (int r = j * t.gw; r < (j + 1) * t.gw; ++r)
      if (!off(r)) m = fmaxf(m, src[r * lds_s + c]);
    for (int r = j * t.gw; r < (j + 1) * t.gw; ++r) {
      dst[r * lds_d + c] = off(r) ? 0.f : m;
      if (off(r)) src[r * lds_s + c] = 0.f;
    }
  }
}

template <int MT, bool EXT>
__device__ void op_poolmax(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  const float* src = t.b(s.src) + s.src_col;
  const int lds_s = t.l(s.src);
  const uint8_t* mask = (const uint8_t*)s.p1;
  float* out = (float*)s.p0;
  for (int e = threadIdx.x; e < s.n * t.ng; e += blockDim.x) {
    const int j = e / s.n, c = e - j * s.n;
    const int r0 = t.ng == 1 ? 0 : j * t.gw, r1 = t.ng == 1 ? t.n_valid : (j + 1) * t.gw;
    float m = -INFINITY;
    bool any = false;
    for (int r = r0; r < r1; ++r) {
      if (mask != nullptr && gld1(mask + t.g0 + r) != 0) continue;
      any = true;
      m = fmaxf(m, src[r * lds_s + c]);
    }
    const float v = any ? m : 0.f;
    gst1(out + (t.group + j) * (int64_t)s.ld + s.dst_col + c, v);
    if (s.flags & TBX_F_POOL_KEEP) t.b(s.dst)[j * t.l(s.dst) + s.k + c] = v;  // (dst != src: other threads still read the group's rows)
  }
}

template <int MT, bool EXT>
__device__ void op_store(const tbx_stage_t& s, const Tile<MT, EXT>& t) {
  constexpr int ROWS = Tile<MT, EXT>::ROWS;
  TBX_WAVE_ROWS;
  const float* src = t.b(s.src) + s.src_col;
  const int lds_s = t.l(s.src);
  float* out = (float*)s.p0;
  const int n = s.n;
  if (s.flags & TBX_F_OUT_BF16) {  // bfloat16 destination (ld, dst_col in elements): 4 values = 8 bytes per lane
    for (int r = wave; r < ROWS && r < t.n_valid; r += nwave) {
      TBX_GLOBAL uint16_t* orow = (TBX_GLOBAL uint16_t*)out + (t.g0 + r) * (int64_t)s.ld + s.dst_col;
      for (int c4 = lane; c4 < (n >> 2); c4 += 64) {
        const float4 v = *(const float4*)(src + r * lds_s + c4 * 4);
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 pk;
        pk[0] = (uint32_t)to_bf16(v.x) | ((uint32_t)to_bf16(v.y) << 16);
        pk[1] = (uint32_t)to_bf16(v.z) | ((uint32_t)to_bf16(v.w) << 16);
        *(TBX_GLOBAL u32x2*)(orow + c4 * 4) = pk;
      }
    }
    return;
  }
  if (s.flags & TBX_F_MASKED_SUM) {  // sum of the groups whose mask byte is clear, in group order from 0 (0 + y == y exactly)
    const uint8_t* mask = (const uint8_t*)s.p1;
    for (int r = wave; r < ROWS && r < t.n_valid; r += nwave) {
      float* orow = out + (t.g0 + r) * (int64_t)s.ld + s.dst_col;
      for (int c = lane; c < n; c += 64) {
        float v = 0.f;
        for (int i = 0; i < s.reserved; ++i)
          if (gld1(mask + (int64_t)i * s.k + t.g0 + r) == 0) v += src[r * lds_s + i * s.div + c];
        gst1(orow + c, v);
      }
    }
    return;
  }
  if ((n & 3) == 0 && (s.ld & 3) == 0 && (s.dst_col & 3) == 0 && (s.src_col & 3) == 0 && ((((uintptr_t)out) & 15) == 0)) {
    const int w4 = n >> 2;
    for (int r = wave; r < ROWS && r < t.n_valid; r += nwave) {
      float* orow = out + (t.g0 + r) * (int64_t)s.ld + s.dst_col;
      for (int c4 = lane; c4 < w4; c4 += 64) gst4(orow + c4 * 4, *(const float4*)(src + r * lds_s + c4 * 4));
    }
    return;
  }
  for (int r = wave; r < ROWS && r < t.n_valid; r += nwave) {
    float* orow = out + (t.g0 + r) * (int64_t)s.ld + s.dst_col;
    for (int c = lane; c < n; c += 64) gst1(orow + c, src[r * lds_s + c]);
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

// FULL = false: the lean instantiation every program of the hot path runs on - packed exact-fp32 LINEAR stages only, no DROPOUT.
// Row-major weights (the ABI's reference path), split-bf16 stages and the training dropouts select FULL = true. The interpreter
// is sensitive to its own size (its text is ~2/3 of the instruction cache): what a program does not use is not compiled in.
template <int MT, bool EXT, bool FULL, bool LIVE = false>
__global__ __launch_bounds__(512) void rowchain_kernel(const RowchainArgs a) {
  constexpr int ROWS = Tile<MT, EXT>::ROWS;
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
  t.gw = ROWS;
  t.ng = 1;
  if (a.group_rows > 0) {
    // as many whole groups as fit the tile (consecutive groups are consecutive rows): the per-stage fixed costs (weight
    // stream, barriers, decode) are shared by all of them
    const int per = ROWS / a.group_rows;
    const int64_t n_groups = a.n_rows / a.group_rows;
    t.group = (int64_t)blockIdx.x * per;
    const int64_t left = n_groups - t.group;
    t.gw = a.group_rows;
    t.ng = left < per ? (int)left : per;
    t.g0 = t.group * a.group_rows;
    t.n_valid = t.ng * a.group_rows;
  } else {
    const int per = a.live_rows > 0 ? a.live_rows : ROWS;
    t.g0 = (int64_t)blockIdx.x * per;
    const int64_t left = a.n_rows - t.g0;
    t.n_valid = left < per ? (int)left : per;
  }
  t.rows_live = a.live_rows > 0 ? a.live_rows : ROWS;
  // The program is copied kernarg -> LDS once (one parallel read) and each stage descriptor is decoded from LDS into
  // SGPRs (readfirstlane): a scalar load per stage from the kernarg segment costs ~0.6 us of exposed latency per stage.
  // live kernel: [activations | two 64 KiB weight slots | program]
  float* wslots = lds + (size_t)ROWS * (a.ldw0 + a.ldw1 + a.ld_aux);
  uint32_t* skipflags = (uint32_t*)(wslots + (LIVE ? 2 * GSLOT : 0));  // live kernel: one word per stage, bit r = skip row r
  uint32_t* prog = skipflags + (LIVE ? TBX_MAX_STAGES : 0);
  {
    const uint32_t* ka = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int e = threadIdx.x; e < a.n_stages * SW; e += blockDim.x) prog[e] = ka[e];
  }
  __syncthreads();
  WeightAhead wa;
#pragma unroll
  for (int q = 0; q < CH; ++q) wa.w[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  wa.bias = 0.f;
  wa.stage = -1;
  if (!LIVE && a.first_packed >= 0)
    weights_ahead(prog, a.first_packed, (int)(threadIdx.x & 63), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wa);
  WeightStream ws;
  ws.stage = -1, ws.slot = 0;
  if constexpr (LIVE) {  // TBX_F_ROWSKIP masks of every LINEAR stage -> LDS, before any weight DMA is in flight
    if ((int)threadIdx.x < a.n_stages) {
      const int j = (int)threadIdx.x;
      uint32_t f = 0u;
      if (prog[j * SW + 0] == (uint32_t)TBX_OP_LINEAR && (prog[j * SW + 8] & (uint32_t)TBX_F_ROWSKIP)) {
        const uint8_t* m = (const uint8_t*)(((uint64_t)prog[j * SW + 19] << 32) | prog[j * SW + 18]);
        const bool inv = (prog[j * SW + 8] & (uint32_t)TBX_F_MASK_INV) != 0u;
        for (int r = 0; r < t.n_valid; ++r) f |= ((gld1(m + t.g0 + r) != 0) != inv) ? (1u << r) : 0u;
      }
      skipflags[j] = f;
    }
    __syncthreads();
  }
  if (LIVE && a.first_packed >= 0) {  // the first LINEAR stage's first chunk: on its way while the stages before it run
    const int j = a.first_packed;
    gemv_dma((const float*)(((uint64_t)prog_word(prog, j, 17) << 32) | prog_word(prog, j, 16)), gemv_kblocks(prog, j), 0, 0, wslots,
             (int)(threadIdx.x & 63), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(blockDim.x >> 6),
             2 * t.rows_live < (int)(blockDim.x >> 6) ? 2 * t.rows_live : 0);
    ws.stage = j;
  }
  for (int i = 0; i < a.n_stages; ++i) {
    const uint32_t* ps = prog + i * SW;
    auto rd = [&](int w) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)ps[w]); };
    auto rdp = [&](int w) -> const void* { return (const void*)(((uint64_t)rd(w + 1) << 32) | rd(w)); };
    tbx_stage_t s;
    s.op = (int32_t)rd(0), s.src = (int32_t)rd(1), s.dst = (int32_t)rd(2), s.src_col = (int32_t)rd(3);
    s.dst_col = (int32_t)rd(4), s.k = (int32_t)rd(5), s.n = (int32_t)rd(6), s.act = (int32_t)rd(7);
    s.flags = (int32_t)rd(8), s.ld = (int32_t)rd(9), s.div = (int32_t)rd(10), s.reserved = (int32_t)rd(11);
    s.ld2 = (int32_t)rd(12), s.pad = (int32_t)rd(13);
    s.f0 = __uint_as_float(rd(14)), s.f1 = __uint_as_float(rd(15));
    s.p0 = rdp(16), s.p1 = rdp(18), s.p2 = rdp(20);
    switch (s.op) {
      case TBX_OP_LOAD: op_load<MT, EXT>(s, t); break;
      case TBX_OP_LINEAR:
        if constexpr (LIVE)
          linear_gemv<MT, EXT>(s, t, prog, wslots, skipflags, ws, i);
        else
          op_linear<MT, EXT, FULL, LIVE>(s, t, prog, i, wa);
        break;
      case TBX_OP_LAYERNORM: op_layernorm<MT, EXT>(s, t); break;
      case TBX_OP_ADD:
      case TBX_OP_COPY:
      case TBX_OP_CLAMP: op_elementwise<MT, EXT>(s, t); break;
      case TBX_OP_ROWMASK: op_rowmask<MT, EXT>(s, t); break;
      case TBX_OP_DROPOUT:
        if constexpr (FULL) op_dropout<MT, EXT>(s, t);
        break;
      case TBX_OP_GROUPMAX:
        if constexpr (!LIVE) op_groupmax<MT, EXT>(s, t);
        break;
      case TBX_OP_POOLMAX:
        if constexpr (!LIVE) op_poolmax<MT, EXT>(s, t);
        break;
      case TBX_OP_STORE: op_store<MT, EXT>(s, t); break;
      default: break;
    }
    // Stage barrier: orders LDS only. Weight loads issued ahead stay in flight across it, and global memory written by
    // a stage (STORE / POOLMAX / LINEAR-to-global) is an output of the launch, never read back by a later stage.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (!LIVE) {
      if (s.op == TBX_OP_POOLMAX && (s.flags & TBX_F_POOL_KEEP)) {  // the tile goes on as a flat tile of its pooled rows
        t.g0 = t.group;
        t.n_valid = t.ng;
        t.gw = ROWS;
        t.ng = 1;
      }
    }
    TBX_CLK(i + 1);
  }
#ifdef TBX_STAGE_CLOCK
  if (blockIdx.x == 0 && threadIdx.x == 0 && clk_slot < (unsigned)CLK_LAUNCHES) g_clk[clk_slot * CLK_SLOTS + CLK_SLOTS - 2] = clock64();
#endif
}

// Image layout per 16-column tile (tile index = grp*n_tiles + tile): kblocks x [64 lanes][4] weights, where lane l / slot t
// holds W_grp[col = tile*16 + (l & 15)][kk = kb*16 + (l >> 4)*4 + t], then [64] bias values bias[grp*n + col(l)].
// split == 0: the fp32 image. split != 0: the same 16 bytes per (lane, k-block) hold the four weights as bf16 hi (8 bytes)
// followed by bf16 lo = bf16(w - hi) (8 bytes): operands of the three-product bf16 MFMA path (TBX_F_WSPLIT).
__global__ void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ bias, int n, int k, int ld,
                                   int groups, int wt, int split, float* __restrict__ out, int64_t total) {
  const int n_tiles = (n + 15) / 16, kblocks = (k + 15) / 16;
  const int tstride = kblocks * 256 + 64;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tix = e / tstride;
    const int o = (int)(e - tix * tstride);
    const int tile = (int)(tix % n_tiles), grp = (int)(tix / n_tiles);
    float v = 0.f;
    if (o < kblocks * 256) {
      const int t = o & 3, lane = (o >> 2) & 63, kb = o >> 8;
      const int col = tile * 16 + (lane & 15);
      auto wv = [&](int tt) {
        const int kk = kb * 16 + (lane >> 4) * 4 + tt;
        return (col < n && kk < k) ? (wt ? w[((int64_t)grp * k + kk) * ld + col] : w[((int64_t)grp * n + col) * ld + kk]) : 0.f;
      };
      if (!split) {
        v = wv(t);
      } else {  // dword t of the lane's 16 bytes: t = 0,1 -> hi pairs (0,1),(2,3); t = 2,3 -> lo pairs
        const int p = (t & 1) * 2;
        const float a = wv(p), b = wv(p + 1);
        const __bf16 ha = (__bf16)a, hb = (__bf16)b;
        __bf16 x = ha, y = hb;
        if (t >= 2) {
          x = (__bf16)(a - (float)ha);
          y = (__bf16)(b - (float)hb);
        }
        const uint32_t bits = (uint32_t)__builtin_bit_cast(unsigned short, x) | ((uint32_t)__builtin_bit_cast(unsigned short, y) << 16);
        v = __uint_as_float(bits);
      }
    } else {
      const int col = tile * 16 + ((o - kblocks * 256) & 15);
      if (bias != nullptr && col < n) v = bias[(int64_t)grp * n + col];
    }
    out[e] = v;
  }
}

// tbx_pack_weight_gemv image: [ncb column blocks][1 + kblocks*4 float4 rows][128 columns][4]. Row 0 of a block: (bias, 0, 0, 0)
// per column; row 1 + q: W_g[c][kb*16 + {0,4,8,12} + t] of output o = (g, c), q = kb*4 + t (zero beyond k / beyond the outputs).
__global__ void pack_weight_gemv_kernel(const float* __restrict__ w, const float* __restrict__ bias, int n, int k, int ld,
                                        int groups, int wt, float* __restrict__ out, int64_t total) {
  const int kblocks = (k + 15) / 16, NO = groups * n;
  const int64_t per_cb = (int64_t)(1 + kblocks * 4) * GROW;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int cb = (int)(e / per_cb);
    const int64_t f = e - cb * per_cb;
    const int j = (int)(f & 3);
    const int cl = (int)((f >> 2) % GCOLS);
    const int row = (int)((f >> 2) / GCOLS);
    const int o = cb * GCOLS + cl;
    float v = 0.f;
    if (row == 0) {
      if (j == 0 && bias != nullptr && o < NO) v = bias[o];
    } else {
      const int q = row - 1;
      const int kk = (q >> 2) * 16 + j * 4 + (q & 3);
      if (o < NO && kk < k) {
        const int grp = o / n, c = o - grp * n;
        v = wt ? w[((int64_t)grp * k + kk) * ld + c] : w[((int64_t)grp * n + c) * ld + kk];
      }
    }
    out[e] = v;
  }
}

int check_stage(const tbx_stage_t& s, int ldw0, int ldw1, int ld_aux, int tile_rows) {
  auto buf_ld = [&](int b) { return b == 0 ? ldw0 : (b == 1 ? ldw1 : (b == 2 ? ld_aux : (1 << 30))); };
  if (s.op < TBX_OP_LOAD || s.op > TBX_OP_DROPOUT) return TBX_ERR_ARG;
  if (s.op == TBX_OP_DROPOUT && (s.p0 == nullptr || s.k < 0)) return TBX_ERR_ARG;
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
  if (s.op == TBX_OP_LOAD && (s.flags & TBX_F_LOAD2)) {  // only the whole-float4 path carries the second source
    if (s.p0 == nullptr || s.p2 == nullptr || (s.flags & TBX_F_ACCUM) || s.k > s.n) return TBX_ERR_ARG;
    if ((s.n & 3) || (s.ld & 3) || (s.dst_col & 3) || (s.reserved & 3) || (s.ld2 & 3) || (s.src_col & 3) || s.reserved <= 0 || s.reserved > 256 ||
        s.ld2 < s.reserved)
      return TBX_ERR_UNSUPPORTED;
    if ((((uintptr_t)s.p0) | ((uintptr_t)s.p2)) & 15) return TBX_ERR_ALIGN;
    if (s.src < 0 || s.src > 2 || s.src_col < 0 || s.src_col + s.reserved > buf_ld(s.src)) return TBX_ERR_UNSUPPORTED;
    if (s.src == s.dst && !(s.src_col >= s.dst_col + s.n || s.dst_col >= s.src_col + s.reserved)) return TBX_ERR_UNSUPPORTED;
  }
  if (s.op == TBX_OP_LINEAR && (s.flags & TBX_F_ROWSKIP) && (!(s.flags & (TBX_F_WPACK | TBX_F_WGEMV)) || gdst || s.p1 == nullptr)) return TBX_ERR_UNSUPPORTED;
  if (s.op == TBX_OP_LINEAR && (s.flags & TBX_F_WGEMV) && (s.flags & (TBX_F_WPACK | TBX_F_WSPLIT | TBX_F_WT))) return TBX_ERR_ARG;
  if (s.flags & TBX_F_OUT_BF16) {
    if (s.op == TBX_OP_STORE) {
      if ((s.n & 3) || (s.ld & 3) || (s.dst_col & 3) || (s.src_col & 3) || (((uintptr_t)s.p0) & 7)) return TBX_ERR_ALIGN;
    } else if (!(gdst && (s.flags & (TBX_F_WPACK | TBX_F_WGEMV)) && !(s.flags & TBX_F_WSPLIT))) {
      return TBX_ERR_UNSUPPORTED;
    }
  }
  if ((s.flags & TBX_F_ROWZERO) && !(s.op == TBX_OP_LINEAR && (s.flags & TBX_F_ROWSKIP))) return TBX_ERR_ARG;
  if (s.flags & TBX_F_MASKED_SUM) {
    if (s.op != TBX_OP_STORE || (s.flags & TBX_F_OUT_BF16)) return TBX_ERR_UNSUPPORTED;
    if (s.p1 == nullptr || s.reserved <= 0 || s.div <= 0 || s.k <= 0) return TBX_ERR_ARG;
    if (s.src_col + (s.reserved - 1) * s.div + s.n > buf_ld(s.src)) return TBX_ERR_UNSUPPORTED;
  }
  if (s.op == TBX_OP_LAYERNORM && (s.n > 512 || s.p0 == nullptr || s.p1 == nullptr)) return TBX_ERR_UNSUPPORTED;
  if ((s.op == TBX_OP_POOLMAX || s.op == TBX_OP_STORE) && (s.p0 == nullptr || s.ld <= 0)) return TBX_ERR_ARG;
  if (s.flags & TBX_F_POOL_KEEP) {
    if (s.op != TBX_OP_POOLMAX || s.dst == s.src || s.dst > 2) return TBX_ERR_ARG;
    if (s.k < 0 || s.k + s.n > buf_ld(s.dst)) return TBX_ERR_UNSUPPORTED;
  }
  if (s.op != TBX_OP_LINEAR && (s.flags & (TBX_F_ROW_DIV | TBX_F_ROW_MOD | TBX_F_ROW_BATCH_MOD)) && s.div <= 0) return TBX_ERR_ARG;
  if ((s.flags & TBX_F_ROW_BATCH_MOD) && s.k <= 0) return TBX_ERR_ARG;
  (void)tile_rows;
  return TBX_OK;
}

}  // namespace

extern "C" int tbx_rowchain_ex(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                               int ldw0, int ldw1, int ld_aux, void* stream);

#ifdef TBX_STAGE_CLOCK
extern "C" int tbx_debug_sub_dump(unsigned long long* host_out) {
  if (hipDeviceSynchronize() != hipSuccess) return TBX_ERR_LAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sub), 8 * sizeof(unsigned long long)) == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
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
  return (int64_t)groups * ((n + 15) / 16) * (((k + 15) / 16) * 256 + 64);
}

namespace {
int pack_weight_impl(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, int split, float* out,
                     void* stream) {
  if (w == nullptr || out == nullptr || n <= 0 || k <= 0 || ld <= 0 || groups <= 0) return TBX_ERR_ARG;
  if (groups > 1 && n % 16 != 0) return TBX_ERR_UNSUPPORTED;
  const int64_t total = tbx_pack_weight_size(n, k, groups);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, bias, n, k, ld, groups, wt, split,
                     out, total);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
}  // namespace

extern "C" int tbx_pack_weight(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out,
                               void* stream) {
  return pack_weight_impl(w, bias, n, k, ld, groups, wt, 0, out, stream);
}

extern "C" int tbx_pack_weight_split(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out,
                                     void* stream) {
  return pack_weight_impl(w, bias, n, k, ld, groups, wt, 1, out, stream);
}

extern "C" int64_t tbx_pack_weight_gemv_size(int n, int k, int groups) {
  if (n <= 0 || k <= 0 || groups <= 0) return TBX_ERR_ARG;
  const int64_t ncb = ((int64_t)groups * n + GCOLS - 1) / GCOLS;
  return ncb * (int64_t)(1 + ((k + 15) / 16) * 4) * GROW;
}

extern "C" int tbx_pack_weight_gemv(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out,
                                    void* stream) {
  if (w == nullptr || out == nullptr || n <= 0 || k <= 0 || ld <= 0 || groups <= 0) return TBX_ERR_ARG;
  const int64_t total = tbx_pack_weight_gemv_size(n, k, groups);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weight_gemv_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, bias, n, k, ld, groups, wt, out,
                     total);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

namespace {
int rowchain_launch(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows, int live_rows, int ldw0,
                    int ldw1, int ld_aux, void* stream);
}

extern "C" int tbx_rowchain_live(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int live_rows, int ldw0, int ldw1,
                                 int ld_aux, void* stream) {
  if (live_rows != 1 && live_rows != 2 && live_rows != 4) return TBX_ERR_UNSUPPORTED;
  return rowchain_launch(stages, n_stages, n_rows, 0, 16, live_rows, ldw0, ldw1, ld_aux, stream);
}

extern "C" int tbx_rowchain(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                            int ldw, void* stream) {
  return tbx_rowchain_ex(stages, n_stages, n_rows, group_rows, tile_rows, ldw, ldw, TBX_AUX_LD, stream);
}

extern "C" int tbx_rowchain_ex(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                               int ldw0, int ldw1, int ld_aux, void* stream) {
  return rowchain_launch(stages, n_stages, n_rows, group_rows, tile_rows, 0, ldw0, ldw1, ld_aux, stream);
}

namespace {
int rowchain_launch(const tbx_stage_t* stages, int n_stages, int64_t n_rows, int group_rows, int tile_rows, int live_rows, int ldw0,
                    int ldw1, int ld_aux, void* stream) {
  if (stages == nullptr || n_stages <= 0 || n_rows <= 0) return TBX_ERR_ARG;
  if (n_stages > TBX_MAX_STAGES) return TBX_ERR_UNSUPPORTED;
  if (tile_rows != 16 && tile_rows != 32 && tile_rows != 48) return TBX_ERR_UNSUPPORTED;
  if (ldw0 <= 0 || ldw1 <= 0 || ld_aux <= 0 || ldw0 % 4 != 0 || ldw1 % 4 != 0 || ld_aux % 4 != 0) return TBX_ERR_ALIGN;
  if (group_rows < 0 || group_rows > tile_rows) return TBX_ERR_UNSUPPORTED;
  if (group_rows > 0 && n_rows % group_rows != 0) return TBX_ERR_ARG;
  // activation buffers + the program itself (decoded from LDS by the kernel)
  const size_t lds_bytes = live_rows > 0 ? (size_t)(ldw0 + ldw1 + ld_aux) * 4 * sizeof(float) + 2 * GSLOT * sizeof(float) +
                                               TBX_MAX_STAGES * sizeof(uint32_t) + (size_t)n_stages * sizeof(tbx_stage_t)
                                         : (size_t)(ldw0 + ldw1 + ld_aux) * tile_rows * sizeof(float) + (size_t)n_stages * sizeof(tbx_stage_t);
  if (lds_bytes > 160 * 1024) return TBX_ERR_UNSUPPORTED;
  RowchainArgs a;
  for (int i = 0; i < n_stages; ++i) {
    const int rc = check_stage(stages[i], ldw0, ldw1, ld_aux, tile_rows);
    if (rc != TBX_OK) return rc;
    // thread-per-column LINEAR stages exist for live-row tiles only; group-wise stages need whole tiles
    const bool gemv = stages[i].op == TBX_OP_LINEAR && (stages[i].flags & TBX_F_WGEMV);
    if (gemv != (live_rows > 0 && stages[i].op == TBX_OP_LINEAR)) return TBX_ERR_UNSUPPORTED;
    if (live_rows > 0 && (stages[i].op == TBX_OP_GROUPMAX || stages[i].op == TBX_OP_POOLMAX || stages[i].op == TBX_OP_DROPOUT))
      return TBX_ERR_UNSUPPORTED;
    a.st[i] = stages[i];
  }
  // pad = index of the next packed LINEAR stage after this one (0: none): the kernel fetches that stage's first weights
  // while this one is still computing
  int next_packed = 0;
  for (int i = n_stages - 1; i >= 0; --i) {
    a.st[i].pad = next_packed;
    if (a.st[i].op == TBX_OP_LINEAR && (a.st[i].flags & (TBX_F_WPACK | TBX_F_WGEMV))) next_packed = i;
  }
  a.first_packed = -1;
  for (int i = n_stages - 1; i >= 0; --i)
    if (a.st[i].op == TBX_OP_LINEAR && (a.st[i].flags & (TBX_F_WPACK | TBX_F_WGEMV))) a.first_packed = i;
  a.n_stages = n_stages;
  a.group_rows = group_rows;
  a.ldw0 = ldw0;
  a.ldw1 = ldw1;
  a.ld_aux = ld_aux;
  a.n_rows = n_rows;
  a.live_rows = live_rows;
  const int per_tile = group_rows > 0 ? tile_rows / group_rows : 1;  // whole groups per tile
  const int rows_per_tile = live_rows > 0 ? live_rows : tile_rows;
  const int64_t n_tiles = group_rows > 0 ? (n_rows / group_rows + per_tile - 1) / per_tile : (n_rows + rows_per_tile - 1) / rows_per_tile;
  hipStream_t s = (hipStream_t)stream;
  bool ext = !(ldw0 == ldw1 && ld_aux == TBX_AUX_LD);
  for (int i = 0; i < n_stages; ++i) ext = ext || (stages[i].op == TBX_OP_LINEAR && stages[i].dst == TBX_BUF_GLOBAL);
  bool full = false;
  for (int i = 0; i < n_stages; ++i)
    full = full || stages[i].op == TBX_OP_DROPOUT ||
           (stages[i].op == TBX_OP_LINEAR && !(stages[i].flags & TBX_F_WGEMV) &&
            (!(stages[i].flags & TBX_F_WPACK) || (stages[i].flags & TBX_F_WSPLIT)));
#define TBX_RC_LAUNCH(MT, EXTF, FULLF, NT)                                                                                 \
  do {                                                                                                                     \
    if (lds_bytes > 64 * 1024)                                                                                             \
      (void)hipFuncSetAttribute((const void*)rowchain_kernel<MT, EXTF, FULLF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds_bytes);                                                                           \
    hipLaunchKernelGGL((rowchain_kernel<MT, EXTF, FULLF>), dim3((unsigned)n_tiles), dim3(NT), lds_bytes, s, a);            \
  } while (0)
#define TBX_RC_PICK(MT)                      \
  do {                                       \
    if (ext && full)                         \
      TBX_RC_LAUNCH(MT, true, true, 512);    \
    else if (ext)                            \
      TBX_RC_LAUNCH(MT, true, false, 512);   \
    else if (full)                           \
      TBX_RC_LAUNCH(MT, false, true, 512);   \
    else                                     \
      TBX_RC_LAUNCH(MT, false, false, 512);  \
  } while (0)
  if (live_rows > 0) {
    if (lds_bytes > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)rowchain_kernel<0, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL((rowchain_kernel<0, true, false, true>), dim3((unsigned)n_tiles), dim3(512), lds_bytes, s, a);
  } else if (tile_rows == 16)
    TBX_RC_PICK(1);
  else if (tile_rows == 32)
    TBX_RC_PICK(2);
  else
    TBX_RC_PICK(3);
#undef TBX_RC_PICK
#undef TBX_RC_LAUNCH
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
}  // namespace
