// tbx_knarpe_attn_fwd_mfma: the KNARPE attention of LARGE launches (a wavefront per source row) on the bf16 matrix cores.
//
// modules/attention_rpe.py:137-190 in the factorised form of include/tbx_hip.h (K6):
//   score[h,t] = q_h . k_h[idx_t] + qt_h . e_t          out = [ sum_t a[h,t] v_h[idx_t] | sum_t a[h,t] e_t ]
// Both halves are matrix products per source row with M = 4 heads:
//   stage 1   S[t][h]  = sum_c X[t][c] Q[c][h]      X[t] = [k row | e_t] (256 channels), Q = [block-diagonal q | qt]
//   stage 2   O[c'][h] = sum_t Y[t][c'] P[t][h]     Y[t] = [v row | e_t] (256 channels), P = softmax weights
// The VALU form (attn.hip) spends 47 vector instructions per pair on them (profiles/r03_attn_counters.json: the kernel is VALU-
// issue / latency bound, HBM at 12 %). Here v_mfma_f32_16x16x32_bf16 does both: heads padded 4 -> 16 in the N dimension (a
// quarter of the tile is useful - still 4x the fp32 VALU rate per useful product), 32 targets per chunk:
//   * A operand of stage 1 = 16 targets x 32 channels: lane (target m = l & 15, octet kb = l >> 4) holds 8 consecutive channels of
//     ITS target's row - K channels straight from the gathered table row (16-byte loads), embedding channels from its own
//     sin / cos evaluations (lane kb owns 16 of the pair's 64 arguments: x f_i | y f_i | yaw harmonics 1-16 | 17-32). The
//     contraction order of the embedding channels is a free permutation (applied to qt in the B operand as well).
//   * the scores of head h land in lanes (l & 15) == h, targets (l >> 4) * 4 + r: softmax state and stage-2 accumulators of a head
//     live in those 4 lanes; the probabilities ARE the B operand of stage 2 (k-index order chosen to match: no cross-lane move).
//   * stage 2 contracts over targets, so its A operand wants 8 targets of one channel per lane: the V rows and the embedding rows
//     go to a per-wave LDS image [32 targets][144 halfwords] (row-major, padded) and come back through ds_read_b64_tr_b16 (the
//     16-lane transposing read) - two reads + one MFMA per 16 output channels and 32 targets.
//   * online softmax over the chunks with a lazy reference (rescale only when a chunk's maximum exceeds the reference by > 2^24).
// Operands are bf16 (q, qt, K, V, e and the softmax weights rounded to bf16; fp32 accumulation, fp32 softmax): the bf16-ARITHMETIC
// schedule BASELINE configs[1] names; tolerance in tests/test_hip_attn_mfma.py. (Measured and dropped: every product as hi*hi +
// hi*lo + lo*hi of bf16 splits - fp32-class like the tile kernels' LINEAR stages - needs both planes of K, V and e in registers /
// LDS: one workgroup per CU and ~100-200 spilled registers, 66-147 us per launch against the VALU kernel's 46: the fp32-class path
// stays tbx_knarpe_attn_fwd.)
// Results: out [n_rows, ldo >= 640] and row_no_valid exactly as tbx_knarpe_attn_fwd (same layout; different rounding).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/tbx_hip.h"
#include "attn_core.h"
#include "tbx_common.h"

#ifdef TBX_ATTN_CLOCK
// Profiling build only (libtbx_hip_clk.so, tools/attn_mfma_clock.py): wave 0 of workgroup 0 sums the 100 MHz s_memtime ticks of its
// phases: [0] wait for the chunk's V rows + LDS write, [1] embedding + stage 1, [2] K request + softmax, [3] stage 2, [4] chunks,
// [5] row prologue (frequencies, first gathers, query operands), [7] rows
__device__ unsigned long long g_mclk[8];
extern "C" int tbx_debug_attn_mfma_clock(unsigned long long* host_out) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mclk), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_mclk), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#undef TBX_ACLK
#define TBX_ACLK(VAR, DEP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(VAR) : "v"(DEP) : "memory")
#endif

namespace {

using namespace tbx_attn;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int CHUNK = 32;            // targets per chunk = two 16-target MFMA tiles
constexpr int YS = 272;              // halfwords per LDS row: 128 V channels | 128 embedding positions | 16 of padding (544 B = 136 banks:
                                     // the 4 rows of a transposing read land in 4 disjoint 32-byte bank groups)
constexpr int YPLANE = CHUNK * YS;   // halfwords of one [32][272] image
constexpr int WAVES = 4;

struct MArgs {
  const float* qbuf;
  const float* fxy;
  const float* fyaw;
  float* out;
  uint8_t* row_no_valid;
  tbx_attn_seg_t seg[2];
  int ldq, q_off, qt_off, ldo, n_rows, n_src, n_seg, batch_major;
  float scale2;  // log2(e) / sqrt(d_head)
};

// LDS bytes per wave: the [V | embedding] image + the 2 KiB staging area of the next row's qt; per workgroup: + the 4 x 16
// frequency table
struct Geom {
  static constexpr int WAVE_BYTES = YPLANE * 2 + NH * DR * 4;
  static constexpr int BYTES = WAVES * WAVE_BYTES + 64 * 4;
};

// 8 consecutive floats as two 16-byte loads (16-byte alignment is all the ABI asks for)
__device__ __forceinline__ f32x8 load8f(const float* p) {
  const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 cvt8(const f32x8 v) { return __builtin_convertvector(v, bf16x8); }

// channel of the embedding-space vector that contraction position (octet owner kb, i = 0..31) stands for: lane kb evaluates
// cos (i < 16) and sin (i >= 16) of its 16 arguments  x f_i | y f_i | (i + 1) yaw | (i + 17) yaw  (pose_emb.py:50-55 layout)
__device__ __forceinline__ int e_channel(int kb, int i) {
  return kb < 2 ? kb * 32 + i : (i < 16 ? 64 + (kb - 2) * 16 + i : 96 + (kb - 2) * 16 + (i - 16));
}

// 8 channels of a table row as bf16 hi (+ lo), from `base` (wave-uniform: stays in scalar registers) + a 32-bit byte offset (one
// VGPR per request instead of a 64-bit pointer): bf16 tables are exact (lo = 0), fp32 tables are rounded / split
template <bool KV16>
__device__ __forceinline__ bf16x8 row8(const char* base, uint32_t byte_off) {
  if constexpr (KV16) return *(const bf16x8*)(base + byte_off);
  else return cvt8(load8f((const float*)(base + byte_off)));
}

__device__ __forceinline__ f32x4 mma(const bf16x8 a, const bf16x8 b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }

// max / sum over the 4 lanes l, l ^ 16, l ^ 32, l ^ 48 (one head's lanes)
__device__ __forceinline__ float quad_max(float v) {
  float a, b;
  tbx::swap16(v, &a, &b);
  v = fmaxf(a, b);
  tbx::swap32(v, &a, &b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float quad_sum(float v) {
  float a, b;
  tbx::swap16(v, &a, &b);
  v = a + b;
  tbx::swap32(v, &a, &b);
  return a + b;
}

template <bool KV16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void knarpe_attn_mfma_kernel(const MArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m = lane & 15, kb = lane >> 4;
  unsigned char* wbase = lds_raw + wave * Geom::WAVE_BYTES;
  uint16_t* yv = (uint16_t*)wbase;      // [32][YS]: a row = [V channels 0..127 | embedding positions 0..127 | padding]
  uint16_t* ye = yv + D;
  float* qst = (float*)(wbase + YPLANE * 2);           // the NEXT row's qt [4][128] floats, by LDS-DMA
  float* frt = (float*)(lds_raw + WAVES * Geom::WAVE_BYTES);  // [4 octet owners][16] embedding frequencies (workgroup-wide)

  // ---- the lanes' 16 embedding frequencies (the reference's buffers are repeat-interleaved [f0,f0,f1,f1,..]), pre-multiplied by
  // 1 / 2 pi (see below): once per workgroup
  if (threadIdx.x < 64) {
    const int okb = threadIdx.x >> 4, oi = threadIdx.x & 15;
    const float f = okb < 2 ? a.fxy[2 * oi] : a.fyaw[2 * (16 * (okb - 2) + oi)];
    frt[threadIdx.x] = f * 0.15915494309189535f;
  }
  __syncthreads();

  // ---- this wave's rows: the same wave slot of quads blockIdx.x, blockIdx.x + gridDim.x, ... (a persistent launch: <= 2 workgroups
  // per CU). batch_major: a quad = the SAME source token in 4 consecutive batch entries (rollouts of a scene gather the same
  // table rows at about the same time: one L2 fetch, three L1 hits)
  const int n_quads = a.batch_major ? a.n_rows / WAVES : (a.n_rows + WAVES - 1) / WAVES;
  auto row_at = [&](int i) -> int {  // the wave's i-th row, or -1 past its last
    const int q = (int)blockIdx.x + i * (int)gridDim.x;
    if (q >= n_quads) return -1;
    const int r = a.batch_major ? (WAVES * (q / a.n_src) + wave) * a.n_src + q % a.n_src : q * WAVES + wave;
    return r < a.n_rows ? r : -1;
  };
  constexpr int EB = KV16 ? 2 : 4;  // bytes per table element
  const int nch0 = (a.seg[0].k + CHUNK - 1) / CHUNK;
  const int nch = nch0 + (a.n_seg > 1 ? (a.seg[1].k + CHUNK - 1) / CHUNK : 0);

  // ---- the chunk stream (32 targets per chunk; a segment's last chunk is ragged) runs ACROSS the wave's rows, software-pipelined
  // over three levels so that a chunk's arithmetic runs under the gathers of the next - also the next ROW's first: the (index, mask,
  // pose) triple is requested two chunks ahead, the V rows one chunk ahead as soon as this chunk's V registers have gone to LDS,
  // the K rows one chunk ahead as soon as stage 1 has consumed this chunk's; the next row's query operands come by LDS-DMA
  // (qt: 2 KiB into the staging area, no registers) and 8 registers (q) during this row's first chunk. VMEM returns in order.
  struct Meta {
    int j[2];
    uint8_t inv[2];
    float u[2];
  };
  const int ucol = kb < 2 ? kb : 2;  // the lane's embedding arguments come from x | y | yaw of the relative pose
  auto seg_of = [&](int c) -> const tbx_attn_seg_t& { return a.seg[c >= nch0 ? 1 : 0]; };
  auto t0_of = [&](int c) { return (c >= nch0 ? c - nch0 : c) * CHUNK; };
  auto load_meta = [&](int row, int c, Meta& M) {
    const tbx_attn_seg_t& S = seg_of(c);
    const int t0 = t0_of(c);
    const int64_t pbase = (int64_t)row * S.k;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int t = t0 + tl * 16 + m;
      const int64_t pi = pbase + (t < S.k ? t : S.k - 1);
      M.j[tl] = S.idx[pi];
      M.inv[tl] = S.invalid[pi];
      M.u[tl] = S.rel_pose[pi * 3 + ucol];
    }
  };
  auto seg_base = [&](int row, int c) -> const char* {  // the segment's table of this row's batch entry (wave-uniform)
    const tbx_attn_seg_t& S = seg_of(c);
    return (const char*)S.kv + ((int64_t)((row / a.n_src) / S.batch_div) * S.n_tgt * S.ld_kv) * EB;
  };
  auto ok_mask = [&](int c, const Meta& M, int tl) -> uint64_t {  // bits 0..15: validity of the tile's 16 targets
    const tbx_attn_seg_t& S = seg_of(c);
    return __ballot(M.inv[tl] == 0 && t0_of(c) + tl * 16 + m < S.k);
  };
  bf16x8 akh[2][4], vvh[2][4];
  auto load_k = [&](int row, int c, const Meta& M) {
    const tbx_attn_seg_t& S = seg_of(c);
    const char* base = seg_base(row, c);
    const uint32_t ldb = (uint32_t)S.ld_kv * EB, off = (uint32_t)(S.k_off + kb * 8) * EB;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const uint32_t r = (uint32_t)M.j[tl] * ldb + off;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) akh[tl][ks] = row8<KV16>(base, r + ks * 32 * EB);
    }
  };
  // (measured and dropped: V requests re-mapped so that aligned quads of lanes fetch 64 contiguous bytes of one row - the LDS image
  // address is free - 4x fewer L1 accesses, no faster: 26.3 -> 27.3 us per launch at the WOSAC shape)
  auto load_v = [&](int row, int c, const Meta& M) {  // (request q: the target's channels [q * 32, + 32) as 4 lanes x 16 B, like the K rows)
    const tbx_attn_seg_t& S = seg_of(c);
    const char* base = seg_base(row, c);
    const uint32_t ldb = (uint32_t)S.ld_kv * EB, off = (uint32_t)(S.v_off + kb * 8) * EB;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const uint32_t r = (uint32_t)M.j[tl] * ldb + off;
#pragma unroll
      for (int q = 0; q < 4; ++q) vvh[tl][q] = row8<KV16>(base, r + q * 32 * EB);
    }
  };
  // the query side of a row: qt [4][128] floats -> the staging area (2 x 1 KiB LDS-DMA), q's 8 channels of THIS lane's head block
  // (m & 3, octet kb) -> registers
  const uint32_t qst_lds = lds_addr(qst);
  f32x8 qk;
  auto request_q = [&](int row) {
    const float* qrow = a.qbuf + (int64_t)row * a.ldq;
    glds_1k(qrow + a.qt_off + lane * 4, qst_lds);
    glds_1k(qrow + a.qt_off + 256 + lane * 4, qst_lds + 1024u);
    qk = load8f(qrow + a.q_off + (m & (NH - 1)) * DH + kb * 8);
  };

  int i_row = 0, row = row_at(0);
  if (row < 0) return;
  int c = 0;                 // this chunk: (row, c); the next: (row1, c1); the one after: (row2, c2)
  int c1 = nch > 1 ? 1 : 0, row1 = nch > 1 ? row : row_at(1), i1 = nch > 1 ? 0 : 1;
  Meta m1, m2;
  float u_cur[2];
  uint64_t okm_cur[2];
  {
    request_q(row);
    Meta m0;
    load_meta(row, 0, m0);
    if (row1 >= 0) load_meta(row1, c1, m1);
    load_v(row, 0, m0);
    load_k(row, 0, m0);
    u_cur[0] = m0.u[0], u_cur[1] = m0.u[1];
    okm_cur[0] = ok_mask(0, m0, 0), okm_cur[1] = ok_mask(0, m0, 1);
  }
  bf16x8 bqh[8];  // B operands of stage 1: k-steps 0..3 = K channels of head ks (column n = head: only head ks is non-zero),
                          // k-steps 4..7 = embedding positions (kb, i = (ks - 4) * 8 + j) of qt_n
  f32x4 acc[16];  // stage-2 sums: tile nt = channels nt * 16 .. + 16 of [v | e-positions]; lane (head l & 15, rows (l >> 4) * 4 + r)
  float m_run = -INFINITY, l_run = 0.f;
  // LDS image: row r (a target of the chunk) at r * YS halfwords (544 B = 8 banks mod 64: the 8 rows a transposing read's 32 lanes
  // touch land in 8 disjoint 32-byte windows). A ds_write_b128 is served 8 consecutive lanes at a time = rows r .. r + 7 at ONE
  // column: rows r and r + 4 share a window, so rows with bit 2 set keep the two 16-byte halves of every 32-byte block SWAPPED
  // (hsw) - conflict-free writes; the 4 rows of a transposing read share bit 2, so its lanes swap their halves alike
  const int hsw = (m >> 2) & 1;  // writes: of image row (tl * 16 + m)
  const int tr_off = (kb * 4 + (m >> 2)) * YS + ((((m & 3) >> 1) ^ (kb & 1)) * 8 + (m & 1) * 4);  // reads: rows kb * 4 + .., bit 2 = kb & 1

  while (true) {
    // the chunk after the next
    int c2 = c1 + 1, row2 = row1, i2 = i1;
    if (row1 >= 0 && c2 >= nch) c2 = 0, i2 = i1 + 1, row2 = row_at(i2);
#ifdef TBX_ATTN_CLOCK
    unsigned long long k0, k1, k2, k3, k4;
    TBX_ACLK(k0, l_run);
#endif
    if (c == 0) {
      // ---- row start: the row's query operands from the staging area / the prefetched registers (requested a row ago: every
      // older VMEM request has returned once the newest ones - this chunk's K and V rows - have)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const f32x8 z8 = (f32x8){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x8 v = m == ks ? qk : z8;
        bqh[ks] = cvt8(v);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {  // (8 consecutive positions = 8 consecutive channels)
        const f32x8 q8 = load8f(qst + (m & (NH - 1)) * DR + e_channel(kb, ks * 8));
        const f32x8 v = m < NH ? q8 : z8;
        bqh[4 + ks] = cvt8(v);
      }
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      m_run = -INFINITY, l_run = 0.f;
    }
    // ---- this chunk's V rows -> LDS image rows (tl * 16 + m), channels [q * 32 + kb * 8, + 8)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint16_t* dst = yv + (tl * 16 + m) * YS + q * 32 + ((kb ^ hsw) * 8);  // (hsw: the row's 16-byte halves swapped, see tr_off)
        *(bf16x8*)dst = vvh[tl][q];
      }
    }
#ifdef TBX_ATTN_CLOCK
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TBX_ACLK(k1, l_run);
#endif
    // ---- requests (behind the loop head's wait, which is for everything outstanding): the next row's query side during a row's
    // first chunk (the staging area has been read), the index level two chunks ahead, the V rows of the next chunk
    if (c == 0) {
      const int rn = row_at(i_row + 1);
      if (rn >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the staging reads above have returned before the DMA overwrites it)
        request_q(rn);
      }
    }
    if (row2 >= 0) load_meta(row2, c2, m2);
    if (row1 >= 0) load_v(row1, c1, m1);
#ifdef TBX_ATTN_CLOCK
    unsigned long long k1a, k1b = 0;
    TBX_ACLK(k1a, l_run);
#endif
    // ---- stage 1, K half: S[target][head] over the 4 k-steps of the table rows, both tiles (two independent MFMA chains that
    // run under the embedding's VALU work below)
    f32x4 s1[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) s1[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) s1[tl] = mma(akh[tl][ks], bqh[ks], s1[tl]);
    }
    // ---- the lane's 32 embedding values per target (cos | sin of its 16 arguments): A operands of stage 1 + LDS image
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const float u = u_cur[tl];
      f32x8 ec[2], es[2];
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const f32x4 f4 = *(const f32x4*)(frt + kb * 16 + i4 * 4);
#pragma unroll
        for (int i0 = 0; i0 < 4; ++i0) {
          const int i = i4 * 4 + i0;
          float sn, cs;
          // bf16 operands: the hardware's own range reduction of v_sin / v_cos (argument in revolutions, |rev| < 256 - here <= ~80)
          // on a once-rounded product is good to ~5e-5 rad, two orders below the 4e-3 of the bf16 rounding that follows
          const float rev = u * f4[i0];  // (frequencies pre-multiplied by 1 / 2 pi)
          sn = __builtin_amdgcn_sinf(rev), cs = __builtin_amdgcn_cosf(rev);
          ec[i >> 3][i & 7] = cs;
          es[i >> 3][i & 7] = sn;
        }
      }
      const bf16x8 aeh[4] = {cvt8(ec[0]), cvt8(ec[1]), cvt8(es[0]), cvt8(es[1])};
      uint16_t* dst = ye + (tl * 16 + m) * YS + kb * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *(bf16x8*)(dst + ((q ^ hsw) * 8)) = aeh[q];
      }
      // ---- stage 1, embedding half (its own accumulator: a chain of 4, not 8)
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = mma(aeh[ks], bqh[4 + ks], s);
      s1[tl] += s;
#ifdef TBX_ATTN_CLOCK
      if (tl == 0) TBX_ACLK(k1b, s1[0][0]);
#endif
    }
#ifdef TBX_ATTN_CLOCK
    TBX_ACLK(k2, s1[0][0] + s1[1][0]);
#endif
    if (row1 >= 0) load_k(row1, c1, m1);  // (stage 1 has read this chunk's K registers)
    // ---- softmax weights of this lane's 8 (target, head) scores: lane (head l & 15, g = l >> 4) holds targets g * 4 + r of both tiles
    float sc[8];
    float cmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool v0 = (okm_cur[0] >> (kb * 4 + r)) & 1ull, v1 = (okm_cur[1] >> (kb * 4 + r)) & 1ull;
      sc[r] = v0 ? s1[0][r] * a.scale2 : -INFINITY;
      sc[4 + r] = v1 ? s1[1][r] * a.scale2 : -INFINITY;
      cmax = fmaxf(cmax, fmaxf(sc[r], sc[4 + r]));
    }
    cmax = quad_max(cmax);  // the head's maximum over the chunk (uniform over its 4 lanes)
    // lazy reference: the first valid chunk sets it; it moves only when a later chunk exceeds it by more than 24 (fp32 headroom 2^24
    // on top of sums of at most 128 terms); a wave-uniform, practically never taken branch
    const bool first = m_run == -INFINITY;
    const bool jump = !first && cmax - m_run > 24.f;
    if (__builtin_expect(__ballot(jump) != 0ull, 0)) {
      if (jump) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - cmax);
        l_run *= alpha;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) acc[nt] *= alpha;
        m_run = cmax;
      }
    }
    m_run = first ? cmax : m_run;
    f32x8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = sc[j] > -INFINITY ? __builtin_amdgcn_exp2f(sc[j] - m_run) : 0.f;  // (m_run finite whenever a score is)
      l_run += p[j];
    }
    const bf16x8 ph = cvt8(p);
#ifdef TBX_ATTN_CLOCK
    TBX_ACLK(k3, l_run);
#endif
    // ---- stage 2: O[channel][head] += Y[t][channel] P[t][head] over the chunk's 32 targets. A operand of tile nt: lane (channel
    // c = l & 15, g) wants Y[g * 4 + j][nt * 16 + c] (j < 4: tile A rows, j >= 4: tile B rows 16 + ..) - the transposing read
    // hands lane i of a 16-lane group column i of the [4 rows][16 halfwords] block whose row (i >> 2), halfwords 4 * (i & 3) .. + 4
    // it addresses (tools/probes/tr_read_probe.hip)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
      const uint16_t* img = yv + tr_off + nt * 16;
      typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + 16 * YS));
      const bf16x8 yh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
      acc[nt] = mma(yh, ph, acc[nt]);
    }
#ifdef TBX_ATTN_CLOCK
    TBX_ACLK(k4, (acc[0][0] + acc[7][1]) + (acc[8][2] + acc[15][3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_mclk[0] += k1 - k0, g_mclk[1] += k2 - k1, g_mclk[2] += k3 - k2, g_mclk[3] += k4 - k3, g_mclk[4] += 1;
      g_mclk[5] += k1a - k1, g_mclk[6] += k1b - k1a;
    }
#endif
    if (c + 1 == nch) {
      // ---- the row's last chunk: normalise and store. Lane (head h = l & 15 < 4, g) holds channels nt * 16 + g * 4 + r of every
      // tile for ITS head
      const float l_tot = quad_sum(l_run);
      const bool any_valid = m_run > -INFINITY;  // (masks are per target: every head sees the same validity)
      const float inv_l = any_valid ? 1.0f / l_tot : 0.f;
      float* orow = a.out + (int64_t)row * a.ldo;
      if (m < NH) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          if ((nt >> 1) == m) *(f32x4*)(orow + nt * 16 + kb * 4) = acc[nt] * inv_l;  // V channels of head m: [m * 32, m * 32 + 32)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)  // embedding positions (kb' = nt >> 1, i = (nt & 1) * 16 + g * 4 + r) -> their channels
          *(f32x4*)(orow + D + m * DR + e_channel(nt >> 1, (nt & 1) * 16 + kb * 4)) = acc[8 + nt] * inv_l;
      }
      if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
#ifdef TBX_ATTN_CLOCK
      if (blockIdx.x == 0 && threadIdx.x == 0) g_mclk[7] += 1;
#endif
    }
    // ---- rotate: the next chunk becomes this one
    if (row1 < 0) break;
    u_cur[0] = m1.u[0], u_cur[1] = m1.u[1];
    okm_cur[0] = ok_mask(c1, m1, 0), okm_cur[1] = ok_mask(c1, m1, 1);
    m1 = m2;
    row = row1, c = c1, i_row = i1;
    row1 = row2, c1 = c2, i1 = i2;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention at THREE waves per SIMD (knarpe_attn_mfma3_kernel). knarpe_attn_mfma_kernel holds 64 accumulator registers per
// lane of which a quarter are useful (heads padded 4 -> 16 in stage 2) and 32 targets' K / V rows: 250 VGPRs, two waves per SIMD,
// a third of the SIMD's cycles busy (profiles/r04_attn_mfma_counters.json). Here
//   * stage 2 runs on v_mfma_f32_4x4x4_16B_bf16 (16 blocks of [4 heads] x [4 channels] x [4 targets], probed:
//     tools/probes/mfma4x4_probe.hip): block b = lane >> 2 owns channels 4 b .. 4 b + 3 of a 64-channel group, its B operand (4
//     targets of one channel per lane) is ONE transposing read of the [V | embedding] image for all 64 lanes, and the A operand -
//     the 4 heads' weights of those 4 targets - sits, straight out of stage 1, in the four lanes 16 g .. 16 g + 3 = block 4 g: the
//     instruction's A-broadcast (cbsz = 4, abid = 4 g) hands it to all 16 blocks. Every lane's 4 results are useful: 16 accumulator
//     registers (4 channel groups) instead of 64;
//   * a chunk is ONE 16-target tile: 16 + 16 registers of K / V rows, 9 KiB of LDS image per wave;
// => <= 168 VGPRs and 11 KiB of LDS per wave: three workgroups (12 waves) per CU. Same pipeline across chunks and rows.
constexpr int CH3 = 16;             // targets per chunk
constexpr int YS3 = 288;            // halfwords per image row: 128 V | 128 embedding | 32 padding (576 B = 16 banks mod 64: the 4 rows
                                    // of a transposing read land in 4 disjoint 64-byte windows per 32 lanes)
struct Geom3 {
  static constexpr int WAVE_BYTES = CH3 * YS3 * 2 + NH * DR * 4;
  static constexpr int BYTES = WAVES * WAVE_BYTES + 64 * 4;
};
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <bool KV16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void knarpe_attn_mfma3_kernel(const MArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m = lane & 15, kb = lane >> 4;
  unsigned char* wbase = lds_raw + wave * Geom3::WAVE_BYTES;
  uint16_t* img = (uint16_t*)wbase;                           // [16][YS3]
  float* qst = (float*)(wbase + CH3 * YS3 * 2);               // the NEXT row's qt [4][128] floats, by LDS-DMA
  float* frt = (float*)(lds_raw + WAVES * Geom3::WAVE_BYTES);  // [4 octet owners][16] frequencies / 2 pi (workgroup-wide)
  if (threadIdx.x < 64) {
    const int okb = threadIdx.x >> 4, oi = threadIdx.x & 15;
    const float f = okb < 2 ? a.fxy[2 * oi] : a.fyaw[2 * (16 * (okb - 2) + oi)];
    frt[threadIdx.x] = f * 0.15915494309189535f;
  }
  __syncthreads();
  const int n_quads = a.batch_major ? a.n_rows / WAVES : (a.n_rows + WAVES - 1) / WAVES;
  auto row_at = [&](int i) -> int {
    const int q = (int)blockIdx.x + i * (int)gridDim.x;
    if (q >= n_quads) return -1;
    const int r = a.batch_major ? (WAVES * (q / a.n_src) + wave) * a.n_src + q % a.n_src : q * WAVES + wave;
    return r < a.n_rows ? r : -1;
  };
  constexpr int EB = KV16 ? 2 : 4;
  const int nch0 = (a.seg[0].k + CH3 - 1) / CH3;
  const int nch = nch0 + (a.n_seg > 1 ? (a.seg[1].k + CH3 - 1) / CH3 : 0);
  struct Meta {
    int j;
    uint8_t inv;
    float u;
  };
  const int ucol = kb < 2 ? kb : 2;
  auto seg_of = [&](int c) -> const tbx_attn_seg_t& { return a.seg[c >= nch0 ? 1 : 0]; };
  auto t0_of = [&](int c) { return (c >= nch0 ? c - nch0 : c) * CH3; };
  auto load_meta = [&](int row, int c, Meta& M) {
    const tbx_attn_seg_t& S = seg_of(c);
    const int t = t0_of(c) + m;
    const int64_t pi = (int64_t)row * S.k + (t < S.k ? t : S.k - 1);
    M.j = S.idx[pi];
    M.inv = S.invalid[pi];
    M.u = S.rel_pose[pi * 3 + ucol];
  };
  auto seg_base = [&](int row, int c) -> const char* {
    const tbx_attn_seg_t& S = seg_of(c);
    return (const char*)S.kv + ((int64_t)((row / a.n_src) / S.batch_div) * S.n_tgt * S.ld_kv) * EB;
  };
  auto ok_mask = [&](int c, const Meta& M) -> uint64_t { return __ballot(M.inv == 0 && t0_of(c) + m < seg_of(c).k); };
  bf16x8 akh[4], vvh[4];
  auto load_k = [&](int row, int c, const Meta& M) {
    const tbx_attn_seg_t& S = seg_of(c);
    const char* base = seg_base(row, c);
    const uint32_t r = (uint32_t)M.j * ((uint32_t)S.ld_kv * EB) + (uint32_t)(S.k_off + kb * 8) * EB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) akh[ks] = row8<KV16>(base, r + ks * 32 * EB);
  };
  auto load_v = [&](int row, int c, const Meta& M) {
    const tbx_attn_seg_t& S = seg_of(c);
    const char* base = seg_base(row, c);
    const uint32_t r = (uint32_t)M.j * ((uint32_t)S.ld_kv * EB) + (uint32_t)(S.v_off + kb * 8) * EB;
#pragma unroll
    for (int q = 0; q < 4; ++q) vvh[q] = row8<KV16>(base, r + q * 32 * EB);
  };
  const uint32_t qst_lds = lds_addr(qst);
  f32x8 qk;
  auto request_q = [&](int row) {
    const float* qrow = a.qbuf + (int64_t)row * a.ldq;
    glds_1k(qrow + a.qt_off + lane * 4, qst_lds);
    glds_1k(qrow + a.qt_off + 256 + lane * 4, qst_lds + 1024u);
    qk = load8f(qrow + a.q_off + (m & (NH - 1)) * DH + kb * 8);
  };

  int i_row = 0, row = row_at(0);
  if (row < 0) return;
  int c = 0;
  int c1 = nch > 1 ? 1 : 0, row1 = nch > 1 ? row : row_at(1), i1 = nch > 1 ? 0 : 1;
  Meta m1, m2;
  float u_cur;
  uint64_t okm_cur;
  {
    request_q(row);
    Meta m0;
    load_meta(row, 0, m0);
    if (row1 >= 0) load_meta(row1, c1, m1);
    load_v(row, 0, m0);
    load_k(row, 0, m0);
    u_cur = m0.u;
    okm_cur = ok_mask(0, m0);
  }
  bf16x8 bqh[8];
  f32x4 acc[4];  // lane l, channel group cg: O[head 0..3][channel cg * 64 + l] of [v | e-positions]
  float m_run = -INFINITY, l_run = 0.f;
  // LDS image: 16-byte piece p of row r is stored at piece p ^ ((r >> 1) & 3) (the low two bits): the 8 rows a ds_write_b128 group
  // touches at one column land in 8 distinct 4-bank windows (rows alternate between two 64-byte windows at 576-byte rows); a
  // transposing read's lanes address the same swizzle per row
  const int sw_w = (m >> 1) & 3;                    // writes: image row m
  const int grp = kb, li = m;                       // transposing read: 16-lane group, lane in group
  const int rr = li >> 2;                           // its row within the 4-row block
  const int hq = (grp * 16 + 4 * (li & 3)) >> 3;    // logical piece within a 64-channel group (0..7), + 8 per channel group
  const int ho = 4 * (li & 1);                      // halfword offset inside the piece

  while (true) {
    int c2 = c1 + 1, row2 = row1, i2 = i1;
    if (row1 >= 0 && c2 >= nch) c2 = 0, i2 = i1 + 1, row2 = row_at(i2);
    if (c == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const f32x8 z8 = (f32x8){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bqh[ks] = cvt8(m == ks ? qk : z8);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x8 q8 = load8f(qst + (m & (NH - 1)) * DR + e_channel(kb, ks * 8));
        bqh[4 + ks] = cvt8(m < NH ? q8 : z8);
      }
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) acc[cg] = (f32x4){0.f, 0.f, 0.f, 0.f};
      m_run = -INFINITY, l_run = 0.f;
    }
    // ---- this chunk's V rows -> image row m, logical pieces q * 4 + kb
#pragma unroll
    for (int q = 0; q < 4; ++q) *(bf16x8*)(img + m * YS3 + (q * 4 + (kb ^ sw_w)) * 8) = vvh[q];
    if (c == 0) {
      const int rn = row_at(i_row + 1);
      if (rn >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        request_q(rn);
      }
    }
    if (row2 >= 0) load_meta(row2, c2, m2);
    if (row1 >= 0) load_v(row1, c1, m1);
    // ---- stage 1: K half, then the embedding (16 sincos per lane) and its half
    f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s1 = mma(akh[ks], bqh[ks], s1);
    {
      // (8 arguments at a time - cos octet, then sin octet - so that only 16 fp32 temporaries are live beside the K / V registers)
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const f32x4 fa = *(const f32x4*)(frt + kb * 16 + h2 * 8), fb = *(const f32x4*)(frt + kb * 16 + h2 * 8 + 4);
        f32x8 rev, t8;
#pragma unroll
        for (int i0 = 0; i0 < 4; ++i0) rev[i0] = u_cur * fa[i0], rev[4 + i0] = u_cur * fb[i0];
#pragma unroll
        for (int i0 = 0; i0 < 8; ++i0) t8[i0] = __builtin_amdgcn_cosf(rev[i0]);
        const bf16x8 ac = cvt8(t8);
        *(bf16x8*)(img + m * YS3 + (16 + kb * 4 + (h2 ^ sw_w)) * 8) = ac;          // positions i = h2 * 8 .. + 8 (cos)
        s = mma(ac, bqh[4 + h2], s);
#pragma unroll
        for (int i0 = 0; i0 < 8; ++i0) t8[i0] = __builtin_amdgcn_sinf(rev[i0]);
        const bf16x8 as = cvt8(t8);
        *(bf16x8*)(img + m * YS3 + (16 + kb * 4 + ((2 + h2) ^ sw_w)) * 8) = as;    // positions i = 16 + h2 * 8 .. + 8 (sin)
        s = mma(as, bqh[6 + h2], s);
      }
      s1 += s;
    }
    if (row1 >= 0) load_k(row1, c1, m1);
    // ---- softmax weights: lane (head l & 15, g = l >> 4) holds targets g * 4 + r
    float sc[4];
    float cmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[r] = ((okm_cur >> (kb * 4 + r)) & 1ull) ? s1[r] * a.scale2 : -INFINITY;
      cmax = fmaxf(cmax, sc[r]);
    }
    cmax = quad_max(cmax);
    const bool first = m_run == -INFINITY;
    const bool jump = !first && cmax - m_run > 24.f;
    if (__builtin_expect(__ballot(jump && m < NH) != 0ull, 0)) {
      // (the accumulators of head i live in register i of EVERY lane here: the rescale factor of each head comes from its lanes)
      float alpha = jump ? __builtin_amdgcn_exp2f(m_run - cmax) : 1.f;
      if (jump) l_run *= alpha, m_run = cmax;
      float al[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) al[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), h));
#pragma unroll
      for (int cg = 0; cg < 4; ++cg)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[cg][h] *= al[h];
    }
    m_run = first ? cmax : m_run;
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = sc[r] > -INFINITY ? __builtin_amdgcn_exp2f(sc[r] - m_run) : 0.f;
      l_run += p[r];
    }
    const bf16x4 pb = {(__bf16)p[0], (__bf16)p[1], (__bf16)p[2], (__bf16)p[3]};
    const s16x4 pa = __builtin_bit_cast(s16x4, pb);
    // ---- stage 2: per 4 targets (k-step g: the weights in lanes 16 g .. 16 g + 3 = block 4 g, broadcast) and 64-channel group
#define TBX_STAGE2(G)                                                                                                              \
  _Pragma("unroll") for (int cg = 0; cg < 4; ++cg) {                                                                               \
    const int r_ = 4 * (G) + rr;                                                                                                   \
    const uint16_t* src_ = img + r_ * YS3 + (((cg * 8 + hq) ^ ((r_ >> 1) & 3)) * 8) + ho;                                          \
    const s16x4 y_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)src_);                                                  \
    acc[cg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(pa, y_, acc[cg], 4, 4 * (G), 0);                                              \
  }
    TBX_STAGE2(0)
    TBX_STAGE2(1)
    TBX_STAGE2(2)
    TBX_STAGE2(3)
#undef TBX_STAGE2
    if (c + 1 == nch) {
      // ---- the row's last chunk: normalise and store. l_run / m_run of head h live in lanes (l & 15) == h
      const float l_tot = quad_sum(l_run);
      const bool any_valid = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m_run), 0)) > -INFINITY;
      const float inv_mine = any_valid ? 1.0f / l_tot : 0.f;
      float inv_l[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) inv_l[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv_mine), h));
      float* orow = a.out + (int64_t)row * a.ldo;
      // V channels c = cg * 64 + lane (cg = 0, 1) belong to head c >> 5: each lane keeps its head's register
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const int hc = (cg * 64 + lane) >> 5;
        const float v = hc == 0 ? acc[cg][0] * inv_l[0] : (hc == 1 ? acc[cg][1] * inv_l[1] : (hc == 2 ? acc[cg][2] * inv_l[2] : acc[cg][3] * inv_l[3]));
        orow[cg * 64 + lane] = v;
      }
      // embedding positions p = (cg - 2) * 64 + lane = (octet owner p >> 5, i = p & 31) -> their channels, all 4 heads
#pragma unroll
      for (int cg = 2; cg < 4; ++cg) {
        const int pp = (cg - 2) * 64 + lane;
        const int ch = e_channel(pp >> 5, pp & 31);
#pragma unroll
        for (int h = 0; h < NH; ++h) orow[D + h * DR + ch] = acc[cg][h] * inv_l[h];
      }
      if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
    }
    if (row1 < 0) break;
    u_cur = m1.u;
    okm_cur = ok_mask(c1, m1);
    m1 = m2;
    row = row1, c = c1, i_row = i1;
    row1 = row2, c1 = c2, i1 = i2;
  }
}

}  // namespace

extern "C" int tbx_knarpe_attn_fwd_mfma(const float* qbuf, int ldq, int q_off, int qt_off, int n_batch, int n_src,
                                        const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                                        const float* freqs_xy, const float* freqs_yaw, void* stream) {
  if (!qbuf || !segs || !out || !row_no_valid || !freqs_xy || !freqs_yaw || n_batch <= 0 || n_src <= 0) return TBX_ERR_ARG;
  if (n_seg < 1 || n_seg > 2 || ldo < D + NH * DR) return TBX_ERR_UNSUPPORTED;
  if ((ldq % 4) || (q_off % 4) || (qt_off % 4) || (ldo % 4) || (((uintptr_t)qbuf) & 15) || (((uintptr_t)out) & 15)) return TBX_ERR_ALIGN;
  MArgs a;
  int ktot = 0;
  for (int i = 0; i < n_seg; ++i) {
    const tbx_attn_seg_t& s = segs[i];
    if (!s.kv || !s.idx || !s.invalid || !s.rel_pose || s.k <= 0 || s.n_tgt <= 0 || s.batch_div <= 0) return TBX_ERR_ARG;
    if (s.emb != nullptr) return TBX_ERR_UNSUPPORTED;  // relative-pose form only
    if ((s.ld_kv % 8) || (s.k_off % 8) || (s.v_off % 8) || (((uintptr_t)s.kv) & 15)) return TBX_ERR_ALIGN;
    if ((s.kv_bf16 != 0) != (segs[0].kv_bf16 != 0)) return TBX_ERR_UNSUPPORTED;
    if ((int64_t)s.n_tgt * s.ld_kv * 4 >= (1ll << 32)) return TBX_ERR_UNSUPPORTED;  // (32-bit byte offsets inside a batch entry's table)
    ktot += s.k;
    a.seg[i] = s;
  }
  if (n_seg == 1) a.seg[1] = a.seg[0];
  if (ktot > KMAX) return TBX_ERR_UNSUPPORTED;
  a.qbuf = qbuf, a.fxy = freqs_xy, a.fyaw = freqs_yaw, a.out = out, a.row_no_valid = row_no_valid;
  a.ldq = ldq, a.q_off = q_off, a.qt_off = qt_off, a.ldo = ldo, a.n_rows = n_batch * n_src, a.n_src = n_src, a.n_seg = n_seg;
  a.scale2 = 1.4426950408889634f / sqrtf((float)DH);
  bool shared = false;
  for (int i = 0; i < n_seg; ++i) shared = shared || segs[i].batch_div > 1;
  a.batch_major = (shared && n_batch % WAVES == 0) ? 1 : 0;
  // persistent: at most 2 workgroups per CU (77 KiB of LDS each, 2 waves per SIMD), each wave walks its slot of quads blockIdx.x + i * grid
  static const int max_wg = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const char* e = getenv("TBX_ATTN_MFMA_WG_PER_CU");
    return (e && atoi(e) > 0 ? atoi(e) : 2) * (cus > 0 ? cus : 256);
  }();
  const int n_quads = (a.n_rows + WAVES - 1) / WAVES;
  const dim3 grid((unsigned)(n_quads < max_wg ? n_quads : max_wg)), block(WAVES * 64);
  hipStream_t hs = (hipStream_t)stream;
  const bool kv16 = segs[0].kv_bf16 != 0;
#define TBX_MFMA_LAUNCH(KV16F)                                                                                                   \
  do {                                                                                                                           \
    static tbx::PerDeviceOnce lds_attr;                                                                                          \
    if (!lds_attr([&] {                                                                                                          \
          return hipFuncSetAttribute((const void*)knarpe_attn_mfma_kernel<KV16F>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                     Geom::BYTES) == hipSuccess;                                                                 \
        }))                                                                                                                      \
      return TBX_ERR_LAUNCH;                                                                                                     \
    hipLaunchKernelGGL((knarpe_attn_mfma_kernel<KV16F>), grid, block, Geom::BYTES, hs, a);                                       \
  } while (0)
  static const int v3 = [] { const char* e = getenv("TBX_ATTN_MFMA_V3"); return e ? atoi(e) : 0; }();
  if (v3) {
    static const int max_wg3 = [] {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return 3 * (cus > 0 ? cus : 256);
    }();
    const dim3 grid3((unsigned)(n_quads < max_wg3 ? n_quads : max_wg3));
#define TBX_MFMA3_LAUNCH(KV16F)                                                                                                  \
  do {                                                                                                                           \
    static tbx::PerDeviceOnce lds_attr3;                                                                                         \
    if (!lds_attr3([&] {                                                                                                         \
          return hipFuncSetAttribute((const void*)knarpe_attn_mfma3_kernel<KV16F>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                     Geom3::BYTES) == hipSuccess;                                                                \
        }))                                                                                                                      \
      return TBX_ERR_LAUNCH;                                                                                                     \
    hipLaunchKernelGGL((knarpe_attn_mfma3_kernel<KV16F>), grid3, block, Geom3::BYTES, hs, a);                                    \
  } while (0)
    if (kv16) TBX_MFMA3_LAUNCH(true);
    else TBX_MFMA3_LAUNCH(false);
#undef TBX_MFMA3_LAUNCH
    return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
  }
  if (kv16) TBX_MFMA_LAUNCH(true);
  else TBX_MFMA_LAUNCH(false);
#undef TBX_MFMA_LAUNCH
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
