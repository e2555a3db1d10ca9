// tbx_knarpe_attn_fwd_mfma: the KNARPE attention of LARGE launches (a wavefront per source row) on the bf16 matrix cores.
//
// modules/attention_rpe.py:137-190 in the factorised form of include/tbx_hip.h (K6):
//   score[h,t] = q_h . k_h[idx_t] + qt_h . e_t          out = [ sum_t a[h,t] v_h[idx_t] | sum_t a[h,t] e_t ]
// Both halves are matrix products per source row with M = 4 heads:
//   stage 1   S[t][h]  = sum_c X[t][c] Q[c][h]      X[t] = [k row | e_t] (256 channels), Q = [block-diagonal q | qt]
//   stage 2   O[c'][h] = sum_t Y[t][c'] P[t][h]     Y[t] = [v row | e_t] (256 channels), P = softmax weights
// The VALU form (attn.hip) spends 47 vector instructions per pair on them (profiles/r04_attn_counters.json: VALU-issue / latency
// bound, HBM at 12 %). Here the matrix cores do both, 16 targets per chunk:
//   * stage 1 on v_mfma_f32_16x16x32_bf16 (M = 16 targets, N = heads padded 4 -> 16, 8 k-steps of 32 channels): lane (target m =
//     l & 15, octet kb = l >> 4) holds 8 consecutive channels of ITS target's row - K channels straight from the gathered table row
//     (16-byte loads), embedding channels from its own v_sin / v_cos (lane kb owns 16 of the pair's 64 arguments: x f_i | y f_i | yaw
//     harmonics 1-16 | 17-32). The contraction order of the embedding channels is a free permutation (applied to qt in the B operand
//     as well).
//   * the scores of head h land in lanes (l & 15) == h, targets (l >> 4) * 4 + r: the head's softmax state lives in those 4 lanes,
//     and their 4 probabilities ARE the A operand of stage 2 (no cross-lane move: see the geometry note below).
//   * stage 2 contracts over targets, so its B operand wants 4 targets of one channel per lane: the V rows and the embedding rows go
//     to a per-wave LDS image [16 targets][288 halfwords] (row-major, padded, swizzled) and come back through ds_read_b64_tr_b16 (the
//     16-lane transposing read, probed: tools/probes/tr_read_probe.hip) - one read + one v_mfma_f32_4x4x4_16B_bf16 per 64 output
//     channels and 4 targets.
//   * V requests are re-mapped so that aligned quads of lanes fetch 64 contiguous bytes of one row (the kernel is bound by the L1's
//     lookup rate: one access per lane request otherwise).
//   * persistent waves; the chunk stream runs across a wave's rows with the gathers of the next chunk / row in flight under the
//     current chunk's arithmetic; online softmax with a lazy reference (rescale only when a chunk exceeds it by > 2^24).
// Operands are bf16 (q, qt, K, V, e and the softmax weights rounded to bf16; fp32 accumulation, fp32 softmax): the bf16-ARITHMETIC
// schedule BASELINE configs[1] names; tolerance in tests/test_hip_attn_mfma.py. (Measured and dropped: every product as hi*hi +
// hi*lo + lo*hi of bf16 splits - fp32-class like the tile kernels' LINEAR stages - needs both planes of K, V and e in registers /
// LDS: one workgroup per CU and ~100-200 spilled registers, 66-147 us per launch against the VALU kernel's 46: the fp32-class path
// stays tbx_knarpe_attn_fwd.)
// Results: out [n_rows, ldo >= 640] and row_no_valid exactly as tbx_knarpe_attn_fwd (same layout; different rounding).
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/tbx_hip.h"
#include "attn_core.h"
#include "tbx_common.h"


namespace {

using namespace tbx_attn;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

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
  // dropout on the attention probabilities (training, attention_rpe.py:171-172): the fields and the key of attn.hip's AttnArgs /
  // attn_core.h's DropKey - (seed, call, scene row, closed-loop step, global target slot, head) - so that the backward kernels
  // (tbx_knarpe_attn_bwd_*: they recompute the probabilities and regenerate the mask) drop exactly what this forward dropped
  const uint64_t* drop_seed;
  uint32_t drop_call, drop_thresh;
  float drop_scale;
  int drop_time_batch, drop_time0;
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

// ---------------------------------------------------------------------------------------------------------------------
// Geometry. The first form of this kernel (round 4, git history) ran stage 2 on v_mfma_f32_16x16x32_bf16 too: 64 accumulator
// registers per lane of which a quarter were useful (heads padded 4 -> 16), 32 targets per chunk, 250 VGPRs = two waves per SIMD,
// 26 us per launch at the WOSAC shape. Here
//   * stage 2 runs on v_mfma_f32_4x4x4_16B_bf16 (16 blocks of [4 heads] x [4 channels] x [4 targets], probed:
//     tools/probes/mfma4x4_probe.hip): block b = lane >> 2 owns channels 4 b .. 4 b + 3 of a 64-channel group, its B operand (4
//     targets of one channel per lane) is ONE transposing read of the [V | embedding] image for all 64 lanes, and the A operand -
//     the 4 heads' weights of those 4 targets - sits, straight out of stage 1, in the four lanes 16 g .. 16 g + 3 = block 4 g: the
//     instruction's A-broadcast (cbsz = 4, abid = 4 g) hands it to all 16 blocks. Every lane's 4 results are useful: 16 accumulator
//     registers (4 channel groups) instead of 64;
//   * a chunk is ONE 16-target tile: 16 + 16 registers of K / V rows, 9 KiB of LDS image per wave;
// => <= 168 VGPRs and 11 KiB of LDS per wave: three workgroups (12 waves) per CU.
constexpr int CHUNK = 16;             // targets per chunk
constexpr int YS = 288;            // halfwords per image row: 128 V | 128 embedding | 32 padding (576 B = 16 banks mod 64: the 4 rows
                                    // of a transposing read land in 4 disjoint 64-byte windows per 32 lanes)
struct Geom {
  static constexpr int WAVE_BYTES = CHUNK * YS * 2 + NH * DR * 4;
  static constexpr int BYTES = WAVES * WAVE_BYTES + 64 * 4;
};
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

template <bool KV16, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void knarpe_attn_mfma_kernel(const MArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m = lane & 15, kb = lane >> 4;
  unsigned char* wbase = lds_raw + wave * Geom::WAVE_BYTES;
  uint16_t* img = (uint16_t*)wbase;                           // [16][YS]
  float* qst = (float*)(wbase + CHUNK * YS * 2);               // the NEXT row's qt [4][128] floats, by LDS-DMA
  float* frt = (float*)(lds_raw + WAVES * Geom::WAVE_BYTES);  // [4 octet owners][16] frequencies / 2 pi (workgroup-wide)
  if (threadIdx.x < 64) {
    const int okb = threadIdx.x >> 4, oi = threadIdx.x & 15;
    const float f = okb < 2 ? a.fxy[2 * oi] : a.fyaw[2 * (16 * (okb - 2) + oi)];
    frt[threadIdx.x] = f * 0.15915494309189535f;
  }
  __syncthreads();
  const int n_quads = a.batch_major ? a.n_rows / WAVES : (a.n_rows + WAVES - 1) / WAVES;
  constexpr int EB = KV16 ? 2 : 4;
  // Everything that depends on the ROW only - its index, its batch entry's table of each segment - is formed once per row (three
  // integer divisions), not once per chunk and request: the scalar unit is shared by the CU's 12 waves, and the per-chunk form of
  // this bookkeeping was ~400 scalar instructions per chunk.
  struct RowCtx {
    int i, row, b;          // the wave's i-th row (row < 0: past its last), its batch entry
    const char* kvb[2];     // the segments' tables of this row's batch entry
  };
  auto row_ctx = [&](int i) -> RowCtx {
    RowCtx r;
    r.i = i, r.row = -1, r.b = 0, r.kvb[0] = r.kvb[1] = nullptr;
    // (an XCD-contiguous walk - XCD x's workgroups on a contiguous eighth of the quads, as attn.hip's forward kernels do - was measured in
    //  round 6: 32 x 128 agents 10.50 -> 10.43 M agent-steps/s, the training step + 0.4 ms, and its index arithmetic cost 2 registers
    //  of a kernel that lives at the 168-register limit of three waves per SIMD; removed)
    const int q = (int)blockIdx.x + i * (int)gridDim.x;
    if (q >= n_quads) return r;
    int b, row;
    if (a.batch_major) {
      const int qb = q / a.n_src;
      b = WAVES * qb + wave, row = b * a.n_src + (q - qb * a.n_src);
    } else {
      row = q * WAVES + wave, b = row / a.n_src;
    }
    if (row >= a.n_rows) return r;
    r.row = row, r.b = b;
    r.kvb[0] = (const char*)a.seg[0].kv + ((int64_t)(b / a.seg[0].batch_div) * a.seg[0].n_tgt * a.seg[0].ld_kv) * EB;
    r.kvb[1] = (const char*)a.seg[1].kv + ((int64_t)(b / a.seg[1].batch_div) * a.seg[1].n_tgt * a.seg[1].ld_kv) * EB;
    return r;
  };
  const int nch0 = (a.seg[0].k + CHUNK - 1) / CHUNK;
  const int nch = nch0 + (a.n_seg > 1 ? (a.seg[1].k + CHUNK - 1) / CHUNK : 0);
  struct Meta {
    int j;
    uint8_t inv;
    float u;
  };
  const int ucol = kb < 2 ? kb : 2;
  auto seg_of = [&](int c) -> const tbx_attn_seg_t& { return a.seg[c >= nch0 ? 1 : 0]; };
  auto t0_of = [&](int c) { return (c >= nch0 ? c - nch0 : c) * CHUNK; };
  auto load_meta = [&](const RowCtx& R, int c, Meta& M) {
    const tbx_attn_seg_t& S = seg_of(c);
    const int t = t0_of(c) + m;
    const int64_t pi = (int64_t)R.row * S.k + (t < S.k ? t : S.k - 1);
    M.j = S.idx[pi];
    M.inv = S.invalid[pi];
    M.u = S.rel_pose[pi * 3 + ucol];
  };
  auto seg_base = [&](const RowCtx& R, int c) -> const char* { return c >= nch0 ? R.kvb[1] : R.kvb[0]; };
  auto ok_mask = [&](int c, const Meta& M) -> uint64_t { return __ballot(M.inv == 0 && t0_of(c) + m < seg_of(c).k); };
  bf16x8 akh[4], vvh[4];
  // V requests, coalesced: request q fetches, per aligned quad of lanes (m & 3 = 0..3, one kb), 64 CONTIGUOUS bytes of ONE target's
  // row - target 4 (m >> 2) + q of the tile, 16-byte pieces 4 kb + (m & 3) - so the quad shares one L1 access instead of making
  // four (the kernel is bound by the L1's lookup rate: 9.15 M accesses per launch = one per lane request,
  // profiles/r04_attn_mfma_counters.json): 25.8 -> 22.6 us per launch. The V data only goes to the LDS image, whose address is free.
  // (The K rows the same way - parked in the image's embedding half and read back as stage 1's A operands, 4 + 4 more LDS
  // instructions - measured 23.1 us: no gain over V alone; K keeps its direct operand-layout requests.)
  auto quad_bcast = [&](int v, auto Q) -> int { return __builtin_amdgcn_update_dpp(0, v, decltype(Q)::value * 0x55, 0xf, 0xf, true); };
  auto load_rows = [&](const RowCtx& R, int c, const Meta& M, int col_off, bf16x8 (&dst)[4]) {
    const tbx_attn_seg_t& S = seg_of(c);
    const char* base = seg_base(R, c);
    const uint32_t ldb = (uint32_t)S.ld_kv * EB, off = (uint32_t)(col_off + (4 * kb + (m & 3)) * 8) * EB;
    const int jq[4] = {quad_bcast(M.j, std::integral_constant<int, 0>()), quad_bcast(M.j, std::integral_constant<int, 1>()),
                       quad_bcast(M.j, std::integral_constant<int, 2>()), quad_bcast(M.j, std::integral_constant<int, 3>())};
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = row8<KV16>(base, (uint32_t)jq[q] * ldb + off);
  };
  auto load_v = [&](const RowCtx& R, int c, const Meta& M) { load_rows(R, c, M, seg_of(c).v_off, vvh); };
  auto load_k = [&](const RowCtx& R, int c, const Meta& M) {  // (straight into stage 1's A-operand layout: row m, channels ks * 32 + kb * 8 ..)
    const tbx_attn_seg_t& S = seg_of(c);
    const char* base = seg_base(R, c);
    const uint32_t r = (uint32_t)M.j * ((uint32_t)S.ld_kv * EB) + (uint32_t)(S.k_off + kb * 8) * EB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) akh[ks] = row8<KV16>(base, r + ks * 32 * EB);
  };
  const uint32_t qst_lds = lds_addr(qst);
  f32x8 qk;
  auto request_q = [&](int row) {
    const float* qrow = a.qbuf + (int64_t)row * a.ldq;
    glds_1k(qrow + a.qt_off + lane * 4, qst_lds);
    glds_1k(qrow + a.qt_off + 256 + lane * 4, qst_lds + 1024u);
    qk = load8f(qrow + a.q_off + (m & (NH - 1)) * DH + kb * 8);
  };

  RowCtx R = row_ctx(0);  // this chunk: (R, c); the next: (R1, c1); the one after: (R2, c2)
  if (R.row < 0) return;
  int c = 0;
  int c1 = nch > 1 ? 1 : 0;
  RowCtx R1 = nch > 1 ? R : row_ctx(1);
  Meta m1, m2;
  float u_cur;
  uint64_t okm_cur;
  {
    request_q(R.row);
    Meta m0;
    load_meta(R, 0, m0);
    if (R1.row >= 0) load_meta(R1, c1, m1);
    load_v(R, 0, m0);
    load_k(R, 0, m0);
    u_cur = m0.u;
    okm_cur = ok_mask(0, m0);
  }
  bf16x8 bqh[8];
  f32x4 acc[4];  // lane l, channel group cg: O[head 0..3][channel cg * 64 + l] of [v | e-positions]
  float m_run = -INFINITY, l_run = 0.f;
  DropKey dk;
  dk.lo = dk.hi = dk.krow = 0u;
  // LDS image: 16-byte piece p of row r is stored at piece p ^ ((r >> 1) & 3) (the low two bits): the 8 rows a ds_write_b128 group
  // touches at one column land in 8 distinct 4-bank windows (rows alternate between two 64-byte windows at 576-byte rows); a
  // transposing read's lanes address the same swizzle per row
  const int sw_w = (m >> 1) & 3;                    // writes: image row m
  const int grp = kb, li = m;                       // transposing read: 16-lane group, lane in group
  const int rr = li >> 2;                           // its row within the 4-row block
  const int hq = (grp * 16 + 4 * (li & 3)) >> 3;    // logical piece within a 64-channel group (0..7), + 8 per channel group
  const int ho = 4 * (li & 1);                      // halfword offset inside the piece

  while (true) {
    int c2 = c1 + 1;
    RowCtx R2 = R1;
    if (R1.row >= 0 && c2 >= nch) c2 = 0, R2 = row_ctx(R1.i + 1);
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
      if constexpr (DROP) dk.init(a, R.row, R.b);
    }
    // ---- this chunk's V pieces -> the image: register q holds logical piece 4 kb + (m & 3) of row 4 (m >> 2) + q
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r_ = (m & 12) + q;
      *(bf16x8*)(img + r_ * YS + (4 * kb + ((m & 3) ^ ((r_ >> 1) & 3))) * 8) = vvh[q];
    }
    if (c == 0) {
      // (the next ROW's query side: that row is R1's or R2's when the row has one or two chunks, else it is formed here)
      const int rn = nch == 1 ? R1.row : (nch == 2 ? R2.row : row_ctx(R.i + 1).row);
      if (rn >= 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        request_q(rn);
      }
    }
    if (R2.row >= 0) load_meta(R2, c2, m2);
    if (R1.row >= 0) load_v(R1, c1, m1);
    // ---- stage 1: K half, then the embedding (16 sincos per lane) and its half
    f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s1 = mma(akh[ks], bqh[ks], s1);
    if (R1.row >= 0) load_k(R1, c1, m1);  // (stage 1 has read this chunk's K registers)
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
        *(bf16x8*)(img + m * YS + (16 + kb * 4 + (h2 ^ sw_w)) * 8) = ac;          // positions i = h2 * 8 .. + 8 (cos)
        s = mma(ac, bqh[4 + h2], s);
#pragma unroll
        for (int i0 = 0; i0 < 8; ++i0) t8[i0] = __builtin_amdgcn_sinf(rev[i0]);
        const bf16x8 as = cvt8(t8);
        *(bf16x8*)(img + m * YS + (16 + kb * 4 + ((2 + h2) ^ sw_w)) * 8) = as;    // positions i = 16 + h2 * 8 .. + 8 (sin)
        s = mma(as, bqh[6 + h2], s);
      }
      s1 += s;
    }
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
      l_run += p[r];  // (the normaliser is that of the un-dropped softmax)
    }
    if constexpr (DROP) {
      // the lane's 4 weights are targets kb * 4 + r of this chunk for head m (lanes m >= 4 hold padding): global slot = targets of
      // the earlier segment + the chunk's first target + kb * 4 + r, as attn_core.h's sweep counts them
      const uint32_t tg = (uint32_t)((c >= nch0 ? a.seg[0].k + (c - nch0) * CHUNK : c * CHUNK) + kb * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = dk.keep(tg + (uint32_t)r, (uint32_t)(m & (NH - 1)), a.drop_thresh) ? p[r] * a.drop_scale : 0.f;
    }
    const bf16x4 pb = {(__bf16)p[0], (__bf16)p[1], (__bf16)p[2], (__bf16)p[3]};
    const s16x4 pa = __builtin_bit_cast(s16x4, pb);
    // ---- stage 2: per 4 targets (k-step g: the weights in lanes 16 g .. 16 g + 3 = block 4 g, broadcast) and 64-channel group
#define TBX_STAGE2(G)                                                                                                              \
  _Pragma("unroll") for (int cg = 0; cg < 4; ++cg) {                                                                               \
    const int r_ = 4 * (G) + rr;                                                                                                   \
    const uint16_t* src_ = img + r_ * YS + (((cg * 8 + hq) ^ ((r_ >> 1) & 3)) * 8) + ho;                                          \
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
      float* orow = a.out + (int64_t)R.row * a.ldo;
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
      if (lane == 0) a.row_no_valid[R.row] = any_valid ? 0 : 1;
    }
    if (R1.row < 0) break;
    u_cur = m1.u;
    okm_cur = ok_mask(c1, m1);
    m1 = m2;
    R = R1, c = c1;
    R1 = R2, c1 = c2;
  }
}

}  // namespace

static int mfma_launch(const float* qbuf, int ldq, int q_off, int qt_off, int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg,
                       float* out, int ldo, uint8_t* row_no_valid, const float* freqs_xy, const float* freqs_yaw, float p_drop,
                       const uint64_t* drop_seed, uint32_t drop_call, int time_batch, int time0, void* stream) {
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
  a.drop_seed = drop_seed, a.drop_call = drop_call, a.drop_thresh = 0u, a.drop_scale = 1.f, a.drop_time_batch = time_batch, a.drop_time0 = time0;
  if (p_drop < 0.f || p_drop >= 1.f || time_batch < 1 || time0 < 0) return TBX_ERR_ARG;
  if (p_drop > 0.f) {  // (threshold and scale exactly as attn.hip's set_dropout: the backward regenerates the mask from them)
    if (!drop_seed) return TBX_ERR_ARG;
    const double th = (double)p_drop * 4294967296.0;
    a.drop_thresh = th < 1.0 ? 1u : (uint32_t)th;
    a.drop_scale = 1.0f / (1.0f - p_drop);
  }
  const bool drop = a.drop_thresh != 0u;
  bool shared = false;
  for (int i = 0; i < n_seg; ++i) shared = shared || segs[i].batch_div > 1;
  a.batch_major = (shared && n_batch % WAVES == 0) ? 1 : 0;
  // persistent: at most 3 workgroups per CU (44 KiB of LDS each, 3 waves per SIMD); each wave walks its slot of quads blockIdx.x + i * grid
  // (cached per device ordinal: a process that drives several devices must not size device 1's grid by device 0's CU count)
  static std::atomic<int> wg_cap[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int max_wg = wg_cap[dev].load(std::memory_order_relaxed);
  if (max_wg <= 0) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const char* e = getenv("TBX_ATTN_MFMA_WG_PER_CU");
    max_wg = (e && atoi(e) > 0 ? atoi(e) : 3) * (cus > 0 ? cus : 256);
    wg_cap[dev].store(max_wg, std::memory_order_relaxed);
  }
  const int n_quads = (a.n_rows + WAVES - 1) / WAVES;
  const dim3 grid((unsigned)(n_quads < max_wg ? n_quads : max_wg)), block(WAVES * 64);
  hipStream_t hs = (hipStream_t)stream;
  const bool kv16 = segs[0].kv_bf16 != 0;
#define TBX_MFMA_LAUNCH(KV16F, DROPF)                                                                                            \
  do {                                                                                                                           \
    static tbx::PerDeviceOnce lds_attr;                                                                                          \
    if (!lds_attr([&] {                                                                                                          \
          return hipFuncSetAttribute((const void*)knarpe_attn_mfma_kernel<KV16F, DROPF>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     Geom::BYTES) == hipSuccess;                                                                \
        }))                                                                                                                      \
      return TBX_ERR_LAUNCH;                                                                                                     \
    hipLaunchKernelGGL((knarpe_attn_mfma_kernel<KV16F, DROPF>), grid, block, Geom::BYTES, hs, a);                               \
  } while (0)
  if (kv16 && drop) TBX_MFMA_LAUNCH(true, true);
  else if (kv16) TBX_MFMA_LAUNCH(true, false);
  else if (drop) TBX_MFMA_LAUNCH(false, true);
  else TBX_MFMA_LAUNCH(false, false);
#undef TBX_MFMA_LAUNCH
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_knarpe_attn_fwd_mfma(const float* qbuf, int ldq, int q_off, int qt_off, int n_batch, int n_src,
                                        const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                                        const float* freqs_xy, const float* freqs_yaw, void* stream) {
  return mfma_launch(qbuf, ldq, q_off, qt_off, n_batch, n_src, segs, n_seg, out, ldo, row_no_valid, freqs_xy, freqs_yaw, 0.f, nullptr, 0u, 1, 0,
                     stream);
}

extern "C" int tbx_knarpe_attn_fwd_mfma_dropout_tb(const float* qbuf, int ldq, int q_off, int qt_off, int n_batch, int n_src,
                                                   const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                                                   const float* freqs_xy, const float* freqs_yaw, float p_drop, const uint64_t* drop_seed,
                                                   uint32_t drop_call, int time_batch, int time0, void* stream) {
  return mfma_launch(qbuf, ldq, q_off, qt_off, n_batch, n_src, segs, n_seg, out, ldo, row_no_valid, freqs_xy, freqs_yaw, p_drop, drop_seed,
                     drop_call, time_batch, time0, stream);
}
