// tbx_knarpe_attn_fwd / tbx_knarpe_attn_bwd: fused KNARPE attention (see include/tbx_hip.h for the math and layouts).
//
// One wavefront per source token, 4 tokens per 256-thread workgroup (large grids), or 4 wavefronts per token (small grids).
// Forward: single pass with an online softmax: per (src, tgt) pair it reads the K row and the V row exactly once (full
// 128-B lines per 8-lane group), 4 B of index, 1 B of mask and EITHER the 512-B materialised pose embedding OR the 12-B
// relative pose, from which the 128-d embedding is rebuilt in registers (one sincosf per (cos, sin) channel pair: the
// kernel then moves exactly the algorithmic 1041 B per pair). Masked targets contribute nothing; rows without any valid
// target are written as zeros and flagged. d_model 128, 4 heads of 32, d_rpe 128.
//
// Channel ownership inside an 8-lane target group (lane slot s8 = lane & 7):
//   K / V / q / out[0:128]   : 4 float4 at channels st*32 + s8*4 (st = 0..3 is also the head of that channel block)
//   embedding e / qt / E-sums : the (cos, sin) pairs of 8 arguments -> 16 channels:
//       x f_i, i in {2s8, 2s8+1}   -> cos at 2s8+.., sin at 16+2s8+..      y f_i likewise at 32+.. / 48+..
//       k yaw, k-1 in {4s8..4s8+3} -> cos at 64+4s8+.., sin at 96+4s8+..
//   (pose_emb.py:50-55, positional_emb.py:25,53: [cos(x f)16 sin(x f)16 cos(y f)16 sin(y f)16 cos(k yaw)32 sin(k yaw)32]).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tbx_hip.h"
#include "attn_core.h"
#include "tbx_common.h"

#ifdef TBX_ATTN_CLOCK
namespace tbx_attn {
__device__ unsigned long long g_attn_clk[8];
}
extern "C" int tbx_debug_attn_clock(unsigned long long* host_out) {  // copies and clears the phase sums (profiling build only)
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tbx_attn::g_attn_clk), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(tbx_attn::g_attn_clk), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

namespace {

using namespace tbx_attn;

struct AttnArgs {
  const float* qbuf;
  const float* rpe_k_bias;
  const float* fxy;   // pose_rpe.pe_xy.freqs  [32]  (only for segments given as relative poses)
  const float* fyaw;  // pose_rpe.pe_yaw.freqs [64]
  float* out;
  uint8_t* row_no_valid;
  tbx_attn_seg_t seg[2];
  int ldq, q_off, qt_off, ldo, n_rows, n_src, n_seg;
  float scale;   // 1 / sqrt(d_head)
  float scale2;  // log2(e) / sqrt(d_head): the forward's softmax runs in base 2
  // attention-probability dropout (training, attention_rpe.py:171-172): keep(row, t, head) is a counter-based hash of the
  // 64-bit seed read from device memory, so the backward regenerates the forward's mask and a captured graph draws a new
  // mask per replay (the host refills *drop_seed between replays)
  const uint64_t* drop_seed;
  uint32_t drop_call, drop_thresh;  // call id mixed into the seed; drop when hash < thresh = p * 2^32
  float drop_scale;                 // 1 / (1 - p)
  // time-batched calls (training: the T closed-loop steps of a scene evaluated as T consecutive batch entries): batch entry
  // b is step drop_time0 + b % T of scene b / T, and the mask is keyed by (scene row, step) - the masks of the batched call
  // are exactly those of T per-step calls with (T = 1, time0 = step). T = 1, time0 = 0: the plain (row) key.
  int drop_time_batch, drop_time0;
  // tbx_knarpe_attn_fwd_folded: the tbx_pack_weight_gemv image of linear_rpe's value half (4 groups x 32 outputs, k = 128):
  // the epilogue then forms (sum a v)_h + W_rpe_v,h (sum a e)_h + b_rpe_v,h itself and stores 128 floats per row instead of 640
  const float* fold_img;
  // wave-per-row form: the 4 rows of a workgroup are the SAME source token in 4 consecutive batch entries (row = (4*(quad / n_src)
  // + wave) * n_src + quad % n_src) instead of 4 consecutive tokens of one entry. Rollouts of a scene share the map and light
  // K/V tables and an agent's K-nearest sets barely differ between rollouts, so the workgroup's waves gather the same table rows
  // at about the same time (one L2 fetch, three L1 hits). Set when n_batch % 4 == 0 and a segment is shared (batch_div > 1).
  int batch_major;
  // consecutive workgroups' rows on ONE XCD (tbx::xcd_block): the rows of a rollout / scene read that rollout's / scene's tables
  int xcd;
};


// WPR = wavefronts cooperating on one source row: 1 for large grids (a wave per row, 4 rows per workgroup), 4 for small
// grids (the closed loop at a few scenes is latency-bound: 4 waves split a row's targets and combine through LDS).
//
// Single pass, online softmax: 8 lanes per target, 8 targets per wave per pass. Each 8-lane group keeps, for ITS target
// slot, a running (max, sum) per head and the un-normalised partial sums
//   O_slot[c] += p[h(c)] v[c] ,  E_slot[h][c] += p[h] e[c]      (the lane's channel slices)
// rescaled when the slot's running max grows. The 8 slots (and the WPR waves) are merged once per row:
//   out = sum_slots exp(m_slot - M) acc_slot / sum_slots exp(m_slot - M) l_slot.
// No LDS traffic and no barrier inside the target loop.
constexpr int FOLD_ROWS = 1 + (DR / 16) * 4;  // float4 rows of the value-fold image: a bias row + 32 weight rows of [128][4]

template <int WPR, bool DROP, bool KV16 = false, bool FOLD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPR == 1 ? 2 : 1))) void knarpe_attn_kernel(const AttnArgs a) {
  constexpr int OUTW = D + NH * DR;  // 640
  __shared__ float red_s[WPR > 1 ? WPR : 1][WPR > 1 ? (OUTW + 2 * NH) : 1];
  __shared__ __attribute__((aligned(16))) float fold_s[FOLD ? FOLD_ROWS * D * 4 : 4];  // 66 KiB: the fold image, by LDS-DMA
  __shared__ __attribute__((aligned(16))) float comb_s[FOLD ? (WPR == 1 ? 4 : 1) * OUTW : 1];  // WPR == 1: the workgroup's 4 rows
  if constexpr (FOLD) {
    // the image is requested first and lands while the targets are swept: 1 KiB per wave instruction, straight into LDS (asm:
    // the compiler would order every later LDS read behind a DMA it knows of - see csrc/rowchain.hip gemv_dma)
    const uint32_t lds0 = lds_addr(fold_s);
    const int w4 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int p = w4; p < FOLD_ROWS * 2; p += 4) glds_1k(a.fold_img + p * 256 + (threadIdx.x & 63) * 4, lds0 + (uint32_t)p * 1024u);
  }
  constexpr int RPB = 4 / WPR;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rib = wave / WPR;
  const int wir = wave % WPR;
  // The wave-per-row folded form walks row quads blockIdx.x, blockIdx.x + gridDim.x, ... (the launch has at most 2 workgroups per
  // CU): the 66 KiB fold image is fetched once per workgroup, not once per 4 rows. Every other form: one pass, quad = blockIdx.x.
  constexpr bool LOOP = FOLD && WPR == 1;
  const int n_quads = LOOP ? (a.n_rows + 3) / 4 : 0;
  const int quad0 = (!LOOP && a.xcd) ? tbx::xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  for (int quad = quad0; quad == quad0 || (LOOP && quad < n_quads); quad += gridDim.x) {
  int row = __builtin_amdgcn_readfirstlane(quad * RPB + rib);  // a wave works on one row: keep it in an SGPR
  if (WPR == 1 && a.batch_major) row = __builtin_amdgcn_readfirstlane((4 * (quad / a.n_src) + rib) * a.n_src + quad % a.n_src);
  if constexpr (LOOP) {
    row = row < a.n_rows ? row : a.n_rows - 1;  // the folded epilogue has workgroup barriers: a spare wave repeats the last row
  } else {
    if (row >= a.n_rows) return;  // uniform per row group (and per workgroup when WPR == 4)
  }
  const int b = row / a.n_src;
  const int s8 = lane & 7, tg = lane >> 3;

  // ---- query side in registers
  const float* qrow = a.qbuf + (int64_t)row * a.ldq;
  float4 qv[NH];
  ESlice qt[NH];
  float qb[NH];
  EFreq fq;
  fq.init(a.fxy, a.fyaw, s8);
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
    const float4 bk = *(const float4*)(a.rpe_k_bias + h * DH + s8 * 4);
    qb[h] = tbx::group8_sum(dot4(qv[h], bk));
    qt[h].load(qrow + a.qt_off + h * DR, s8);
  }

  // ---- this wave's share of the row's targets (attn_core.h), then the merge of its 8 target slots
  RowAcc st;
  st.zero();
  sweep<WPR, DROP, KV16>(a, row, b, wir, s8, tg, qv, qt, qb, fq, st);
  // (merge_slots_by_head: lane l holds head l >> 4's sums of channel slice s8; lanes l and l ^ 8 the same numbers)
  float M[NH], Mh, Lh;
  float4 om;
  ESlice em;
  merge_slots_by_head(st, M, Mh, Lh, om, em);
  const int hq = lane >> 4;
  float* orow = a.out + (int64_t)row * a.ldo;
  if constexpr (WPR == 1 && FOLD) {
    // a wave per row, 4 rows per workgroup: the normalised row goes to LDS, then thread (c = tid & 127, r0 = tid >> 7) runs the
    // value fold of output column c for rows r0 and r0 + 2 (one read of the weight image for both) - the same k-ordered v_fma
    // chain as the 4-waves-per-row form below and as the LINEAR stage it replaces: 128 floats per row leave the kernel
    const bool any_valid = M[0] > -INFINITY;
    if ((lane & 8) == 0) {
      const float inv_l = any_valid ? 1.0f / Lh : 0.f;
      scale4(om, inv_l);
      *(float4*)(&comb_s[rib * OUTW + hq * DH + s8 * 4]) = om;
      em.scale(inv_l);
      em.store(&comb_s[rib * OUTW + D + hq * DR], s8);
    }
    if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the fold image have landed
    __syncthreads();
    const int c = threadIdx.x & (D - 1), r0 = threadIdx.x >> 7, h = c / DH;
    const float4* wq = (const float4*)fold_s + c;  // row r of the image: wq[r * 128]
    const float* e0 = comb_s + r0 * OUTW + D + h * DR;
    const float* e1 = e0 + 2 * OUTW;
    float acc0 = fold_s[c * 4] + comb_s[r0 * OUTW + c], acc1 = fold_s[c * 4] + comb_s[(r0 + 2) * OUTW + c];
    const float4* e04 = (const float4*)e0;  // comb_s rows are 16-byte aligned (OUTW, D, DR multiples of 4)
    const float4* e14 = (const float4*)e1;
#pragma unroll 2
    for (int kb = 0; kb < DR / 16; ++kb) {
      // the 16 activations of the k-block as 4 broadcast ds_read_b128 per row (element [g*4 + t] multiplies W[c][kb*16 + g*4 + t])
      const float4 x0 = e04[kb * 4], x1 = e04[kb * 4 + 1], x2 = e04[kb * 4 + 2], x3 = e04[kb * 4 + 3];
      const float4 y0 = e14[kb * 4], y1 = e14[kb * 4 + 1], y2 = e14[kb * 4 + 2], y3 = e14[kb * 4 + 3];
      const float4 w0 = wq[(1 + kb * 4) * D], w1 = wq[(2 + kb * 4) * D], w2 = wq[(3 + kb * 4) * D], w3 = wq[(4 + kb * 4) * D];
#define TBX_FOLD_STEP(W, XC)                                                                              \
  acc0 = __builtin_fmaf(x0.XC, W.x, acc0); acc1 = __builtin_fmaf(y0.XC, W.x, acc1);                       \
  acc0 = __builtin_fmaf(x1.XC, W.y, acc0); acc1 = __builtin_fmaf(y1.XC, W.y, acc1);                       \
  acc0 = __builtin_fmaf(x2.XC, W.z, acc0); acc1 = __builtin_fmaf(y2.XC, W.z, acc1);                       \
  acc0 = __builtin_fmaf(x3.XC, W.w, acc0); acc1 = __builtin_fmaf(y3.XC, W.w, acc1)
      TBX_FOLD_STEP(w0, x);
      TBX_FOLD_STEP(w1, y);
      TBX_FOLD_STEP(w2, z);
      TBX_FOLD_STEP(w3, w);
#undef TBX_FOLD_STEP
    }
    int64_t ra = (int64_t)quad * 4 + r0, rb = ra + 2;
    if (a.batch_major) {
      ra = (int64_t)(4 * (quad / a.n_src) + r0) * a.n_src + quad % a.n_src;
      rb = ra + 2 * (int64_t)a.n_src;
    }
    if (ra < a.n_rows) a.out[ra * a.ldo + c] = acc0;
    if (rb < a.n_rows) a.out[rb * a.ldo + c] = acc1;
    __syncthreads();  // comb_s is rewritten by the next quad
  } else if constexpr (WPR == 1) {
    const bool any_valid = M[0] > -INFINITY;  // masks are per target, so every head sees the same validity
    if ((lane & 8) == 0) {
      const float inv_l = any_valid ? 1.0f / Lh : 0.f;
      scale4(om, inv_l);
      *(float4*)(orow + hq * DH + s8 * 4) = om;
      em.scale(inv_l);
      em.store(orow + D + hq * DR, s8);
    }
    if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
  } else {
    // per-wave (M, L, un-normalised sums) -> LDS, then all threads combine the WPR waves
    if ((lane & 8) == 0) {
      *(float4*)(&red_s[wir][hq * DH + s8 * 4]) = om;
      em.store(&red_s[wir][D + hq * DR], s8);
    }
    if ((lane & 15) == 0) {
      red_s[wir][OUTW + hq] = Mh;
      red_s[wir][OUTW + NH + hq] = Lh;
    }
    if constexpr (FOLD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the fold image have landed
    __syncthreads();
    float inv_l[NH], fw[WPR][NH];
    bool any_valid = false;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < WPR; ++w) mm = fmaxf(mm, red_s[w][OUTW + h]);
      float ll = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) {
        const float mw = red_s[w][OUTW + h];
        fw[w][h] = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - mm);
        ll = __builtin_fmaf(fw[w][h], red_s[w][OUTW + NH + h], ll);
      }
      any_valid = any_valid || mm > -INFINITY;
      inv_l[h] = (mm > -INFINITY) ? 1.0f / ll : 0.f;
    }
    if constexpr (FOLD) {
      // the combined row goes to LDS instead of global memory; thread c < 128 then runs the value fold of its output column:
      // out[c] = b[c] + (sum a v)[c] + sum_k W_rpe_v[c][k] (sum a e)_head(c)[k] as a v_fma chain in the k order of the LINEAR
      // stage it replaces (per k-block of 16: t = 0..3, g = 0..3, k = kb*16 + g*4 + t, starting from bias + (sum a v)[c]):
      // bit-identical to [640-wide store -> chain LOAD -> grouped LINEAR stage], without the 2.5 KB round trip per row
      for (int c = threadIdx.x; c < OUTW; c += 256) {
        const int h = c < D ? c / DH : (c - D) / DR;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) acc = __builtin_fmaf(fw[w][h], red_s[w][c], acc);
        comb_s[c] = acc * inv_l[h];
      }
      __syncthreads();
      if (threadIdx.x < D) {
        const int c = threadIdx.x, h = c / DH;
        const float acc = gemv_chain(fold_s, c, comb_s + D + h * DR, DR / 16, fold_s[c * 4] + comb_s[c]);
        orow[c] = acc;
      }
    } else {
      for (int c = threadIdx.x; c < OUTW; c += 256) {
        const int h = c < D ? c / DH : (c - D) / DR;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) acc = __builtin_fmaf(fw[w][h], red_s[w][c], acc);
        orow[c] = acc * inv_l[h];
      }
    }
    if (threadIdx.x == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
  }
  }  // quad loop
}

// (A software-pipelined wave-per-row form - the query side in a per-wave LDS copy, the next pass's K / V rows and pose requested one
// pass ahead - was built in round 3 and measured SLOWER at the 32 x 128 shape (fp32 tables 5.97 -> 5.16 M agent-steps/s with 15
// spilled registers, bf16 tables 5.93 -> 5.57 M): git history (knarpe_attn_pf_kernel) and profiles/MEASUREMENT_LOG.md, not the build.
// The matrix-core form of large launches is csrc/attn_mfma.hip.)

// =====================================================================================================================
// LDS-ring form of the wave-per-row forward (large launches, inference). Why: at 4096 rows x 104 pairs the sweep above runs
// 13 dependent passes per wave - index -> 8 K/V row gathers (L2 / HBM, ~1-1.5 us under load) -> ~650 VALU cycles - with
// one pass in flight per wave and 2 waves per SIMD (250 VGPRs): 4096 waves / 2048 slots x 13 passes x ~1.7 us = the measured 46 us
// with the VALU < 50 % busy, whatever the bytes per pair (fp32 and bf16 tables cost the same). Memory-level parallelism per wave is
// what is missing, and registers cannot hold a second pass of K/V rows. Here the rows of the NEXT R - 1 passes are in flight as
// LDS-DMA gathers (global_load_lds_dwordx4: per-lane global address, 1 KiB per instruction straight into the wave's private ring of
// R slots in LDS, no destination registers), issued from target indices that were loaded ONCE per row into registers (lane l
// holds targets l and l + 64 of each segment; a pass fetches its 8 indices with ds_bpermute) - no index -> row dependency left in
// the loop. A pass then reads its K / V fragments and relative poses back from its slot (each lane exactly the bytes its own DMA
// lane wrote: conflict-free) and runs the same arithmetic as `sweep` (attn_core.h), in the same order: bit-identical rows.
// fp32 tables: 8.25 KiB per slot, R = 4, one wave per SIMD (4 waves = 132 KiB per CU); bf16 tables: 4.25 KiB per slot, R = 4, two
// waves per SIMD. Completion = the issuing wave's own vmcnt (VMEM returns in order): no barriers anywhere.
__device__ __forceinline__ void glds_4(const void* gsrc_lane, uint32_t lds_byte_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_byte_addr)
               : "memory");
}

template <bool KV16>
struct RingGeom {
  static constexpr int KVB = KV16 ? 4096 : 8192;  // K then V fragments of a pass
  static constexpr int SLOT = KVB + 256;          // + 8 targets x 8 lanes x 4 B of relative poses (3 of 8 lanes used)
  static constexpr int NDMA = (KV16 ? 4 : 8) + 1;
};

template <bool KV16, int R, int WPE, bool DROP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void knarpe_attn_ring_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char ring_s[];
  using G = RingGeom<KV16>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int row = __builtin_amdgcn_readfirstlane((a.xcd ? tbx::xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 4 + wave);
  if (row >= a.n_rows) return;
  const int b = row / a.n_src;
  const int s8 = lane & 7, tg = lane >> 3;
  char* ring = ring_s + wave * R * G::SLOT;
  const uint32_t ring_lds = lds_addr(ring);

  // ---- the row's target indices / masks, once: lane l holds targets l and l + 64 of each segment
  int idx_lo[2] = {0, 0}, idx_hi[2] = {0, 0}, inv_lo[2] = {1, 1}, inv_hi[2] = {1, 1};
#pragma unroll
  for (int sg = 0; sg < 2; ++sg) {
    if (sg < a.n_seg) {
      const tbx_attn_seg_t& S = a.seg[sg];
      const int64_t pb = (int64_t)row * S.k;
      if (lane < S.k) idx_lo[sg] = S.idx[pb + lane], inv_lo[sg] = S.invalid[pb + lane];
      if (lane + 64 < S.k) idx_hi[sg] = S.idx[pb + lane + 64], inv_hi[sg] = S.invalid[pb + lane + 64];
    }
  }
  // ---- query side in registers
  const float* qrow = a.qbuf + (int64_t)row * a.ldq;
  float4 qv[NH];
  ESlice qt[NH];
  float qb[NH];
  EFreq fq;
  fq.init(a.fxy, a.fyaw, s8);
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
    const float4 bk = *(const float4*)(a.rpe_k_bias + h * DH + s8 * 4);
    qb[h] = tbx::group8_sum(dot4(qv[h], bk));
    qt[h].load(qrow + a.qt_off + h * DR, s8);
  }
  RowAcc st;
  st.zero();
  float(&m_run)[NH] = st.m_run;
  float(&l_run)[NH] = st.l_run;
  float4(&oacc)[NH] = st.oacc;
  ESlice(&eacc)[NH] = st.eacc;

  // the 8 lanes of target group tg fetch that target's index / mask from the lane that holds it
  auto pick = [&](int sg, int tt, const int (&lo)[2], const int (&hi)[2]) -> int {
    const int src = (tt & 63) * 4;
    const int vlo = __builtin_amdgcn_ds_bpermute(src, sg == 0 ? lo[0] : lo[1]);
    const int vhi = __builtin_amdgcn_ds_bpermute(src, sg == 0 ? hi[0] : hi[1]);
    return tt < 64 ? vlo : vhi;
  };
  auto issue = [&](int sg, int base, int slot) {
    const tbx_attn_seg_t& S = a.seg[sg];
    const int t = base + tg;
    const int tt = t < S.k ? t : S.k - 1;
    const int j = pick(sg, tt, idx_lo, idx_hi);
    constexpr int EB = KV16 ? 2 : 4;  // bytes per table element
    const char* trow = (const char*)S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt + j) * (int64_t)S.ld_kv * EB;
    const uint32_t dst = ring_lds + (uint32_t)(slot * G::SLOT);
    if constexpr (!KV16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) glds_1k((const float*)(trow + (S.k_off + q * 32 + s8 * 4) * 4), dst + q * 1024);
#pragma unroll
      for (int q = 0; q < 4; ++q) glds_1k((const float*)(trow + (S.v_off + q * 32 + s8 * 4) * 4), dst + 4096 + q * 1024);
    } else {  // 8 bf16 channels per lane: a DMA covers 64 channels (two heads) of the pass's 8 rows
#pragma unroll
      for (int q = 0; q < 2; ++q) glds_1k((const float*)(trow + (S.k_off + q * 64 + s8 * 8) * 2), dst + q * 1024);
#pragma unroll
      for (int q = 0; q < 2; ++q) glds_1k((const float*)(trow + (S.v_off + q * 64 + s8 * 8) * 2), dst + 2048 + q * 1024);
    }
    glds_4(S.rel_pose + ((int64_t)row * S.k + tt) * 3 + (s8 < 3 ? s8 : 0), dst + G::KVB);
  };
  DropKey dk;
  if constexpr (DROP) dk.init(a, row, b);
  auto consume = [&](int sg, int base, int slot, int t_off) {
#pragma clang fp contract(off)
    const tbx_attn_seg_t& S = a.seg[sg];
    const int t = base + tg;
    const bool active = t < S.k;
    const int tt = active ? t : S.k - 1;
    const bool ok = (pick(sg, tt, inv_lo, inv_hi) == 0) & active;
    const char* sl = ring + slot * G::SLOT;
    float4 kq[4], v[4];
    if constexpr (!KV16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        kq[q] = *(const float4*)(sl + q * 1024 + lane * 16);
        v[q] = *(const float4*)(sl + 4096 + q * 1024 + lane * 16);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // channels q * 32 + s8 * 4 .. + 4: the half (s8 & 1) of DMA lane (q & 1) * 4 + s8 / 2 of DMA q / 2
        const int o = (q >> 1) * 1024 + (tg * 8 + (q & 1) * 4 + (s8 >> 1)) * 16 + (s8 & 1) * 8;
        const uint2 rk = *(const uint2*)(sl + o), rv = *(const uint2*)(sl + 2048 + o);
        kq[q] = make_float4(__uint_as_float(rk.x << 16), __uint_as_float(rk.x & 0xffff0000u), __uint_as_float(rk.y << 16),
                            __uint_as_float(rk.y & 0xffff0000u));
        v[q] = make_float4(__uint_as_float(rv.x << 16), __uint_as_float(rv.x & 0xffff0000u), __uint_as_float(rv.y << 16),
                           __uint_as_float(rv.y & 0xffff0000u));
      }
    }
    ESlice e;
    const float4 rel = *(const float4*)(sl + G::KVB + tg * 32);
    const float rel3[3] = {rel.x, rel.y, rel.z};
    fq.embed(rel3, e);
    // ---- the arithmetic of attn_core.h's sweep (no dropout), statement for statement
    float sc[NH];
    bool jump = false;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      sc[h] = (tbx::group8_sum(pair_score(kq[h], qv[h], e, qt[h])) + qb[h]) * a.scale2;
      jump = jump || (ok && m_run[h] > -INFINITY && sc[h] - m_run[h] > 64.f);
    }
    if (__builtin_expect(__ballot(jump) != 0ull, 0)) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if (ok && m_run[h] > -INFINITY && sc[h] - m_run[h] > 64.f) {
          const float alpha = __builtin_amdgcn_exp2f(m_run[h] - sc[h]);
          l_run[h] *= alpha;
          scale4(oacc[h], alpha);
          eacc[h].scale(alpha);
          m_run[h] = sc[h];
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      m_run[h] = (ok && m_run[h] == -INFINITY) ? sc[h] : m_run[h];
      const float pr = ok ? __builtin_amdgcn_exp2f(sc[h] - m_run[h]) : 0.f;
      l_run[h] += pr;  // the normaliser is that of the un-dropped softmax
      float pd = pr;
      if constexpr (DROP) pd = dk.keep((uint32_t)(t_off + t), (uint32_t)h, a.drop_thresh) ? pr * a.drop_scale : 0.f;
      fma4(oacc[h], pd, v[h]);
      eacc[h].fma(pd, e);
    }
  };

  // ---- the pass stream: (segment, base) cursors; the issue cursor runs R - 1 passes ahead of the consume cursor
  int isg = 0, ibase = 0, islot = 0, ahead = 0;
  auto issue_next = [&]() {
    while (isg < a.n_seg && ibase >= a.seg[isg].k) ++isg, ibase = 0;
    if (isg >= a.n_seg) return;
    issue(isg, ibase, islot);
    ibase += 8;
    islot = islot + 1 == R ? 0 : islot + 1;
    ++ahead;
  };
#pragma unroll 1
  for (int i = 0; i < R - 1; ++i) issue_next();
  int cslot = 0, t_off = 0;
  for (int sg = 0; sg < a.n_seg; t_off += a.seg[sg].k, ++sg) {
    const int ks = a.seg[sg].k;
#pragma unroll 1
    for (int base = 0; base < ks; base += 8) {
#ifdef TBX_ATTN_CLOCK
      unsigned long long r0, r1, r2, r3;
      TBX_ACLK(r0, l_run[0]);
#endif
      issue_next();  // into the slot consumed in the previous pass (its fragments are in registers / used: lgkmcnt is waited below)
#ifdef TBX_ATTN_CLOCK
      TBX_ACLK(r1, l_run[0]);
#endif
      // passes issued and not yet consumed: `ahead` (this one included). The oldest outstanding DMA group is this pass's.
      if (ahead == R)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 1) * G::NDMA) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TBX_ATTN_CLOCK
      TBX_ACLK(r2, l_run[0]);
#endif
      consume(sg, base, cslot, t_off);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot's bytes are in registers before a later DMA may overwrite it
#ifdef TBX_ATTN_CLOCK
      TBX_ACLK(r3, (oacc[0].x + l_run[1]) + (eacc[3].ws.w + eacc[0].xc.x));
      if (blockIdx.x == 0 && threadIdx.x == 0)
        g_attn_clk[5] += r1 - r0, g_attn_clk[6] += r2 - r1, g_attn_clk[7] += r3 - r2, g_attn_clk[4] += 1;
#endif
      cslot = cslot + 1 == R ? 0 : cslot + 1;
      --ahead;
    }
  }

  float M[NH], Mh, Lh;
  float4 om;
  ESlice em;
  merge_slots_by_head(st, M, Mh, Lh, om, em);  // (lane l: head l >> 4's sums of channel slice s8; lanes l and l ^ 8 the same numbers)
  const int hq = lane >> 4;
  float* orow = a.out + (int64_t)row * a.ldo;
  const bool any_valid = M[0] > -INFINITY;
  if ((lane & 8) == 0) {
    const float inv_l = any_valid ? 1.0f / Lh : 0.f;
    scale4(om, inv_l);
    *(float4*)(orow + hq * DH + s8 * 4) = om;
    em.scale(inv_l);
    em.store(orow + D + hq * DR, s8);
  }
  if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
}

// =====================================================================================================================
// Backward (training). Same factorised math as the forward:
//   s[h,t] = q_h.(k_h[idx_t] + bk_h) + qt_h.e_t ;  a = softmax(s * scale) (masked) ;  out = [sum_t a v_h | sum_t a e_t]
// Given dout = [dO (128) | dE (4 x 128)] per row it produces dq, dqt (written to dqbuf at q_off / qt_off), dK / dV
// scattered with atomicAdd into the K/V-table-shaped gradient of each segment, and d(rpe_k_bias) per row ([rows, 128]).
// The pose embeddings carry no gradient (relative poses are computed under no_grad in the reference, utils/rpe.py:7).
// One wavefront per source row; probabilities are recomputed (nothing but the inputs is saved by the forward).
struct AttnBwdArgs {
  AttnArgs f;         // forward arguments (qbuf, segs, ...); f.out is unused
  const float* dout;  // [rows, ldo] = dO | dE
  float* dqbuf;       // [rows, ldq]: dq at q_off, dqt at qt_off (overwritten)
  float* dkv[2];      // per segment, same [.., ld_kv] layout as seg.kv (accumulated)
  float* dbias_k;     // [rows, 128]: per-row gradient of rpe_k_bias (overwritten; the parameter's gradient is the column sum)
  // gather mode (coef != nullptr): instead of scattering dK / dV with atomics the row kernel stores, per pair, the four
  // dS[h] and the four (dropped) probabilities; knarpe_attn_dkv_kernel then sums each target token's contributions through
  // the inverse K-nearest lists - no atomics, ~4x less time at 1024 rows x 89 pairs
  float* coef;        // [rows, ktot, 8]
};

// WANT_DB: the per-row d(rpe_k_bias) is computed and stored (b.dbias_k != NULL). It is identically zero in exact arithmetic - adding
// q_h . bk_h to every score of a row leaves its softmax unchanged, so sum_t dS[h,t] = 0 - and what the accumulation yields is round-off
// of the size of the last bits of dq; the training step passes NULL (16 fmas per pair, 128 floats per row and a column sum less).
template <bool WANT_DB>
__global__ __launch_bounds__(256) void knarpe_attn_bwd_kernel(const AttnBwdArgs b) {
  const AttnArgs& a = b.f;
  __shared__ float p_s[4][NH][KMAX];  // raw scores, then probabilities a[h,t]
  __shared__ float d_s[4][NH][KMAX];  // da[h,t] (dropout factor applied), then dS[h,t]
  __shared__ float k_s[4][NH][KMAX];  // dropout factor m / (1 - p) of (h, t); 1 without dropout
  __shared__ uint8_t ok_s[4][KMAX];   // 1 = valid target
  const int lane = threadIdx.x & 63;
  const int rib = threadIdx.x >> 6;
  const int row = __builtin_amdgcn_readfirstlane((a.xcd ? tbx::xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 4 + rib);  // a wave works on one row: scalar addressing below
  if (row >= a.n_rows) return;
  const int bidx = row / a.n_src;
  const bool drop = a.drop_thresh != 0u;
  DropKey dkey;
  dkey.lo = dkey.hi = dkey.krow = 0u;
  if (drop) dkey.init(a, row, bidx);
  const int k0 = a.seg[0].k;
  const int ktot = k0 + (a.n_seg > 1 ? a.seg[1].k : 0);
  const int s8 = lane & 7, tg = lane >> 3;
  const float* qrow = a.qbuf + (int64_t)row * a.ldq;
  const float* drow = b.dout + (int64_t)row * a.ldo;
  float4 qv[NH], bkv[NH], dov[NH];
  EFreq fq;
  fq.init(a.fxy, a.fyaw, s8);
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
    bkv[h] = *(const float4*)(a.rpe_k_bias + h * DH + s8 * 4);
    dov[h] = *(const float4*)(drow + h * DH + s8 * 4);
  }
  // ---- pass 1 (one sweep over the targets, segment by segment, straight-line like the forward): raw scores
  //      s[h,t] = q_h.k_h + qt_h.e + q_h.bk_h and da[h,t] = dO_h.v_h[idx_t] + dE_h.e_t (times the dropout factor) -> LDS.
  //      Slots past a segment's K re-read its last pair and are simply not stored.
  bool any_valid = false;
  {
    ESlice qt[NH], dev[NH];
    float qb[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      qb[h] = tbx::group8_sum(dot4(qv[h], bkv[h]));
      qt[h].load(qrow + a.qt_off + h * DR, s8);
      dev[h].load(drow + D + h * DR, s8);
    }
    int t_off = 0;
    for (int sg = 0; sg < a.n_seg; t_off += a.seg[sg].k, ++sg) {
      const tbx_attn_seg_t& S = a.seg[sg];
      const float* kvb = S.kv + (int64_t)(bidx / S.batch_div) * S.n_tgt * S.ld_kv;
      const int64_t pbase = (int64_t)row * S.k;
      for (int base = 0; base < S.k; base += 8) {
        const int tl = base + tg;
        const bool active = tl < S.k;
        const int64_t pi = pbase + (active ? tl : S.k - 1);
        const int j = S.idx[pi];
        const bool ok = (S.invalid[pi] == 0) & active;
        const float* trow = kvb + (int64_t)j * S.ld_kv;
        ESlice e;
        load_e(S, pi, s8, fq, e);
        float sc[NH], da[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float4 kq = *(const float4*)(trow + S.k_off + h * 32 + s8 * 4);
          const float4 vv = *(const float4*)(trow + S.v_off + h * 32 + s8 * 4);
          sc[h] = tbx::group8_sum(pair_score(kq, qv[h], e, qt[h])) + qb[h];
          da[h] = tbx::group8_sum(pair_score(vv, dov[h], e, dev[h]));
        }
        if (active && s8 == 0) {
          const int t = t_off + tl;
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            // out = sum_t a_t m_t / (1 - p) (v_t | e_t): the mask factor multiplies d(a_t); kept in k_s for the dV weights
            float kf = 1.f;
            if (drop) kf = dkey.keep((uint32_t)t, (uint32_t)h, a.drop_thresh) ? a.drop_scale : 0.f;
            p_s[rib][h][t] = sc[h];
            k_s[rib][h][t] = kf;
            d_s[rib][h][t] = da[h] * kf;
          }
          ok_s[rib][t] = ok ? 1 : 0;
        }
        any_valid = any_valid || (__ballot(ok) != 0ull);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- softmax in base 2 (probabilities back into p_s) and its backward dS = a (da - sum_t a da) / sqrt(d_h); rows
  //      without a valid target get zero probabilities (their forward output is zero and the caller masks them, so every
  //      gradient of such a row is zero)
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float sv[2], dv[2];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      float sc = -INFINITY;
      dv[q] = 0.f;
      if (t < ktot) {
        if (any_valid && ok_s[rib][t] != 0) sc = p_s[rib][h][t] * a.scale2;
        dv[q] = d_s[rib][h][t];
      }
      sv[q] = sc;
      m = fmaxf(m, sc);
    }
    m = tbx::wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sv[q] = (sv[q] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(sv[q] - m);
      sum += sv[q];
    }
    sum = tbx::wave_sum(sum);
    const float inv_sum = any_valid ? 1.0f / sum : 0.f;
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sv[q] *= inv_sum;
      part += sv[q] * dv[q];
    }
    const float dotv = tbx::wave_sum(part);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      if (t < ktot) {
        p_s[rib][h][t] = sv[q];
        d_s[rib][h][t] = sv[q] * (dv[q] - dotv) * a.scale;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- pass 2: dq, dqt, d bias_k (registers, reduced over the 8 target slots at the end); dK, dV scattered
  float4 dq[NH], dbk[NH];
  ESlice dqt[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    dq[h] = dbk[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    dqt[h].zero();
  }
  {
    int t_off = 0;
    for (int sg = 0; sg < a.n_seg; t_off += a.seg[sg].k, ++sg) {
      const tbx_attn_seg_t& S = a.seg[sg];
      const int64_t tb = (int64_t)(bidx / S.batch_div) * S.n_tgt * S.ld_kv;
      const int64_t pbase = (int64_t)row * S.k;
      float* dkvb = b.dkv[sg];
      for (int base = 0; base < S.k; base += 8) {
        const int tl = base + tg;
        if (tl >= S.k) continue;
        const int t = t_off + tl;
        float ds[NH], pa[NH];
        bool any = false;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          ds[h] = d_s[rib][h][t];
          pa[h] = p_s[rib][h][t] * k_s[rib][h][t];  // weight of v_t in the output (dropped probability)
          any = any || ds[h] != 0.f || pa[h] != 0.f;
        }
        if (b.coef != nullptr) {  // gather mode: the 8 lanes of the target's group store its 8 coefficients (32 contiguous bytes)
          float cv = ds[0];
#pragma unroll
          for (int h = 1; h < NH; ++h) cv = (s8 == h) ? ds[h] : cv;
#pragma unroll
          for (int h = 0; h < NH; ++h) cv = (s8 == NH + h) ? pa[h] : cv;
          b.coef[((int64_t)row * ktot + t) * 8 + s8] = cv;
        }
        if (!any) continue;  // masked target (or an all-masked row): nothing flows
        const int64_t pi = pbase + tl;
        const int64_t trow = tb + (int64_t)S.idx[pi] * S.ld_kv;
        const float* krow = S.kv + trow + S.k_off;
        float* dk = dkvb + trow + S.k_off;
        float* dv = dkvb + trow + S.v_off;
        ESlice e;
        load_e(S, pi, s8, fq, e);
#pragma unroll
        for (int h = 0; h < NH; ++h) {  // h doubles as the K/V channel block
          const float4 kq = *(const float4*)(krow + h * 32 + s8 * 4);
          const float g = ds[h];
          dq[h].x += g * (kq.x + bkv[h].x); dq[h].y += g * (kq.y + bkv[h].y);
          dq[h].z += g * (kq.z + bkv[h].z); dq[h].w += g * (kq.w + bkv[h].w);
          if constexpr (WANT_DB) fma4(dbk[h], g, qv[h]);
          dqt[h].fma(g, e);
          if (b.coef != nullptr) continue;
          const int c0 = h * 32 + s8 * 4;
          atomicAdd(dk + c0 + 0, g * qv[h].x); atomicAdd(dk + c0 + 1, g * qv[h].y);
          atomicAdd(dk + c0 + 2, g * qv[h].z); atomicAdd(dk + c0 + 3, g * qv[h].w);
          const float pv = pa[h];
          atomicAdd(dv + c0 + 0, pv * dov[h].x); atomicAdd(dv + c0 + 1, pv * dov[h].y);
          atomicAdd(dv + c0 + 2, pv * dov[h].z); atomicAdd(dv + c0 + 3, pv * dov[h].w);
        }
      }
    }
  }
  // the 8 target slots' sums by head (tbx::slot_sum4: four values per cross-row exchange, the same butterfly as slot_sum - see
  // merge_slots_by_head): lane l ends with head l >> 4's dq / d bias_k / dqt of channel slice s8, lanes l and l ^ 8 the same numbers
  float4 r, rb = make_float4(0.f, 0.f, 0.f, 0.f);
  ESlice rt;
#define TBX_S4(A, F) tbx::slot_sum4(A[0].F, A[1].F, A[2].F, A[3].F)
  r.x = TBX_S4(dq, x), r.y = TBX_S4(dq, y), r.z = TBX_S4(dq, z), r.w = TBX_S4(dq, w);
  if constexpr (WANT_DB) rb.x = TBX_S4(dbk, x), rb.y = TBX_S4(dbk, y), rb.z = TBX_S4(dbk, z), rb.w = TBX_S4(dbk, w);
  rt.xc.x = TBX_S4(dqt, xc.x), rt.xc.y = TBX_S4(dqt, xc.y), rt.xs.x = TBX_S4(dqt, xs.x), rt.xs.y = TBX_S4(dqt, xs.y);
  rt.yc.x = TBX_S4(dqt, yc.x), rt.yc.y = TBX_S4(dqt, yc.y), rt.ys.x = TBX_S4(dqt, ys.x), rt.ys.y = TBX_S4(dqt, ys.y);
  rt.wc.x = TBX_S4(dqt, wc.x), rt.wc.y = TBX_S4(dqt, wc.y), rt.wc.z = TBX_S4(dqt, wc.z), rt.wc.w = TBX_S4(dqt, wc.w);
  rt.ws.x = TBX_S4(dqt, ws.x), rt.ws.y = TBX_S4(dqt, ws.y), rt.ws.z = TBX_S4(dqt, ws.z), rt.ws.w = TBX_S4(dqt, ws.w);
#undef TBX_S4
  float* dqrow = b.dqbuf + (int64_t)row * a.ldq;
  if ((lane & 8) == 0) {
    const int hq = lane >> 4;
    *(float4*)(dqrow + a.q_off + hq * DH + s8 * 4) = r;
    // per-row part of d(rpe_k_bias): every row adding into the same 128 floats serialises ~10^3-deep at the L2 atomic
    // units (measured: ~200 us of a 335 us launch at 1024 rows); the caller sums the rows
    if constexpr (WANT_DB) *(float4*)(b.dbias_k + (int64_t)row * D + hq * DH + s8 * 4) = rb;
    rt.store(dqrow + a.qt_off + hq * DR, s8);
  }
}

// dK / dV of one target token = sum over the pairs that selected it (inverse K-nearest list) of dS[h] q_h / p[h] dO_h.
// A wavefront per token: lanes 0-31 own the 128 dK channels (float4 each), lanes 32-63 the 128 dV channels; a pair costs
// one 512-B row of q or dO per half-wave and one coefficient per lane. The K and V columns of every token are overwritten.
struct DkvArgs {
  const float* qbuf;   // q at q_off
  const float* dout;   // dO at column 0
  const float* coef;   // [rows, ktot, 8]
  const int32_t* inv_ptr[2];
  const int32_t* inv_list[2];
  float* dkv[2];
  int32_t k[2], t_off[2], n_tgt[2], ld_kv[2], k_off[2], v_off[2], list_cap[2], tok0[2];  // tok0: first wave of the segment
  int ldq, q_off, ldo, ktot, n_tok, xcd;
};

__global__ __launch_bounds__(256) void knarpe_attn_dkv_kernel(const DkvArgs a) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((a.xcd ? tbx::xcd_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 4 + (threadIdx.x >> 6));
  if (w >= a.n_tok) return;
  const int sg = (w >= a.tok0[1]) ? 1 : 0;
  const int tok = w - a.tok0[sg];            // table * n_tgt + j
  const int table = tok / a.n_tgt[sg], j = tok - table * a.n_tgt[sg];
  const int32_t* ptr = a.inv_ptr[sg] + (int64_t)table * (a.n_tgt[sg] + 1);
  const int p0 = ptr[j], p1 = ptr[j + 1];
  const int32_t* list = a.inv_list[sg] + (int64_t)table * a.list_cap[sg];
  const int half = lane >> 5, c4 = lane & 31, h = c4 >> 3;
  const int k = a.k[sg];
  const float* src = half ? a.dout : a.qbuf + a.q_off;
  const int lds = half ? a.ldo : a.ldq;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = p0;
  for (; p + 1 < p1; p += 2) {  // two pairs in flight
    const int g0 = list[p], g1 = list[p + 1];
    const int r0 = g0 / k, r1 = g1 / k;
    const float c0 = a.coef[((int64_t)r0 * a.ktot + a.t_off[sg] + (g0 - r0 * k)) * 8 + half * 4 + h];
    const float c1 = a.coef[((int64_t)r1 * a.ktot + a.t_off[sg] + (g1 - r1 * k)) * 8 + half * 4 + h];
    const float4 v0 = *(const float4*)(src + (int64_t)r0 * lds + c4 * 4);
    const float4 v1 = *(const float4*)(src + (int64_t)r1 * lds + c4 * 4);
    fma4(acc, c0, v0);
    fma4(acc, c1, v1);
  }
  if (p < p1) {
    const int g0 = list[p];
    const int r0 = g0 / k;
    const float c0 = a.coef[((int64_t)r0 * a.ktot + a.t_off[sg] + (g0 - r0 * k)) * 8 + half * 4 + h];
    fma4(acc, c0, *(const float4*)(src + (int64_t)r0 * lds + c4 * 4));
  }
  float* out = a.dkv[sg] + ((int64_t)table * a.n_tgt[sg] + j) * a.ld_kv[sg] + (half ? a.v_off[sg] : a.k_off[sg]) + c4 * 4;
  *(float4*)out = acc;  // every token's K and V gradient columns are written (zero when nobody selected it): no pre-zeroing
}

// XCD-contiguous rows (tbx::xcd_block). TBX_ATTN_XCD: bit 0 the forward kernels (knarpe_attn_kernel, the ring form), bit 1 the backward
// (knarpe_attn_bwd_kernel, knarpe_attn_dkv_kernel), bit 3 training's forward launches too. Default 1 = the forward kernels of
// inference launches (no dropout). Measured (profiles/r06_attn_xcd_ab.txt): 32 x 128 agents 7.14 -> 7.28 M agent-steps/s, 128 x 128
// 9.00 -> 9.16 M, 16 / 64 scenes +1.1 / +2.7 % - a rollout's / scene's K/V tables then live in one XCD's L2; the matrix-core forward
// -0.8 % (its persistent grid already walks the rows in order); the backward kernels 139 -> 149 ms per training step (they stream
// [rows, pairs, 8] coefficient arrays: eight XCDs on eight distant regions instead of one); training's stepping pass +0.3 ms.
static int attn_xcd_mask() {
  static const int v = [] { const char* e = getenv("TBX_ATTN_XCD"); return (e && *e) ? atoi(e) : 1; }();
  return v;
}
static int attn_xcd() { return attn_xcd_mask() & 1; }

int fill_args(AttnArgs& a, const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch, int n_src,
              const tbx_attn_seg_t* segs, int n_seg, int ldo, const float* fxy, const float* fyaw) {
  if (!qbuf || !rpe_k_bias || !segs || n_batch <= 0 || n_src <= 0) return TBX_ERR_ARG;
  if (n_seg < 1 || n_seg > 2 || ldo < D + NH * DR) return TBX_ERR_UNSUPPORTED;
  if ((ldq % 4) || (q_off % 4) || (qt_off % 4) || (ldo % 4) || (((uintptr_t)qbuf) & 15) || (((uintptr_t)rpe_k_bias) & 15))
    return TBX_ERR_ALIGN;
  int ktot = 0;
  for (int i = 0; i < n_seg; ++i) {
    const tbx_attn_seg_t& s = segs[i];
    if (!s.kv || !s.idx || !s.invalid || (!s.emb && !s.rel_pose) || s.k <= 0 || s.n_tgt <= 0 || s.batch_div <= 0) return TBX_ERR_ARG;
    if (!s.emb && (!fxy || !fyaw)) return TBX_ERR_ARG;
    if ((s.ld_kv % 4) || (s.k_off % 4) || (s.v_off % 4) || (((uintptr_t)s.kv) & 15) || (s.emb && (((uintptr_t)s.emb) & 15)))
      return TBX_ERR_ALIGN;
    if ((s.kv_bf16 != 0) != (segs[0].kv_bf16 != 0)) return TBX_ERR_UNSUPPORTED;  // one element type per call
    ktot += s.k;
    a.seg[i] = s;
  }
  if (n_seg == 1) a.seg[1] = a.seg[0];
  if (ktot > KMAX) return TBX_ERR_UNSUPPORTED;
  a.qbuf = qbuf;
  a.rpe_k_bias = rpe_k_bias;
  a.fxy = fxy;
  a.fyaw = fyaw;
  a.out = nullptr;
  a.row_no_valid = nullptr;
  a.ldq = ldq;
  a.q_off = q_off;
  a.qt_off = qt_off;
  a.ldo = ldo;
  a.n_rows = n_batch * n_src;
  a.n_src = n_src;
  a.n_seg = n_seg;
  a.scale = 1.0f / sqrtf((float)DH);
  a.scale2 = 1.4426950408889634f / sqrtf((float)DH);
  a.fold_img = nullptr;
  static const bool bm_env = [] { const char* e = getenv("TBX_ATTN_BATCH_MAJOR"); return !(e && e[0] == '0'); }();
  bool shared = false;
  for (int i = 0; i < n_seg; ++i) shared = shared || segs[i].batch_div > 1;
  a.batch_major = (bm_env && shared && n_batch % 4 == 0) ? 1 : 0;
  a.xcd = attn_xcd();  // (set_dropout clears it for training's launches)
  return TBX_OK;
}

}  // namespace

namespace {
int set_dropout(AttnArgs& a, float p_drop, const uint64_t* drop_seed, uint32_t drop_call, int time_batch, int time0) {
  a.drop_seed = drop_seed;
  a.drop_call = drop_call;
  a.drop_thresh = 0u;
  a.drop_scale = 1.f;
  a.drop_time_batch = time_batch;
  a.drop_time0 = time0;
  if (p_drop < 0.f || p_drop >= 1.f || time_batch < 1 || time0 < 0) return TBX_ERR_ARG;
  if (p_drop > 0.f) {
    if (!drop_seed) return TBX_ERR_ARG;
    const double th = (double)p_drop * 4294967296.0;
    a.drop_thresh = th < 1.0 ? 1u : (uint32_t)th;
    a.drop_scale = 1.0f / (1.0f - p_drop);
    if (!(attn_xcd_mask() & 8)) a.xcd = 0;  // (training's launches: the plain order - see attn_xcd_mask; bit 3 forces it on)
  }
  return TBX_OK;
}
}  // namespace

static int fold_grid_max() {  // 2 workgroups (78 KiB of LDS each) per CU
  static const int g = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const char* e = getenv("TBX_FOLD_WG_PER_CU");
    const int per = e && atoi(e) > 0 ? atoi(e) : 2;
    return per * (cus > 0 ? cus : 256);
  }();
  return g;
}

// from how many source rows a wave takes a whole row (below: 4 waves split a row's targets). Calls with attention dropout (training's
// stepping pass): 1024, measured there at 1024 rows x 89 pairs (28 us with 4 waves per row, ~24 us with one). Inference: 193 - on
// the scenes-per-GPU curve (profiles/r04_scene_curve.json) 256 .. 512 agents' rows run 0.398 -> 0.306 ms per step (4 scenes) and
// 0.470 -> 0.333 (8 scenes) with the wave-per-row forms (the LDS ring when a SIMD has a lone wave) behind the tile kernels.
static int attn_big_rows(bool dropout = true) {
  static const int big_rows = [] {
    const char* e = getenv("TBX_ATTN_BIG_ROWS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 1024;
  }();
  static const int big_rows_infer = [] {
    const char* e = getenv("TBX_ATTN_BIG_ROWS_INFER");
    const int v = e ? atoi(e) : (getenv("TBX_ATTN_BIG_ROWS") ? atoi(getenv("TBX_ATTN_BIG_ROWS")) : 0);
    return v > 0 ? v : 193;
  }();
  return dropout ? big_rows : big_rows_infer;
}

extern "C" int tbx_knarpe_attn_fwd_dropout_tb(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias,
                                              int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo,
                                              uint8_t* row_no_valid, const float* freqs_xy, const float* freqs_yaw, float p_drop,
                                              const uint64_t* drop_seed, uint32_t drop_call, int time_batch, int time0,
                                              void* stream) {
  if (!out || !row_no_valid) return TBX_ERR_ARG;
  if (((uintptr_t)out) & 15) return TBX_ERR_ALIGN;
  AttnArgs a;
  int rc = fill_args(a, qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, ldo, freqs_xy, freqs_yaw);
  if (rc != TBX_OK) return rc;
  rc = set_dropout(a, p_drop, drop_seed, drop_call, time_batch, time0);
  if (rc != TBX_OK) return rc;
  a.out = out;
  a.row_no_valid = row_no_valid;
  const bool big = a.n_rows >= attn_big_rows(a.drop_thresh != 0u);  // a wave per row from here on (below: 4 waves split a row's targets)
  const dim3 grid(big ? (a.n_rows + 3) / 4 : a.n_rows), block(256);
  hipStream_t hs = (hipStream_t)stream;
  // the LDS-ring form (opt-in, TBX_ATTN_RING=1 / 2): large launches without dropout whose segments are all given as relative poses
  static const int ring_mode = [] { const char* e = getenv("TBX_ATTN_RING"); return e ? atoi(e) : 0; }();  // measured: no gain, opt-in
  // ... and, by default, launches that leave every SIMD with at most ONE wave (<= 1024 rows a wave per row: training's stepping pass,
  // 16 scenes x 64 agents): a lone wave has no partner to hide its gathers behind, the ring does (24 -> ~9 us per launch)
  static const int ring_lone_rows = [] { const char* e = getenv("TBX_ATTN_RING_LONE_ROWS"); return e ? atoi(e) : 1536; }();
  const bool lone = big && a.n_rows <= ring_lone_rows && segs[0].kv_bf16 == 0;
  bool ring_ok = big && (ring_mode != 0 || lone) && (a.drop_thresh == 0u || segs[0].kv_bf16 == 0);
  for (int i = 0; i < n_seg; ++i) ring_ok = ring_ok && segs[i].rel_pose != nullptr && segs[i].emb == nullptr && segs[i].k <= 128 && segs[i].k > 0;
  if (ring_ok) {
#define TBX_RING_LAUNCH(KV16F, RF, WPEF, DROPF)                                                                                   \
  do {                                                                                                                            \
    constexpr int bytes = 4 * (RF) * RingGeom<KV16F>::SLOT;                                                                       \
    static bool set = false;                                                                                                      \
    if (!set) {                                                                                                                   \
      if (hipFuncSetAttribute((const void*)knarpe_attn_ring_kernel<KV16F, RF, WPEF, DROPF>,                                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)                                   \
        return TBX_ERR_LAUNCH;                                                                                                    \
      set = true;                                                                                                                 \
    }                                                                                                                             \
    hipLaunchKernelGGL((knarpe_attn_ring_kernel<KV16F, RF, WPEF, DROPF>), grid, block, bytes, hs, a);                             \
  } while (0)
    if (segs[0].kv_bf16 != 0)
      TBX_RING_LAUNCH(true, 4, 2, false);
    else if (a.drop_thresh != 0u)
      TBX_RING_LAUNCH(false, 4, 1, true);
    else if (ring_mode == 2)
      TBX_RING_LAUNCH(false, 2, 2, false);
    else
      TBX_RING_LAUNCH(false, 4, 1, false);
#undef TBX_RING_LAUNCH
    return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
  }
  if (segs[0].kv_bf16 != 0) {  // bf16 K/V tables: inference only
    if (a.drop_thresh != 0u) return TBX_ERR_UNSUPPORTED;
    if (big)
      hipLaunchKernelGGL((knarpe_attn_kernel<1, false, true>), grid, block, 0, hs, a);
    else
      hipLaunchKernelGGL((knarpe_attn_kernel<4, false, true>), grid, block, 0, hs, a);
  } else if (a.drop_thresh != 0u) {
    if (big)
      hipLaunchKernelGGL((knarpe_attn_kernel<1, true>), grid, block, 0, hs, a);
    else
      hipLaunchKernelGGL((knarpe_attn_kernel<4, true>), grid, block, 0, hs, a);
  } else {
    if (big)
      hipLaunchKernelGGL((knarpe_attn_kernel<1, false>), grid, block, 0, hs, a);
    else
      hipLaunchKernelGGL((knarpe_attn_kernel<4, false>), grid, block, 0, hs, a);
  }
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_knarpe_attn_fwd_folded(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                          int n_src, const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                                          const float* freqs_xy, const float* freqs_yaw, const float* fold_image, void* stream) {
  if (!out || !row_no_valid || !fold_image) return TBX_ERR_ARG;
  if ((((uintptr_t)out) & 15) || (((uintptr_t)fold_image) & 15) || (ldo & 3) || ldo < D) return TBX_ERR_ALIGN;
  AttnArgs a;
  int rc = fill_args(a, qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, D + NH * DR, freqs_xy, freqs_yaw);
  if (rc != TBX_OK) return rc;
  rc = set_dropout(a, 0.f, nullptr, 0u, 1, 0);
  if (rc != TBX_OK) return rc;
  a.ldo = ldo;
  a.out = out;
  a.row_no_valid = row_no_valid;
  a.fold_img = fold_image;
  const bool big = a.n_rows >= attn_big_rows();  // a wave per row from here on, 4 rows per workgroup
  const int quads = (a.n_rows + 3) / 4;
  const dim3 grid(big ? (quads < fold_grid_max() ? quads : fold_grid_max()) : a.n_rows), block(256);
  hipStream_t hs = (hipStream_t)stream;
  if (segs[0].kv_bf16 != 0) {
    if (big)
      hipLaunchKernelGGL((knarpe_attn_kernel<1, false, true, true>), grid, block, 0, hs, a);
    else
      hipLaunchKernelGGL((knarpe_attn_kernel<4, false, true, true>), grid, block, 0, hs, a);
  } else {
    if (big)
      hipLaunchKernelGGL((knarpe_attn_kernel<1, false, false, true>), grid, block, 0, hs, a);
    else
      hipLaunchKernelGGL((knarpe_attn_kernel<4, false, false, true>), grid, block, 0, hs, a);
  }
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_knarpe_attn_fwd(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                   int n_src, const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo,
                                   uint8_t* row_no_valid, const float* freqs_xy, const float* freqs_yaw, void* stream) {
  return tbx_knarpe_attn_fwd_dropout_tb(qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, out, ldo, row_no_valid,
                                        freqs_xy, freqs_yaw, 0.f, nullptr, 0u, 1, 0, stream);
}

extern "C" int tbx_knarpe_attn_bwd_dropout_tb(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias,
                                              int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg, const float* dout,
                                              int ldo, float* dqbuf, float* const* dkv, float* dbias_k, const float* freqs_xy,
                                              const float* freqs_yaw, float p_drop, const uint64_t* drop_seed, uint32_t drop_call,
                                              int time_batch, int time0, void* stream) {
  if (!dout || !dqbuf || !dkv) return TBX_ERR_ARG;  // (dbias_k may be NULL: not computed)
  if ((((uintptr_t)dout) & 15) || (((uintptr_t)dqbuf) & 15)) return TBX_ERR_ALIGN;
  AttnBwdArgs b;
  int rc = fill_args(b.f, qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, ldo, freqs_xy, freqs_yaw);
  if (rc != TBX_OK) return rc;
  if (segs[0].kv_bf16 != 0) return TBX_ERR_UNSUPPORTED;  // bf16 K/V tables: forward only
  rc = set_dropout(b.f, p_drop, drop_seed, drop_call, time_batch, time0);
  if (rc != TBX_OK) return rc;
  for (int i = 0; i < n_seg; ++i) {
    if (!dkv[i]) return TBX_ERR_ARG;
    b.dkv[i] = dkv[i];
  }
  if (n_seg == 1) b.dkv[1] = b.dkv[0];
  b.dout = dout;
  b.dqbuf = dqbuf;
  b.dbias_k = dbias_k;
  b.coef = nullptr;
  if (dbias_k != nullptr) hipLaunchKernelGGL(knarpe_attn_bwd_kernel<true>, dim3((b.f.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, b);
  else hipLaunchKernelGGL(knarpe_attn_bwd_kernel<false>, dim3((b.f.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, b);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_knarpe_attn_bwd_gather_tb(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias,
                                             int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg, const float* dout,
                                             int ldo, float* dqbuf, float* const* dkv, float* dbias_k, const float* freqs_xy,
                                             const float* freqs_yaw, float p_drop, const uint64_t* drop_seed, uint32_t drop_call,
                                             int time_batch, int time0, const int32_t* const* inv_ptr,
                                             const int32_t* const* inv_list, float* coef, void* stream) {
  if (!dout || !dqbuf || !dkv || !inv_ptr || !inv_list || !coef) return TBX_ERR_ARG;  // (dbias_k may be NULL: not computed)
  if ((((uintptr_t)dout) & 15) || (((uintptr_t)dqbuf) & 15)) return TBX_ERR_ALIGN;
  AttnBwdArgs b;
  int rc = fill_args(b.f, qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, ldo, freqs_xy, freqs_yaw);
  if (rc != TBX_OK) return rc;
  if (segs[0].kv_bf16 != 0) return TBX_ERR_UNSUPPORTED;  // bf16 K/V tables: forward only
  rc = set_dropout(b.f, p_drop, drop_seed, drop_call, time_batch, time0);
  if (rc != TBX_OK) return rc;
  DkvArgs d;
  d.qbuf = qbuf, d.dout = dout, d.coef = coef;
  d.ldq = ldq, d.q_off = q_off, d.ldo = ldo;
  int t_off = 0, tok = 0;
  for (int i = 0; i < 2; ++i) {
    const int s = i < n_seg ? i : 0;
    if (i < n_seg && (!dkv[i] || !inv_ptr[i] || !inv_list[i] || (((uintptr_t)dkv[i]) & 15) || n_batch % segs[i].batch_div))
      return TBX_ERR_ARG;
    b.dkv[i] = dkv[s];
    d.dkv[i] = dkv[s], d.inv_ptr[i] = inv_ptr[s], d.inv_list[i] = inv_list[s];
    d.k[i] = segs[s].k, d.n_tgt[i] = segs[s].n_tgt, d.ld_kv[i] = segs[s].ld_kv, d.k_off[i] = segs[s].k_off, d.v_off[i] = segs[s].v_off;
    d.list_cap[i] = n_src * segs[s].batch_div * segs[s].k;
    d.t_off[i] = i < n_seg ? t_off : 0;
    d.tok0[i] = i < n_seg ? tok : 0x7fffffff;
    if (i < n_seg) {
      t_off += segs[i].k;
      tok += (n_batch / segs[i].batch_div) * segs[i].n_tgt;
    }
  }
  d.ktot = t_off;
  d.n_tok = tok;
  d.xcd = (attn_xcd_mask() >> 1) & 1;
  b.f.xcd = d.xcd;
  b.dout = dout;
  b.dqbuf = dqbuf;
  b.dbias_k = dbias_k;
  b.coef = coef;
  hipStream_t hs = (hipStream_t)stream;
  if (dbias_k != nullptr) hipLaunchKernelGGL(knarpe_attn_bwd_kernel<true>, dim3((b.f.n_rows + 3) / 4), dim3(256), 0, hs, b);
  else hipLaunchKernelGGL(knarpe_attn_bwd_kernel<false>, dim3((b.f.n_rows + 3) / 4), dim3(256), 0, hs, b);
  if (hipGetLastError() != hipSuccess) return TBX_ERR_LAUNCH;
  hipLaunchKernelGGL(knarpe_attn_dkv_kernel, dim3((d.n_tok + 3) / 4), dim3(256), 0, hs, d);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_knarpe_attn_bwd(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                   int n_src, const tbx_attn_seg_t* segs, int n_seg, const float* dout, int ldo, float* dqbuf,
                                   float* const* dkv, float* dbias_k, const float* freqs_xy, const float* freqs_yaw,
                                   void* stream) {
  return tbx_knarpe_attn_bwd_dropout_tb(qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, dout, ldo, dqbuf, dkv,
                                        dbias_k, freqs_xy, freqs_yaw, 0.f, nullptr, 0u, 1, 0, stream);
}

// The per-call forms (time_batch = 1, time0 = 0): the mask is keyed by (call, row, target slot, head) alone.
extern "C" int tbx_knarpe_attn_fwd_dropout(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias,
                                           int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo,
                                           uint8_t* row_no_valid, const float* freqs_xy, const float* freqs_yaw, float p_drop,
                                           const uint64_t* drop_seed, uint32_t drop_call, void* stream) {
  return tbx_knarpe_attn_fwd_dropout_tb(qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, out, ldo, row_no_valid,
                                        freqs_xy, freqs_yaw, p_drop, drop_seed, drop_call, 1, 0, stream);
}

extern "C" int tbx_knarpe_attn_bwd_dropout(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias,
                                           int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg, const float* dout,
                                           int ldo, float* dqbuf, float* const* dkv, float* dbias_k, const float* freqs_xy,
                                           const float* freqs_yaw, float p_drop, const uint64_t* drop_seed, uint32_t drop_call,
                                           void* stream) {
  return tbx_knarpe_attn_bwd_dropout_tb(qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, dout, ldo, dqbuf, dkv,
                                        dbias_k, freqs_xy, freqs_yaw, p_drop, drop_seed, drop_call, 1, 0, stream);
}

extern "C" int tbx_knarpe_attn_bwd_gather(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias,
                                          int n_batch, int n_src, const tbx_attn_seg_t* segs, int n_seg, const float* dout,
                                          int ldo, float* dqbuf, float* const* dkv, float* dbias_k, const float* freqs_xy,
                                          const float* freqs_yaw, float p_drop, const uint64_t* drop_seed, uint32_t drop_call,
                                          const int32_t* const* inv_ptr, const int32_t* const* inv_list, float* coef,
                                          void* stream) {
  return tbx_knarpe_attn_bwd_gather_tb(qbuf, ldq, q_off, qt_off, rpe_k_bias, n_batch, n_src, segs, n_seg, dout, ldo, dqbuf, dkv,
                                       dbias_k, freqs_xy, freqs_yaw, p_drop, drop_seed, drop_call, 1, 0, inv_ptr, inv_list, coef,
                                       stream);
}
