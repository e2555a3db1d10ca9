// tbx_knarpe_dec_mid: the attention half of a dec_cross_attn transformer layer (transformer_rpe.py:207-233) as ONE launch for
// launches of a few hundred rows (the closed loop at one or a few scenes):
//
//   self attention over the K nearest tokens (K/V rows of the table the previous launch stored)         [tbx_knarpe_attn_fwd_folded]
//   x += rows without a valid target ? 0 : out_proj(.)                                                 [chain: LINEAR, accumulate, row skip]
//   h = LayerNorm_1(x);  q = W_q h + b_q;  qt_h = W_rpe_k,h^T q_h                                        [chain: LN, LINEAR, grouped LINEAR]
//   cross attention over the K nearest map tokens ++ K nearest traffic lights, value fold applied       [tbx_knarpe_attn_fwd_folded]
//
// i.e. what ran as attention kernel -> row chain -> attention kernel: three dependent launches of ~12 + 15 + 15 us whose
// work is a few hundred kFLOP per row. One workgroup (4 wavefronts) per row; the two target sweeps are attn_core.h's (4 waves
// split a row's targets exactly like the stand-alone kernel), the four small GEMVs between them run as thread-per-output v_fma
// chains in the MFMA sequence's k order on tbx_pack_weight_gemv images streamed into two LDS slots by LDS-DMA (see
// csrc/rowchain.hip linear_gemv) - every number equals the three-launch path's bit for bit (tested), the row never leaves the CU
// between the two attentions, and q / W_k^T q / the first attention's output make no round trip through global memory.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "attn_core.h"
#include "tbx_common.h"
#include "tile_core.h"
#include "step_core.h"

namespace {

using namespace tbx_attn;

constexpr int OUTW = D + NH * DR;        // 640
constexpr int RED = OUTW + 2 * NH;       // per-wave partial: sums + (M, L) per head
constexpr int IMG128 = (1 + 8 * 4) * 512;      // floats of a 128-output, k = 128 image: bias row + 32 rows of [128][4] (66 KiB)
constexpr int IMGKF = 4 * (1 + 2 * 4) * 512;   // floats of the 4 x (32 -> 128) query-side fold image: 4 column blocks of 9 rows (72 KiB)

struct Sweep {
  tbx_attn_seg_t seg[2];
  int n_seg;
  float scale2;
};

struct MidArgs {
  const float* qkv;  // [rows, ld_qkv]: q at q_off, W_k^T q at qt_off (this layer's self attention)
  float* x;          // [rows, 128] token rows, updated in place
  Sweep self, cross;
  const float *bias_k1, *bias_k2, *fxy, *fyaw;
  const float *fold1, *wo, *wq, *wkf, *fold2;  // gemv images
  const float *ln_w, *ln_b;
  float* out2;       // [rows, ld_out2]: folded cross-attention output
  uint8_t* flag2;
  float ln_eps;
  int ld_qkv, q_off, qt_off, ld_out2, n_rows, n_src;
  // tbx_knarpe_dec_layer: the rest of the layer in the same launch (wo2 != NULL)
  const float *wo2, *w1, *w2, *wqkv, *wqt;  // gemv images: cross out_proj, linear1, linear2, next layer's in_proj (q|k|v), its query-side fold
  const float *ln2_w, *ln2_b, *ln3_w, *ln3_b;
  const uint8_t* src_invalid;
  float* qkv_out;  // [rows, ld_qkv_out]: q | k | v | W_k^T q (896 columns) of the next layer's self attention, or NULL (last layer)
  uint16_t* kv16_out;  // [rows, 256] bfloat16 copy of k | v (the next layer's self K/V table with bf16 tables), or NULL
  // heads tail (last layer of the agents' block, tbx_heads_tail_t): hw[0..2] = add_navi.mlp, hw[3..5] = add_latent.mlp,
  // hw[6..8] = the action head's three stacked stages (gemv images); NULL hw[0]: none
  const float* hw[9];
  const float *navi_emb, *latent_emb;
  const uint8_t *navi_valid, *latent_invalid, *type_mask;
  float* action_out;
  int mask_stride;
  float ln2_eps, ln3_eps;
  int ld_qkv_out;
  int tail_mfma;  // every LINEAR stage on the split-bf16 matrix path: all images are tbx_pack_weight_mfma32 images (dec_layer_mf_kernel)
  // fused step tail (tbx_heads_tail_t.sim_state / next_prep): the row's agent's simulation step and next feature preparation
  int fused_tail, sim_parts;
  tbx_sim_state_t sim;
  tbx_agent_prep_args_t prep;
  // the lights' tail (tbx_tl_tail_t): tl.kv_out != NULL
  tbx_tl_tail_t tl;
};

#ifdef TBX_STAGE_CLOCK
__device__ unsigned long long g_mid_clk[256 * 16];
__device__ unsigned int g_mid_launch;
#define MID_CLK(i)                                                                                     \
  do {                                                                                                 \
    if (blockIdx.x == 0 && threadIdx.x == 0 && mid_slot < 256u) g_mid_clk[mid_slot * 16 + (i)] = clock64(); \
  } while (0)
#else
#define MID_CLK(i)
#endif

// image `img` (n_pieces KiB) -> LDS `slot`, pieces dealt to waves w0..3. w0 = 2 for an image requested right before a GEMV chain:
// threads 0..127 (waves 0, 1) run the chains, and a wave's LDS reads wait behind its own LDS-DMA (measured:
// tools/scratch/l2_stream_probe.hip) - the chain would stall for the image's whole flight. w0 = 0 where a sweep follows (no LDS
// reads until the epilogue's vmcnt(0)): four issuing waves land an image sooner than two.
__device__ __forceinline__ void dma_image(const float* img, int n_pieces, float* slot, int wave, int lane, int w0, int nw) {
  if (wave < w0) return;
  const uint32_t lds0 = lds_addr(slot);
  for (int p = wave - w0; p < n_pieces; p += nw - w0) glds_1k(img + p * 256 + lane * 4, lds0 + (uint32_t)p * 1024u);
}

// LayerNorm of a 128-float row in LDS by one wavefront, in the chain's ln_row<2> order (rowchain.hip): src -> dst
__device__ __forceinline__ void ln_row128(const float* src, float* dst, int lane, float eps, const float (&g)[2], const float (&bt)[2]) {
  float v[2];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    v[q] = src[lane + 64 * q];
    sum += v[q];
  }
  sum = tbx::wave_sum(sum);
  const float mean = sum / (float)D;
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float d = v[q] - mean;
    var += d * d;
  }
  var = tbx::wave_sum(var) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int q = 0; q < 2; ++q) dst[lane + 64 * q] = (v[q] - mean) * rstd * g[q] + bt[q];
}

// The stand-alone kernel's epilogue for 4 waves per row with the value fold (attn.hip, FOLD): per-wave partials -> red_s, the
// waves' combination -> comb_s, then thread c < 128 runs the fold chain of output column c. Returns the folded value (threads
// c < 128) and whether the row had a valid target. `fold_blk` must have landed (caller waited for the DMA before the barrier).
__device__ __forceinline__ float combine_fold(RowAcc& st, const float (&M)[NH], const float (&L)[NH], float (*red_s)[RED], float* comb_s,
                                              const float* fold_blk, int wir, int lane, int s8, int tg, bool& any_valid, int nt) {
  // (nt threads in the workgroup: 256, or 512 where waves 4..7 only fetch weight images - they hold no partials)
  if (wir < 4 && tg == 0) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      *(float4*)(&red_s[wir][h * DH + s8 * 4]) = st.oacc[h];
      st.eacc[h].store(&red_s[wir][D + h * DR], s8);
    }
  }
  if (wir < 4 && lane < NH) {
    red_s[wir][OUTW + lane] = M[lane];
    red_s[wir][OUTW + NH + lane] = L[lane];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of every image in flight have landed
  __syncthreads();
  float inv_l[NH], fw[4][NH];
  any_valid = false;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red_s[w][OUTW + h]);
    float ll = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = red_s[w][OUTW + h];
      fw[w][h] = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - mm);
      ll = __builtin_fmaf(fw[w][h], red_s[w][OUTW + NH + h], ll);
    }
    any_valid = any_valid || mm > -INFINITY;
    inv_l[h] = (mm > -INFINITY) ? 1.0f / ll : 0.f;
  }
  for (int c = threadIdx.x; c < OUTW; c += nt) {
    const int h = c < D ? c / DH : (c - D) / DR;
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc = __builtin_fmaf(fw[w][h], red_s[w][c], acc);
    comb_s[c] = acc * inv_l[h];
  }
  __syncthreads();
  float out = 0.f;
  if (threadIdx.x < D) {
    const int c = threadIdx.x, h = c / DH;
    out = gemv_chain(fold_blk, c, comb_s + D + h * DR, DR / 16, fold_blk[c * 4] + comb_s[c]);
  }
  return out;
}

// NW = 4: tbx_knarpe_dec_mid. NW = 8: tbx_knarpe_dec_layer - waves 4..7 take no part in the sweeps; they (and waves 2, 3) fetch the
// tail's 13 weight chunks, six requesting waves instead of two (the tail was bound by the two waves' DMA issue: 1.46 us per chunk).
template <bool KV16, int NW>
__global__ __launch_bounds__(NW * 64) void dec_mid_kernel(const MidArgs a) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* slot_a = lds;                       // 66 KiB: fold1 -> W_q -> fold2
  float* slot_b = slot_a + IMG128;           // 72 KiB: W_o -> query-side fold
  float(*red_s)[RED] = (float(*)[RED])(slot_b + IMGKF);
  float* comb_s = (float*)(red_s + 4);
  float* xs = comb_s + OUTW;   // the token row
  float* o1 = xs + D;          // folded self-attention output, then LN_1(x)
  float* q2 = o1 + D;          // q of the cross attention
  float* qt2 = q2 + D;         // W_k^T q, 4 x 128
  float* bk2_s = qt2 + NH * D; // the cross attention's rpe_k_bias (128 floats), parked in LDS until the cross phase
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wir = wave;
  const int row = blockIdx.x;
#ifdef TBX_STAGE_CLOCK
  unsigned mid_slot = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) mid_slot = atomicAdd(&g_mid_launch, 1u);
#endif
  MID_CLK(0);
  const int b = row / a.n_src;
  const int s8 = lane & 7, tg = lane >> 3;
  dma_image(a.fold1, IMG128 / 256, slot_a, wave, lane, 0, NW);
  dma_image(a.wo, IMG128 / 256, slot_b, wave, lane, 0, NW);
  if (threadIdx.x < D) xs[threadIdx.x] = a.x[(int64_t)row * D + threadIdx.x];
  // LayerNorm parameters (wave 0 normalises the row): requested now - an ordinary load issued while an image DMA is in flight
  // makes the compiler wait for everything outstanding (it does not see the DMAs, vmcnt is in order)
  float ln_g[2] = {0.f, 0.f}, ln_bt[2] = {0.f, 0.f};
  if (wave == 0) {
    ln_g[0] = a.ln_w[lane], ln_g[1] = a.ln_w[64 + lane];
    ln_bt[0] = a.ln_b[lane], ln_bt[1] = a.ln_b[64 + lane];
  }
  // ... and so is the cross attention's rpe_k_bias (its q . b_k term): to LDS, not to 16 registers that would live across the self sweep
  if (threadIdx.x < D) bk2_s[threadIdx.x] = a.bias_k2[threadIdx.x];
  EFreq fq;
  fq.init(a.fxy, a.fyaw, s8);
  float4 qv[NH];
  ESlice qt[NH];
  float qb[NH];
  // ---------------------------------------------------------------- self attention
  {
    const float* qrow = a.qkv + (int64_t)row * a.ld_qkv;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
      const float4 bk = *(const float4*)(a.bias_k1 + h * DH + s8 * 4);
      qb[h] = tbx::group8_sum(dot4(qv[h], bk));
      qt[h].load(qrow + a.qt_off + h * DR, s8);
    }
  }
  bool valid1;
  MID_CLK(1);
  {
    RowAcc st;
    st.zero();
    float M[NH] = {0.f, 0.f, 0.f, 0.f}, L[NH] = {0.f, 0.f, 0.f, 0.f};
    if (NW == 4 || wave < 4) {
      sweep<4, false, KV16>(a.self, row, b, wir, s8, tg, qv, qt, qb, fq, st);
      merge_slots(st, M, L);
    }
    MID_CLK(2);
    const float f = combine_fold(st, M, L, red_s, comb_s, slot_a, wir, lane, s8, tg, valid1, NT);
    if (threadIdx.x < D) o1[threadIdx.x] = f;
  }
  __syncthreads();  // o1 complete; slot A (fold1) is free
  MID_CLK(3);
  dma_image(a.wq, IMG128 / 256, slot_a, wave, lane, 2, NW);
  // ---------------------------------------------------------------- x += no valid target ? 0 : out_proj(o1)   (W_o landed: waited in combine_fold)
  if (threadIdx.x < D) {
    const int c = threadIdx.x;
    const float v = gemv_chain(slot_b, c, o1, D / 16, slot_b[c * 4] + xs[c]);
    if (valid1) xs[c] = v;
  }
  __syncthreads();  // xs updated; slot B (W_o) is free
  MID_CLK(4);
  dma_image(a.wkf, IMGKF / 256, slot_b, wave, lane, 2, NW);
  // ---------------------------------------------------------------- LN_1(x) -> o1 (one wavefront, the chain's ln_row<2> order)
  if (wave == 0) {
    float v[2];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      v[q] = xs[lane + 64 * q];
      sum += v[q];
    }
    sum = tbx::wave_sum(sum);
    const float mean = sum / (float)D;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float d = v[q] - mean;
      var += d * d;
    }
    var = tbx::wave_sum(var) / (float)D;
    const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
    for (int q = 0; q < 2; ++q) o1[lane + 64 * q] = (v[q] - mean) * rstd * ln_g[q] + ln_bt[q];
  }
  // W_q has landed once at most the 72 / (NW - 2) pieces per issuing wave of the image requested after it are outstanding (DMA loads only: in order)
  MID_CLK(5);
  if constexpr (NW == 4)
    asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __syncthreads();
  MID_CLK(6);
  // ---------------------------------------------------------------- q = W_q LN(x) + b_q
  if (threadIdx.x < D) {
    const int c = threadIdx.x;
    q2[c] = gemv_chain(slot_a, c, o1, D / 16, slot_a[c * 4]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the query-side fold image has landed
  __syncthreads();  // q2 complete; slot A (W_q) is free
  MID_CLK(7);
  // ---------------------------------------------------------------- qt_h = W_rpe_k,h^T q_h: 4 x (32 -> 128), two outputs per thread
#pragma unroll
  for (int o = (int)threadIdx.x; o < NH * D; o += NT) {
    const int g = o >> 7, c = o & (D - 1);
    const float* blk = slot_b + g * (1 + 2 * 4) * 512;
    qt2[o] = gemv_chain(blk, c, q2 + g * DH, 2, blk[c * 4]);
  }
  __syncthreads();
  MID_CLK(8);
  dma_image(a.fold2, IMG128 / 256, slot_a, wave, lane, 0, NW);  // lands during the sweep
  if (a.wo2 != nullptr) dma_image(a.wo2, IMG128 / 256, slot_b, wave, lane, 0, NW);  // (slot B is free: the tail's first chunk rides along)
  // ---------------------------------------------------------------- cross attention
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qv[h] = *(const float4*)(q2 + h * DH + s8 * 4);
    qb[h] = tbx::group8_sum(dot4(qv[h], *(const float4*)(bk2_s + h * DH + s8 * 4)));
    qt[h].load(qt2 + h * DR, s8);
  }
  {
    RowAcc st;
    st.zero();
    float M[NH] = {0.f, 0.f, 0.f, 0.f}, L[NH] = {0.f, 0.f, 0.f, 0.f};
    if (NW == 4 || wave < 4) {
      sweep<4, false, KV16>(a.cross, row, b, wir, s8, tg, qv, qt, qb, fq, st);
      merge_slots(st, M, L);
    }
    MID_CLK(9);
    bool valid2;
    const float f = combine_fold(st, M, L, red_s, comb_s, slot_a, wir, lane, s8, tg, valid2, NT);
    if (a.wo2 == nullptr) {
      if (threadIdx.x < D) {
        a.out2[(int64_t)row * a.ld_out2 + threadIdx.x] = f;
        a.x[(int64_t)row * D + threadIdx.x] = xs[threadIdx.x];  // the token row after the self-attention residual
      }
      if (threadIdx.x == 0) a.flag2[row] = valid2 ? 0 : 1;
      MID_CLK(10);
      return;
    }
    // ================================================================ tbx_knarpe_dec_layer: the layer's second half in this launch
    // (transformer_rpe.py:233-237 + the next layer's projections, attention_rpe.py:92-98,147) - the stages of the row chain that
    // followed tbx_knarpe_dec_mid, in its arithmetic: LINEAR = bias (+ destination) then the k-ordered fma chain (gemv_chain),
    // LayerNorm = ln_row128. 13 weight chunks alternate through the two LDS slots: waves 2-3 request chunk i + 1 when chunk i's
    // multiply starts (threads 0..127 multiply; a wave's LDS reads would wait behind its own DMA).
    if (threadIdx.x < D) o1[threadIdx.x] = f;
    float* u = comb_s;  // FFN hidden row (512 floats)
    const bool x_valid = a.src_invalid[row] == 0;
    float lg2[2] = {0.f, 0.f}, lb2[2] = {0.f, 0.f}, lg3[2] = {0.f, 0.f}, lb3[2] = {0.f, 0.f};
    if (wave == 0) {
      lg2[0] = a.ln2_w[lane], lg2[1] = a.ln2_w[64 + lane], lb2[0] = a.ln2_b[lane], lb2[1] = a.ln2_b[64 + lane];
      if (a.qkv_out != nullptr) lg3[0] = a.ln3_w[lane], lg3[1] = a.ln3_w[64 + lane], lb3[0] = a.ln3_b[lane], lb3[1] = a.ln3_b[64 + lane];
    }
    // chunk i of the tail -> (image pointer, float4-row offset, rows); slot = B for even i, A for odd i
    const bool heads = a.hw[0] != nullptr && a.qkv_out == nullptr;
    const int n_chunks = a.qkv_out != nullptr ? 13 : (heads ? 9 + 15 : 9);
    auto request = [&](int i) {
      if (i >= n_chunks || wave < 2) return;
      const float* img;
      int row0, rows;
      if (i == 0) img = a.wo2, row0 = 0, rows = 33;
      else if (i <= 4) img = a.w1, row0 = (i - 1) * 33, rows = 33;
      else if (i <= 8) img = a.w2, row0 = i == 5 ? 0 : 1 + (i - 5) * 32, rows = i == 5 ? 33 : 32;
      else if (heads) {
        // heads chunk h = i - 9: the two adders' 256 -> 128 stages take two k-chunks, the action head's 384-wide stages three
        // column blocks
        const int h = i - 9;
        const int im = h < 2 ? 0 : h < 4 ? h - 1 : h < 6 ? 3 : h < 8 ? h - 2 : h < 11 ? 6 : h < 14 ? 7 : 8;
        img = a.hw[im];
        const bool second = h == 1 || h == 5;
        row0 = second ? 33 : (h >= 8 && h < 14 ? ((h - 8) % 3) * 33 : 0);
        rows = second ? 32 : 33;
      } else if (i <= 11) img = a.wqkv, row0 = (i - 9) * 33, rows = 33;
      else img = a.wqt, row0 = 0, rows = 36;
      const uint32_t lds0 = lds_addr((i & 1) ? slot_a : slot_b);
      for (int p = wave - 2; p < rows * 2; p += NW - 2) glds_1k(img + (size_t)row0 * 512 + p * 256 + lane * 4, lds0 + (uint32_t)p * 1024u);
    };
    auto landed = [&]() {  // the chunk requested last has landed (requesting waves wait for their pieces; the barrier collects them)
      if (wave >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    };
    // ---- chunk 0 (requested before the cross sweep, landed with combine_fold's vmcnt(0)): x += row without a valid cross target ? 0 : out_proj(f)
    __syncthreads();  // o1 = f complete; slot A is free (fold2 was consumed by combine_fold)
    request(1);
    if (threadIdx.x < D) {
      const int c = threadIdx.x;
      const float v = gemv_chain(slot_b, c, o1, D / 16, slot_b[c * 4] + xs[c]);
      if (valid2) xs[c] = v;
    }
    __syncthreads();
    MID_CLK(10);
    if (wave == 0) ln_row128(xs, o1, lane, a.ln2_eps, lg2, lb2);  // h = norm2(x)
    MID_CLK(11);
    // ---- chunks 1..4: u = relu(linear1(h))
#pragma unroll 1
    for (int i = 1; i <= 4; ++i) {
      landed();  // (also: h / the previous block's u are complete)
      request(i + 1);
      if (threadIdx.x < D) {
        const int c = threadIdx.x;
        const float* blk = (i & 1) ? slot_a : slot_b;
        u[(i - 1) * D + c] = fmaxf(gemv_chain(blk, c, o1, D / 16, blk[c * 4]), 0.f);
      }
    }
    MID_CLK(12);
    // ---- chunks 5..8: x += linear2(u), one k-chunk of 128 at a time; invalid source rows come out as 0
    float acc = 0.f;
#pragma unroll 1
    for (int i = 5; i <= 8; ++i) {
      landed();
      request(i + 1);
      if (threadIdx.x < D) {
        const int c = threadIdx.x;
        const float* blk = (i & 1) ? slot_a : slot_b;
        if (i == 5) {
          acc = gemv_chain(blk, c, u, D / 16, blk[c * 4] + xs[c]);
        } else {
          acc = gemv_chain(blk - 512, c, u + (i - 5) * D, D / 16, acc);  // (no bias row in this chunk: row q sits at slot row q)
        }
      }
    }
    if (threadIdx.x < D) {
      const float v = x_valid ? acc : 0.f;
      xs[threadIdx.x] = v;
      a.x[(int64_t)row * D + threadIdx.x] = v;
    }
    MID_CLK(13);
    if (heads) {
      // ============================================================ the agents' heads in the last layer's launch: the stages of the
      // heads chain (traffic_bots.py:206-221): x += navi_valid ? mlp([x | navi_emb]) : 0 (add_navi_latent.py:52-65), the same with
      // the latent embedding, the action head's three per-type branches as stacked / block-diagonal stages and their masked sum
      // (action_head.py:74-100). Both embeddings arrive with their invalid rows already zeroed. Same gemv_chain arithmetic.
      float* cat = comb_s;          // [x | z] (256), later the action head's first hidden layer (384)
      float* hb = qt2;              // hidden rows of the adders (128), later the action head's second hidden layer (384)
      float zn = 0.f, zl = 0.f;
      if (threadIdx.x < D) zn = a.navi_emb[(int64_t)row * D + threadIdx.x], zl = a.latent_emb[(int64_t)row * D + threadIdx.x];
      const bool ok_navi = a.navi_valid[row] != 0, ok_lat = a.latent_invalid[row] == 0;
      int ci = 9;  // chunk index: slot = A for odd, B for even
      auto blk_of = [&](int i) -> const float* { return (i & 1) ? slot_a : slot_b; };
#pragma unroll 1
      for (int adder = 0; adder < 2; ++adder) {
        if (threadIdx.x < D) cat[threadIdx.x] = xs[threadIdx.x], cat[D + threadIdx.x] = adder == 0 ? zn : zl;
        float acc2 = 0.f;
        landed();  // chunk ci (k 0..127 of the 256 -> 128 stage); cat complete
        request(ci + 1);
        if (threadIdx.x < D) acc2 = gemv_chain(blk_of(ci), threadIdx.x, cat, D / 16, blk_of(ci)[threadIdx.x * 4]);
        ++ci;
        landed();  // k 128..255
        request(ci + 1);
        if (threadIdx.x < D) o1[threadIdx.x] = fmaxf(gemv_chain(blk_of(ci) - 512, threadIdx.x, cat + D, D / 16, acc2), 0.f);
        ++ci;
        landed();
        request(ci + 1);
        if (threadIdx.x < D) hb[threadIdx.x] = fmaxf(gemv_chain(blk_of(ci), threadIdx.x, o1, D / 16, blk_of(ci)[threadIdx.x * 4]), 0.f);
        ++ci;
        landed();
        request(ci + 1);
        if (threadIdx.x < D) {
          const float v = fmaxf(gemv_chain(blk_of(ci), threadIdx.x, hb, D / 16, blk_of(ci)[threadIdx.x * 4]), 0.f);
          xs[threadIdx.x] = xs[threadIdx.x] + ((adder == 0 ? ok_navi : ok_lat) ? v : 0.f);
        }
        ++ci;
        __syncthreads();  // xs updated before it is copied / read again
      }
      // ---- action head layer 1: 128 -> 3 x 128 (three column blocks), relu
#pragma unroll 1
      for (int g = 0; g < 3; ++g, ++ci) {
        landed();
        request(ci + 1);
        if (threadIdx.x < D) cat[g * D + threadIdx.x] = fmaxf(gemv_chain(blk_of(ci), threadIdx.x, xs, D / 16, blk_of(ci)[threadIdx.x * 4]), 0.f);
      }
      // ---- layer 2: block-diagonal 3 x (128 -> 128), relu
#pragma unroll 1
      for (int g = 0; g < 3; ++g, ++ci) {
        landed();
        request(ci + 1);
        if (threadIdx.x < D) hb[g * D + threadIdx.x] = fmaxf(gemv_chain(blk_of(ci), threadIdx.x, cat + g * D, D / 16, blk_of(ci)[threadIdx.x * 4]), 0.f);
      }
      // ---- layer 3: block-diagonal 3 x (128 -> 16): output o = g * 16 + j of one 128-column block
      landed();
      if (threadIdx.x < D) {
        const int o = threadIdx.x, g = o < 48 ? o >> 4 : 0;
        o1[o] = gemv_chain(blk_of(ci), o, hb + g * D, D / 16, blk_of(ci)[o * 4]);
      }
      __syncthreads();
      if (threadIdx.x < 2) {  // the masked sum over the branches, in branch order from 0 (TBX_F_MASKED_SUM)
        float v = 0.f;
        for (int g = 0; g < 3; ++g)
          if (a.type_mask[(int64_t)g * a.mask_stride + row] == 0) v += o1[g * 16 + threadIdx.x];
        a.action_out[(int64_t)row * 2 + threadIdx.x] = v;
      }
      return;
    }
    if (a.qkv_out == nullptr) return;
    __syncthreads();
    if (wave == 0) ln_row128(xs, o1, lane, a.ln3_eps, lg3, lb3);  // the next layer's norm_src
    float* qrow_out = a.qkv_out + (int64_t)row * a.ld_qkv_out;
    // ---- chunks 9..11: q | k | v = in_proj(h)
#pragma unroll 1
    for (int i = 9; i <= 11; ++i) {
      landed();
      request(i + 1);
      if (threadIdx.x < D) {
        const int c = threadIdx.x;
        const float* blk = (i & 1) ? slot_a : slot_b;
        const float v = gemv_chain(blk, c, o1, D / 16, blk[c * 4]);
        if (i == 9) q2[c] = v;
        qrow_out[(i - 9) * D + c] = v;
        if (i > 9 && a.kv16_out != nullptr)  // round to nearest even, as the chain's TBX_F_OUT_BF16 store
          a.kv16_out[(int64_t)row * (2 * D) + (i - 10) * D + c] = __builtin_bit_cast(uint16_t, (__bf16)v);
      }
    }
    // ---- chunk 12: W_rpe_k^T q per head: 4 x (32 -> 128), two outputs per thread
    landed();
    MID_CLK(14);
#pragma unroll
    for (int o = (int)threadIdx.x; o < NH * D; o += NT) {
      const int g = o >> 7, c = o & (D - 1);
      const float* blk = slot_b + g * (1 + 2 * 4) * 512;
      qrow_out[3 * D + o] = gemv_chain(blk, c, q2 + g * DH, 2, blk[c * 4]);
    }
    MID_CLK(15);
  }
}

// ================================================================================================================================
// tbx_knarpe_dec_layer with tail_mfma32: the WHOLE layer's row-local arithmetic on the split-bf16 matrix path of tile_core.h.
// One workgroup (8 wavefronts) per source row as above; waves 0..3 run the two target sweeps (attn_core.h, unchanged). Every LINEAR
// - the two value folds, out_proj, q, W_k^T q, out_proj2, the FFN, the next layer's q | k | v | W_k^T q or the agents' heads - is a
// v_mfma_f32_16x16x32_bf16 product D = W x^T whose B operand is the ONE row (bf16 hi / lo planes in LDS: hi at P + 2 k, lo `lo`
// bytes behind; all 16 columns read the same row, column 0 is kept): wave w owns output tile w (channels 16 w .. 16 w + 15), its
// weights come as 8 KiB register units (tbx_pack_weight_mfma32) straight from global memory, TWO units ahead of their use through
// three register slots (not across a sweep: its state takes ~190 VGPRs of the 256 - the two units behind a sweep are requested
// when its partial sums have left the registers, and land under the waves' combination). No weight passes through LDS (the exact-
// fp32 form above stages 66 - 72 KiB images by LDS-DMA and runs 128-long dependent fma chains on 128 threads: ~1.15 us per 128 x 128
// stage); a stage costs its 64 KiB of weights through the CU's 64 B/clk L2 port (~0.43 us) - measured per stage in
// profiles/r03_dec_layer_phase_clock.txt.
// Unit sequence: 0 fold_self, 1 out_proj, 2 q, 3 qfold, 4 fold_cross, 5 out_proj2, 6..9 linear1, 10..13 linear2, then either
// 14..16 next in_proj (q, k, v), 17 next qfold, or the heads: 14..17 add_navi (k-chunk x, k-chunk embedding, layer 2, layer 3),
// 18..21 add_latent, 22..24 / 25..27 / 28 the action head's three stacked stages (csrc/tile_heads.hip's entries), or the lights' tail:
// 14..21 k, v of the agents' 4 layers, 22..24 the next-state predictor, or nothing.
namespace mf {
using tbx_tile::Acc;
using tbx_tile::bf16x4;
using tbx_tile::bf16x8;
using tbx_tile::f32x4;
using tbx_tile::u32x2;
using tbx_tile::W;

constexpr int NSW = 8;  // all 8 waves sweep: a row's targets 8 per pass per wave, two waves per SIMD hide each other's load latencies
constexpr int LO128 = 256, LO384 = 768, LO512 = 1024, LO640 = 1280;  // byte offset of the lo plane behind a K-wide hi plane

// (br: the action-head branch units 22 / 23 / 24 are taken from - the agents' heads only)
template <int N>
__device__ __forceinline__ void issue(W& w, const MidArgs& a, bool heads, int wave, int lane, int br = 0) {
  using tbx_tile::load_unit;
  if constexpr (N == 0) load_unit(w, a.fold1, wave, lane);
  else if constexpr (N == 1) load_unit(w, a.wo, wave, lane);
  else if constexpr (N == 2) load_unit(w, a.wq, wave, lane);
  else if constexpr (N == 3) load_unit(w, a.wkf, wave, lane);
  else if constexpr (N == 4) load_unit(w, a.fold2, wave, lane);
  else if constexpr (N == 5) load_unit(w, a.wo2, wave, lane);
  else if constexpr (N <= 9) load_unit(w, a.w1, 8 * (N - 6) + wave, lane);
  else if constexpr (N <= 13) load_unit(w, a.w2, 8 * (N - 10) + wave, lane);
  else if (a.qkv_out != nullptr) {
    if constexpr (N <= 16) load_unit(w, a.wqkv, 8 * (N - 14) + wave, lane);
    else if constexpr (N == 17) load_unit(w, a.wqt, wave, lane);
  } else if (a.tl.kv_out != nullptr) {  // 14..21: k, v of the agents' 4 layers; 22..24: the state predictor
    if constexpr (N >= 14 && N <= 21) load_unit(w, a.tl.kv_images[(N - 14) >> 1], 8 * ((N - 14) & 1) + wave, lane);
    else if constexpr (N == 22 || N == 23) load_unit(w, a.tl.mlp_images[N - 22], wave, lane);
    else if constexpr (N == 24) load_unit(w, a.tl.mlp_images[2], 0, lane);
  } else if (heads) {
    if constexpr (N == 14 || N == 18) load_unit(w, a.hw[N == 14 ? 0 : 3], wave, lane);
    else if constexpr (N == 15 || N == 19) load_unit(w, a.hw[N == 15 ? 0 : 3], 8 + wave, lane);
    else if constexpr (N == 16 || N == 17) load_unit(w, a.hw[N - 15], wave, lane);
    else if constexpr (N == 20 || N == 21) load_unit(w, a.hw[N - 16], wave, lane);
    else if constexpr (N == 22) {  // the action head's three layers of branch br (none: an invalid agent)
      if (br >= 0) load_unit(w, a.hw[6], 8 * br + wave, lane);
    } else if constexpr (N == 23) {
      if (br >= 0) load_unit(w, a.hw[7], 8 * br + wave, lane);
    } else if constexpr (N == 24) {
      if (wave == 0) load_unit(w, a.hw[8], br, lane);
    }
  }
}

// 4 values of the row -> planes (hi at P + 2 c, lo `lo` bytes behind)
__device__ __forceinline__ void put4(char* P, int lo, int c, const f32x4 v) {
  u32x2 hi, l;
  tbx_tile::split4(v, hi, l);
  *(u32x2*)(P + c * 2) = hi;
  *(u32x2*)(P + lo + c * 2) = l;
}

// one 32-k step of D += W x^T with the row's planes as the B operand
__device__ __forceinline__ void step(Acc& acc, const bf16x8 whi, const bf16x8 wlo, const char* P, int lo, int st, int g4) {
  const bf16x8 xh = *(const bf16x8*)(P + (st * 32 + g4 * 8) * 2);
  const bf16x8 xl = *(const bf16x8*)(P + lo + (st * 32 + g4 * 8) * 2);
  acc.hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xh, acc.hh, 0, 0, 0);
  acc.hl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xl, acc.hl, 0, 0, 0);
  acc.lh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, acc.lh, 0, 0, 0);
}

__device__ __forceinline__ f32x4 gemv4(const W& w, const char* P, int lo, int st0, int g4) {
  Acc acc;
  acc.zero();
#pragma unroll
  for (int st = 0; st < 4; ++st) step(acc, w.hi[st], w.lo[st], P, lo, st0 + st, g4);
  return acc.sum();
}

// LayerNorm of the 128-float row `src` by one wavefront (ln_row128's order) -> planes
__device__ __forceinline__ void ln_planes(const float* src, char* P, int lane, float eps, const float (&g)[2], const float (&bt)[2]) {
  float v[2];
  v[0] = src[lane], v[1] = src[lane + 64];
  const float mean = tbx::wave_sum(v[0] + v[1]) / (float)D;
  const float d0 = v[0] - mean, d1 = v[1] - mean;
  const float var = tbx::wave_sum(d0 * d0 + d1 * d1) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float y = (v[q] - mean) * rstd * g[q] + bt[q];
    const __bf16 h = (__bf16)y;
    *(__bf16*)(P + 2 * (lane + 64 * q)) = h;
    *(__bf16*)(P + LO128 + 2 * (lane + 64 * q)) = (__bf16)(y - (float)h);
  }
}

// attn_core.h's `sweep` for NSW waves per row with the FIRST pass of segment 0 handed in: its target index / mask are fetched at
// the top of the kernel and its K / V rows and embedding slice while the previous phase finishes (`Pre`), so the sweep starts on
// operands that are already in registers instead of behind two dependent memory round trips (~2500 shader clocks each at this
// occupancy, profiles/r03_attn_phase_clock.txt). Same passes in the same order as sweep<NSW>: the same sums.
struct Pre {
  float4 kq[4], v[4];
  ESlice e;  // the materialised embedding slice (seg.emb), or - in e.wc.x / y / z - the relative pose it is rebuilt from when the pass
             // runs (rebuilding it where it is requested would wait for the pose there)
};
__device__ __forceinline__ void pre_index(const Sweep& a, int row, int wir, int tg, int& j, bool& ok) {
  const tbx_attn_seg_t& S = a.seg[0];
  const int t = wir * 8 + tg;
  const bool active = t < S.k;
  const int64_t pi = (int64_t)row * S.k + (active ? t : S.k - 1);
  j = S.idx[pi];
  ok = (S.invalid[pi] == 0) & active;
}
// REL: every segment of the launch gives its pairs as relative poses (seg.emb == NULL: the default schedule) - known at compile time,
// because a run-time branch between "load the embedding" and "load the pose" makes the wait-count pass merge the two paths' pending
// loads: the pose path then waited for EVERYTHING in flight (a register of the other path's loads was re-used for its address).
template <bool KV16, bool REL>
__device__ __forceinline__ void pre_rows(const Sweep& a, int row, int b, int wir, int s8, int tg, int j, const EFreq& fq, Pre& p) {
  const tbx_attn_seg_t& S = a.seg[0];
  constexpr int ES = KV16 ? 2 : 1;
  const float* kvb = (const float*)((const char*)S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt * S.ld_kv) * (4 / ES));
  const float* trow = (const float*)((const char*)kvb + ((int64_t)j * S.ld_kv) * (4 / ES));
  const int t = wir * 8 + tg;
  const int64_t pi = (int64_t)row * S.k + (t < S.k ? t : S.k - 1);
  if constexpr (REL) {
    p.e.wc.x = S.rel_pose[pi * 3], p.e.wc.y = S.rel_pose[pi * 3 + 1], p.e.wc.z = S.rel_pose[pi * 3 + 2];
  } else {
    if (S.emb != nullptr)
      p.e.load(S.emb + pi * DR, s8);
    else
      p.e.wc.x = S.rel_pose[pi * 3], p.e.wc.y = S.rel_pose[pi * 3 + 1], p.e.wc.z = S.rel_pose[pi * 3 + 2];
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    p.kq[st] = kv_load4<KV16>(trow, S.k_off + st * 32 + s8 * 4);
    p.v[st] = kv_load4<KV16>(trow, S.v_off + st * 32 + s8 * 4);
  }
}
template <bool KV16, bool REL>
__device__ __forceinline__ void sweep_pf(const Sweep& a, int row, int b, int wir, int s8, int tg, const float4 (&qv)[NH], const ESlice (&qt)[NH],
                                         const float (&qb)[NH], const EFreq& fq, Pre& pre, bool pre_ok, RowAcc& st) {
  float(&m_run)[NH] = st.m_run;
  float(&l_run)[NH] = st.l_run;
  float4(&oacc)[NH] = st.oacc;
  ESlice(&eacc)[NH] = st.eacc;
  auto pass = [&](const bool ok, const float4(&kq)[4], const float4(&v)[4], const ESlice& e) {
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
      l_run[h] += pr;
      fma4(oacc[h], pr, v[h]);
      eacc[h].fma(pr, e);
    }
  };
  if (wir * 8 < a.seg[0].k) {
    if (REL || a.seg[0].emb == nullptr) {
      const float rel[3] = {pre.e.wc.x, pre.e.wc.y, pre.e.wc.z};
      fq.embed(rel, pre.e);
    }
    pass(pre_ok, pre.kq, pre.v, pre.e);
  }
  for (int sg = 0; sg < a.n_seg; ++sg) {
    const tbx_attn_seg_t& S = a.seg[sg];
    constexpr int ES = KV16 ? 2 : 1;
    const float* kvb = (const float*)((const char*)S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt * S.ld_kv) * (4 / ES));
    const int64_t pbase = (int64_t)row * S.k;
    for (int base = wir * 8 + (sg == 0 ? 8 * NSW : 0); base < S.k; base += 8 * NSW) {
      const int t = base + tg;
      const bool active = t < S.k;
      const int64_t pi = pbase + (active ? t : S.k - 1);
      const int j = S.idx[pi];
      const bool ok = (S.invalid[pi] == 0) & active;
      const float* trow = (const float*)((const char*)kvb + ((int64_t)j * S.ld_kv) * (4 / ES));
      float4 kq[4], v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        kq[q] = kv_load4<KV16>(trow, S.k_off + q * 32 + s8 * 4);
        v[q] = kv_load4<KV16>(trow, S.v_off + q * 32 + s8 * 4);
      }
      ESlice e;
      if constexpr (REL) fq.embed(S.rel_pose + pi * 3, e);
      else load_e(S, pi, s8, fq, e);
      pass(ok, kq, v, e);
    }
  }
}

// combine_fold's first half for 8 sweeping waves: their partials -> the row's normalised sums comb_s [640] (fp32) and their planes Pc
template <class F>
__device__ __forceinline__ void combine(float Mh, float Lh, const float4& o, const ESlice& e, float (*red_s)[RED], float* comb_s, char* Pc,
                                        int wir, int lane, int s8, bool& any_valid, F&& request) {
  // (merge_slots_by_head: lane l holds head l >> 4's sums of channel slice s8; lanes l and l ^ 8 the same)
  const int hq = lane >> 4;
  if ((lane & 8) == 0) {
    *(float4*)(&red_s[wir][hq * DH + s8 * 4]) = o;
    e.store(&red_s[wir][D + hq * DR], s8);
  }
  if ((lane & 15) == 0) {
    red_s[wir][OUTW + hq] = Mh;
    red_s[wir][OUTW + NH + hq] = Lh;
  }
  request();  // (the sweep's sums have left the registers: the next two weight units fly under the combination)
  __syncthreads();
  {  // (a valid target gives every head a score: head 0's maxima tell)
    float mm0 = -INFINITY;
#pragma unroll
    for (int w = 0; w < NSW; ++w) mm0 = fmaxf(mm0, red_s[w][OUTW]);
    any_valid = mm0 > -INFINITY;
  }
  if (threadIdx.x < OUTW / 4) {  // 160 threads, 4 consecutive columns each (one head's)
    const int c = (int)threadIdx.x * 4;
    const int h = c < D ? c / DH : (c - D) / DR;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NSW; ++w) mm = fmaxf(mm, red_s[w][OUTW + h]);
    float ll = 0.f, fw[NSW];
#pragma unroll
    for (int w = 0; w < NSW; ++w) {
      const float mw = red_s[w][OUTW + h];
      fw[w] = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - mm);
      ll = __builtin_fmaf(fw[w], red_s[w][OUTW + NH + h], ll);
    }
    const float inv_l = (mm > -INFINITY) ? 1.0f / ll : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NSW; ++w) {
      const f32x4 r = *(const f32x4*)(&red_s[w][c]);
      acc[0] = __builtin_fmaf(fw[w], r[0], acc[0]), acc[1] = __builtin_fmaf(fw[w], r[1], acc[1]);
      acc[2] = __builtin_fmaf(fw[w], r[2], acc[2]), acc[3] = __builtin_fmaf(fw[w], r[3], acc[3]);
    }
    acc *= inv_l;
    if (c < D) *(f32x4*)(comb_s + c) = acc;  // (sum a v: the fold's fp32 addend)
    put4(Pc, LO640, c, acc);
  }
  __syncthreads();
}
}  // namespace mf

template <bool KV16, bool REL>
__global__ __launch_bounds__(512) void dec_layer_mf_kernel(const MidArgs a) {
  using namespace mf;
  __shared__ __attribute__((aligned(16))) float red_s[NSW][RED];
  __shared__ __attribute__((aligned(16))) float comb_s[D], xs[D], q2[D], qt2[NH * D], bk2_s[D], head_o[3 * 2];
  __shared__ __attribute__((aligned(16))) char Pc[2 * LO640], Ph[2 * LO128], Pq[2 * LO128], Pu[2 * LO512], Pv[2 * LO384];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wir = wave;
  const int row = blockIdx.x;
#ifdef TBX_STAGE_CLOCK
  unsigned mid_slot = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) mid_slot = atomicAdd(&g_mid_launch, 1u);
#endif
  MID_CLK(0);
  const int b = row / a.n_src;
  const int s8 = lane & 7, tg = lane >> 3;
  // the target index / mask of BOTH sweeps' first pass (the K-nearest sets are inputs of the launch): the kernel's first requests, so
  // that the self sweep's row gathers leave before anything waits for the query side (they sat behind two dependent round trips)
  int j1, j2;
  bool ok1, ok2;
  pre_index(a.self, row, wir, tg, j1, ok1);
  pre_index(a.cross, row, wir, tg, j2, ok2);
  const int g4 = lane >> 4;
  const bool col0 = (lane & 15) == 0;  // the lanes that hold column 0 = the row: channels c_out .. c_out + 3 of a 128-wide stage
  const int c_out = 16 * wave + 4 * g4;
  const bool heads = a.hw[0] != nullptr && a.qkv_out == nullptr;
  // (the step counter of the fused tail: requested here, a round trip to memory before that tail needs it)
  const int t_step_v = a.fused_tail ? *a.sim.step : 0;
  W wb[3];
  // (the row and the cross bias go to LDS behind the query-side requests: a load straight into an LDS write is waited for on the spot)
  float x_r = 0.f, bk2_r = 0.f;
  if (threadIdx.x < D) x_r = a.x[(int64_t)row * D + threadIdx.x], bk2_r = a.bias_k2[threadIdx.x];
  float lg1[2] = {0.f, 0.f}, lb1[2] = {0.f, 0.f}, lg2[2] = {0.f, 0.f}, lb2[2] = {0.f, 0.f}, lg3[2] = {0.f, 0.f}, lb3[2] = {0.f, 0.f};
  // (LayerNorm parameters: requested with the weight units behind each sweep - nothing is kept in registers across one)
  EFreq fq;
  fq.init(a.fxy, a.fyaw, s8);
  float4 qv[NH];
  ESlice qt[NH];
  float qb[NH];
  Pre pre;
  // ---------------------------------------------------------------- self attention
  {
    const float* qrow = a.qkv + (int64_t)row * a.ld_qkv;
    float4 bk[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
      bk[h] = *(const float4*)(a.bias_k1 + h * DH + s8 * 4);
      qt[h].load(qrow + a.qt_off + h * DR, s8);
    }
    pre_rows<KV16, REL>(a.self, row, b, wir, s8, tg, j1, fq, pre);  // (behind the query-side requests: waiting for those does not wait for these)
    if (threadIdx.x < D) xs[threadIdx.x] = x_r, bk2_s[threadIdx.x] = bk2_r;
#pragma unroll
    for (int h = 0; h < NH; ++h) qb[h] = tbx::group8_sum(dot4(qv[h], bk[h]));
  }
  bool valid1, valid2;
  MID_CLK(1);
  {
    RowAcc st;
    st.zero();
    float M[NH] = {0.f, 0.f, 0.f, 0.f};
    sweep_pf<KV16, REL>(a.self, row, b, wir, s8, tg, qv, qt, qb, fq, pre, ok1, st);
    float Mh, Lh;
    float4 om;
    ESlice em;
    merge_slots_by_head(st, M, Mh, Lh, om, em);
    MID_CLK(2);
    combine(Mh, Lh, om, em, red_s, comb_s, Pc, wir, lane, s8, valid1, [&]() {
      issue<0>(wb[0], a, heads, wave, lane);
      issue<1>(wb[1], a, heads, wave, lane);
      if (wave == 0) lg1[0] = a.ln_w[lane], lg1[1] = a.ln_w[64 + lane], lb1[0] = a.ln_b[lane], lb1[1] = a.ln_b[64 + lane];
      pre_rows<KV16, REL>(a.cross, row, b, wir, s8, tg, j2, fq, pre);  // the cross sweep's first pass: in flight under the layer's middle
    });
  }
  // ---- 0: y = sum a v + W_rpe_v (sum a e) + b (wave w: head w / 2, 16 of its 32 channels)
  issue<2>(wb[2], a, heads, wave, lane);
  {
    const f32x4 y = gemv4(wb[0], Pc, LO640, 4 + 4 * (wave >> 1), g4) + wb[0].bias + *(const f32x4*)(comb_s + c_out);
    if (col0) put4(Ph, LO128, c_out, y);
  }
  __syncthreads();
  MID_CLK(3);
  // ---- 1: x += no valid target ? 0 : out_proj(y)
  issue<3>(wb[0], a, heads, wave, lane);
  {
    const f32x4 u = gemv4(wb[1], Ph, LO128, 0, g4) + wb[1].bias;
    if (col0 && valid1) *(f32x4*)(xs + c_out) = *(const f32x4*)(xs + c_out) + u;
  }
  __syncthreads();
  MID_CLK(4);
  if (wave == 0) ln_planes(xs, Ph, lane, a.ln_eps, lg1, lb1);  // LN_1(x)
  __syncthreads();
  MID_CLK(5);
  // ---- 2: q = W_q LN(x) + b_q
  {
    const f32x4 q = gemv4(wb[2], Ph, LO128, 0, g4) + wb[2].bias;
    if (col0) {
      *(f32x4*)(q2 + c_out) = q;
      put4(Pq, LO128, c_out, q);
    }
  }
  __syncthreads();
  MID_CLK(6);
  // ---- 3: qt_h = W_rpe_k,h^T q_h: wave w = head w / 2, 4 of its 8 tiles of 16 channels, K = 32 (the head's own step)
  {
    const int h = wave >> 1;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      Acc acc;
      acc.zero();
      step(acc, wb[0].hi[st], wb[0].lo[st], Pq, LO128, h, g4);
      if (col0) *(f32x4*)(qt2 + h * D + ((wave & 1) * 4 + st) * 16 + 4 * g4) = acc.sum();
    }
  }
  __syncthreads();
  MID_CLK(7);
  // ---------------------------------------------------------------- cross attention
  {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      qv[h] = *(const float4*)(q2 + h * DH + s8 * 4);
      qb[h] = tbx::group8_sum(dot4(qv[h], *(const float4*)(bk2_s + h * DH + s8 * 4)));
      qt[h].load(qt2 + h * DR, s8);
    }
  }
  MID_CLK(8);
  {
    RowAcc st;
    st.zero();
    float M[NH] = {0.f, 0.f, 0.f, 0.f};
    sweep_pf<KV16, REL>(a.cross, row, b, wir, s8, tg, qv, qt, qb, fq, pre, ok2, st);
    float Mh, Lh;
    float4 om;
    ESlice em;
    merge_slots_by_head(st, M, Mh, Lh, om, em);
    MID_CLK(9);
    combine(Mh, Lh, om, em, red_s, comb_s, Pc, wir, lane, s8, valid2, [&]() {
      issue<4>(wb[1], a, heads, wave, lane);
      issue<5>(wb[2], a, heads, wave, lane);
      if (wave == 0) {
        lg2[0] = a.ln2_w[lane], lg2[1] = a.ln2_w[64 + lane], lb2[0] = a.ln2_b[lane], lb2[1] = a.ln2_b[64 + lane];
        if (a.qkv_out != nullptr) lg3[0] = a.ln3_w[lane], lg3[1] = a.ln3_w[64 + lane], lb3[0] = a.ln3_b[lane], lb3[1] = a.ln3_b[64 + lane];
      }
    });
  }
  const bool x_valid = a.src_invalid[row] == 0;
  // ---- 4: the cross attention's value fold
  issue<6>(wb[0], a, heads, wave, lane);
  {
    const f32x4 y = gemv4(wb[1], Pc, LO640, 4 + 4 * (wave >> 1), g4) + wb[1].bias + *(const f32x4*)(comb_s + c_out);
    if (col0) put4(Ph, LO128, c_out, y);
  }
  __syncthreads();
  // ---- 5: x += no valid cross target ? 0 : out_proj2(y)
  issue<7>(wb[1], a, heads, wave, lane);
  {
    const f32x4 u = gemv4(wb[2], Ph, LO128, 0, g4) + wb[2].bias;
    if (col0 && valid2) *(f32x4*)(xs + c_out) = *(const f32x4*)(xs + c_out) + u;
  }
  __syncthreads();
  MID_CLK(10);
  if (wave == 0) ln_planes(xs, Ph, lane, a.ln2_eps, lg2, lb2);  // h = norm2(x)
  __syncthreads();
  MID_CLK(11);
  // ---- 6..9: u = relu(linear1(h)), 4 rounds of 128 channels (no barrier between them: they read Ph and write disjoint parts of Pu)
#define TBX_MF_L1(N)                                                                 \
  do {                                                                               \
    issue<(N) + 2>(wb[((N) + 2) % 3], a, heads, wave, lane);                         \
    const W& w = wb[(N) % 3];                                                        \
    const f32x4 u = tbx_tile::relu4(gemv4(w, Ph, LO128, 0, g4) + w.bias);            \
    if (col0) put4(Pu, LO512, ((N) - 6) * D + c_out, u);                             \
  } while (0)
  TBX_MF_L1(6);
  TBX_MF_L1(7);
  TBX_MF_L1(8);
  TBX_MF_L1(9);
#undef TBX_MF_L1
  __syncthreads();
  MID_CLK(12);
  {  // ---- 10..13: x += linear2(u), K = 512 as 4 units into one accumulator triple; invalid source rows come out as 0
    Acc acc;
    acc.zero();
    const f32x4 bias = wb[10 % 3].bias;
#define TBX_MF_L2(N)                                                                 \
  do {                                                                               \
    issue<(N) + 2>(wb[((N) + 2) % 3], a, heads, wave, lane);                         \
    const W& w = wb[(N) % 3];                                                        \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) step(acc, w.hi[st], w.lo[st], Pu, LO512, 4 * ((N) - 10) + st, g4); \
  } while (0)
    TBX_MF_L2(10);
    TBX_MF_L2(11);
    TBX_MF_L2(12);
    TBX_MF_L2(13);
#undef TBX_MF_L2
    if (col0) {
      f32x4 v = *(const f32x4*)(xs + c_out) + (acc.sum() + bias);
      if (!x_valid) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(xs + c_out) = v;
      *(TBX_GLOBAL f32x4*)(a.x + (int64_t)row * D + c_out) = v;
    }
  }
  MID_CLK(13);
  if (a.qkv_out != nullptr) {
    // ================================================================ the next layer's projections (attention_rpe.py:92-98,147)
    __syncthreads();
    if (wave == 0) ln_planes(xs, Ph, lane, a.ln3_eps, lg3, lb3);
    __syncthreads();
    float* qrow_out = a.qkv_out + (int64_t)row * a.ld_qkv_out;
    {  // 14: q
      issue<16>(wb[16 % 3], a, heads, wave, lane);
      const W& w = wb[14 % 3];
      const f32x4 q = gemv4(w, Ph, LO128, 0, g4) + w.bias;
      if (col0) {
        put4(Pq, LO128, c_out, q);
        *(TBX_GLOBAL f32x4*)(qrow_out + c_out) = q;
      }
    }
#define TBX_MF_KV(N)                                                                                  \
  do {                                                                                                \
    const W& w = wb[(N) % 3];                                                                         \
    const f32x4 kv = gemv4(w, Ph, LO128, 0, g4) + w.bias;                                             \
    if (col0) {                                                                                       \
      *(TBX_GLOBAL f32x4*)(qrow_out + ((N) - 14) * D + c_out) = kv;                                   \
      if (a.kv16_out != nullptr) {                                                                    \
        const bf16x4 h16 = __builtin_convertvector(kv, bf16x4);                                       \
        *(TBX_GLOBAL u32x2*)(a.kv16_out + (int64_t)row * (2 * D) + ((N) - 15) * D + c_out) = __builtin_bit_cast(u32x2, h16); \
      }                                                                                               \
    }                                                                                                 \
  } while (0)
    issue<17>(wb[17 % 3], a, heads, wave, lane);
    TBX_MF_KV(15);
    TBX_MF_KV(16);
#undef TBX_MF_KV
    __syncthreads();  // q's planes complete
    MID_CLK(14);
    {  // 17: W_rpe_k^T q per head
      const W& w = wb[17 % 3];
      const int h = wave >> 1;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        Acc acc;
        acc.zero();
        step(acc, w.hi[st], w.lo[st], Pq, LO128, h, g4);
        if (col0) *(TBX_GLOBAL f32x4*)(qrow_out + 3 * D + h * D + ((wave & 1) * 4 + st) * 16 + 4 * g4) = acc.sum();
      }
    }
    MID_CLK(15);
    return;
  }
  if (a.tl.kv_out != nullptr) {
    // ================================================================ the lights' tail (traffic_bots.py:188-199): the K/V rows the agents'
    // 4 layers read (emit_kv_tables: LayerNorm_l + in_proj_kv,l) and the next-state logits (traffic_light.py:249-286)
    __shared__ __attribute__((aligned(16))) char Pl[4][2 * LO128];
    __syncthreads();  // xs complete
    if (wave < 4) {   // the four layers' LayerNorms at once, a wave each
      float g[2], bt[2];
      g[0] = a.tl.norm_weight[wave][lane], g[1] = a.tl.norm_weight[wave][64 + lane];
      bt[0] = a.tl.norm_bias[wave][lane], bt[1] = a.tl.norm_bias[wave][64 + lane];
      ln_planes(xs, Pl[wave], lane, a.tl.norm_eps[wave], g, bt);
    }
    if (col0) put4(Ph, LO128, c_out, *(const f32x4*)(xs + c_out));  // (the predictor reads x itself; this lane wrote these 4 channels)
    __syncthreads();
    const int64_t kv_row = (int64_t)row * a.tl.ld_kv;
#define TBX_MF_TLKV(N)                                                                                    \
  do {                                                                                                    \
    issue<(N) + 2>(wb[((N) + 2) % 3], a, heads, wave, lane);                                              \
    const W& w = wb[(N) % 3];                                                                             \
    const f32x4 y = gemv4(w, Pl[((N) - 14) >> 1], LO128, 0, g4) + w.bias;                                 \
    if (col0) {                                                                                           \
      const int64_t o = kv_row + (((N) - 14) >> 1) * 2 * D + (((N) - 14) & 1) * D + c_out;                \
      if (a.tl.kv_bf16) {                                                                                 \
        const bf16x4 h16 = __builtin_convertvector(y, bf16x4);                                            \
        *(TBX_GLOBAL u32x2*)((uint16_t*)a.tl.kv_out + o) = __builtin_bit_cast(u32x2, h16);                \
      } else {                                                                                            \
        *(TBX_GLOBAL f32x4*)((float*)a.tl.kv_out + o) = y;                                                \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
    TBX_MF_TLKV(14);
    TBX_MF_TLKV(15);
    TBX_MF_TLKV(16);
    TBX_MF_TLKV(17);
    TBX_MF_TLKV(18);
    TBX_MF_TLKV(19);
    TBX_MF_TLKV(20);
    TBX_MF_TLKV(21);
#undef TBX_MF_TLKV
    {  // 22, 23: the predictor's hidden layers (Ph -> Pq -> Ph)
      issue<24>(wb[24 % 3], a, heads, wave, lane);
      const W& w = wb[22 % 3];
      const f32x4 h1 = tbx_tile::relu4(gemv4(w, Ph, LO128, 0, g4) + w.bias);
      if (col0) put4(Pq, LO128, c_out, h1);
    }
    __syncthreads();
    {
      const W& w = wb[23 % 3];
      const f32x4 h2 = tbx_tile::relu4(gemv4(w, Pq, LO128, 0, g4) + w.bias);
      if (col0) put4(Ph, LO128, c_out, h2);
    }
    __syncthreads();
    if (wave == 0) {  // 24: n_state <= 16 logits (zero-padded tile 0): masked, clamped
      const W& w = wb[24 % 3];
      const f32x4 o = gemv4(w, Ph, LO128, 0, g4) + w.bias;
      const bool bad = a.tl.tl_invalid[row] != 0;
      if (col0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * g4 + r;
          if (c < a.tl.n_state) a.tl.logits_out[(int64_t)row * a.tl.n_state + c] = fminf(fmaxf(bad ? 0.f : o[r], a.tl.clamp_lo), a.tl.clamp_hi);
        }
      }
    }
    return;
  }
  if (!heads) return;
  // ================================================================ the agents' heads (traffic_bots.py:206-221; csrc/tile_heads.hip's
  // stages on one row): Pv = [x | navi_emb | latent_emb] planes, the adders' hidden rows in Ph / Pq, the action head's in Pu / Pv
  if (threadIdx.x < 64) {
    const int c = ((int)threadIdx.x & 31) * 4;
    const float* src = threadIdx.x < 32 ? a.navi_emb : a.latent_emb;
    put4(Pv, LO384, (threadIdx.x < 32 ? D : 2 * D) + c, *(const TBX_GLOBAL f32x4*)(src + (int64_t)row * D + c));
  }
  if (col0) put4(Pv, LO384, c_out, *(const f32x4*)(xs + c_out));  // (this lane's own 4 channels of x, written above)
  const bool ok_navi = a.navi_valid[row] != 0, ok_lat = a.latent_invalid[row] == 0;
  // the action head's branches this agent's type masks let through (action_head.py:64-100: one per agent type - an agent has one type,
  // an invalid agent none): only those are computed, each 3 weight units instead of all three branches' 7
  int act = 0;
#pragma unroll
  for (int g = 0; g < 3; ++g) act |= (a.type_mask[(int64_t)g * a.mask_stride + row] == 0 ? 1 : 0) << g;
  act = __builtin_amdgcn_readfirstlane(act);
  const int br0 = act ? __builtin_ctz((unsigned)act) : -1;
  __syncthreads();
#define TBX_MF_ADDER(N, ZSTEP, OK)                                                                    \
  do {                                                                                                \
    {                                                                                                 \
      Acc acc;                                                                                        \
      acc.zero();                                                                                     \
      issue<(N) + 2>(wb[((N) + 2) % 3], a, heads, wave, lane, br0);                                        \
      const W& w0 = wb[(N) % 3];                                                                      \
      const f32x4 bias = w0.bias;                                                                     \
      _Pragma("unroll") for (int st = 0; st < 4; ++st) step(acc, w0.hi[st], w0.lo[st], Pv, LO384, st, g4); \
      issue<(N) + 3>(wb[((N) + 3) % 3], a, heads, wave, lane, br0);                                        \
      const W& w1 = wb[((N) + 1) % 3];                                                                \
      _Pragma("unroll") for (int st = 0; st < 4; ++st) step(acc, w1.hi[st], w1.lo[st], Pv, LO384, (ZSTEP) + st, g4); \
      if (col0) put4(Ph, LO128, c_out, tbx_tile::relu4(acc.sum() + bias));                            \
    }                                                                                                 \
    __syncthreads();                                                                                  \
    {                                                                                                 \
      issue<(N) + 4>(wb[((N) + 4) % 3], a, heads, wave, lane, br0);                                        \
      const W& w = wb[((N) + 2) % 3];                                                                 \
      const f32x4 hdn = tbx_tile::relu4(gemv4(w, Ph, LO128, 0, g4) + w.bias);                         \
      if (col0) put4(Pq, LO128, c_out, hdn);                                                          \
    }                                                                                                 \
    __syncthreads();                                                                                  \
    {                                                                                                 \
      issue<(N) + 5>(wb[((N) + 5) % 3], a, heads, wave, lane, br0);                                        \
      const W& w = wb[((N) + 3) % 3];                                                                 \
      const f32x4 upd = tbx_tile::relu4(gemv4(w, Pq, LO128, 0, g4) + w.bias);                         \
      if (col0) {                                                                                     \
        f32x4 xv = *(const f32x4*)(xs + c_out);                                                       \
        if (OK) xv += upd;                                                                            \
        *(f32x4*)(xs + c_out) = xv;                                                                   \
        put4(Pv, LO384, c_out, xv);                                                                   \
      }                                                                                               \
    }                                                                                                 \
    __syncthreads();                                                                                  \
  } while (0)
  TBX_MF_ADDER(14, 4, ok_navi);
  TBX_MF_ADDER(18, 8, ok_lat);
#undef TBX_MF_ADDER
  // ---- action head, branch by branch in branch order (unit slots 22 / 23 / 24 mod 3 for its layers 1 / 2 / 3): 128 -> 128 relu (x
  // planes in Pv -> Pu[br * 128 ..]), 128 -> 128 relu (-> Ph), 128 -> 2 of 16 zero-padded outputs (wave 0)
  for (int br = br0; br >= 0;) {
    int nx = -1;  // the next branch let through (none for a one-hot type)
    for (int g = 2; g > br; --g)
      if ((act >> g) & 1) nx = g;
    {
      issue<24>(wb[24 % 3], a, heads, wave, lane, br);
      const W& w = wb[22 % 3];
      const f32x4 u = tbx_tile::relu4(gemv4(w, Pv, LO384, 0, g4) + w.bias);
      if (col0) put4(Pu, LO512, br * D + c_out, u);
    }
    __syncthreads();
    {
      issue<22>(wb[22 % 3], a, heads, wave, lane, nx);
      const W& w = wb[23 % 3];
      const f32x4 u = tbx_tile::relu4(gemv4(w, Pu, LO512, 4 * br, g4) + w.bias);
      if (col0) put4(Ph, LO128, c_out, u);
    }
    __syncthreads();
    issue<23>(wb[23 % 3], a, heads, wave, lane, nx);
    if (wave == 0) {
      const W& w = wb[24 % 3];
      const f32x4 o = gemv4(w, Ph, LO128, 0, g4) + w.bias;
      if (lane == 0) head_o[br * 2] = o[0], head_o[br * 2 + 1] = o[1];
    }
    __syncthreads();
    br = nx;
  }
  if (threadIdx.x < 2) {  // the masked sum over the branches, in branch order from 0
    float v = 0.f;
    for (int g = 0; g < 3; ++g)
      if ((act >> g) & 1) v += head_o[g * 2 + threadIdx.x];
    a.action_out[(int64_t)row * 2 + threadIdx.x] = v;
  }
  MID_CLK(14);
  if (a.fused_tail) {
    const int t_step = __builtin_amdgcn_readfirstlane(t_step_v);
    // ============================================================== the step's tail for this row's agent (csrc/step_core.h): its
    // tbx_sim_step (dynamics, rule checks, overrides, log, window append: 32 lanes) on the action just written, then the NEXT
    // step's tbx_agent_prep of its new window (4 waves) - neither reads anything of another agent's
    __syncthreads();  // the action is in memory (workgroup scope)
    if (wave == 0 && lane < tbx_step::LPA) tbx_step::sim_agent(a.sim, a.sim_parts, t_step, row, lane, 0);
    if (a.sim_parts & TBX_SIM_ADVANCE) tbx_step::sim_advance(a.sim, t_step, gridDim.x);
    __syncthreads();  // the appended window is
    MID_CLK(15);
    tbx_step::agent_prep(a.prep, row, (int)threadIdx.x, 512);
    __syncthreads();
    MID_CLK(0);  // (profiling build: the launch's end overwrites its first stamp - tools/mid_clock.py reads the tail from 13 -> 14 -> 15 -> 0)
  }
}

int check_seg(const tbx_attn_seg_t& s, const float* fxy, const float* fyaw) {
  if (!s.kv || !s.idx || !s.invalid || (!s.emb && !s.rel_pose) || s.k <= 0 || s.n_tgt <= 0 || s.batch_div <= 0) return TBX_ERR_ARG;
  if (!s.emb && (!fxy || !fyaw)) return TBX_ERR_ARG;
  if ((s.ld_kv % 4) || (s.k_off % 4) || (s.v_off % 4) || (((uintptr_t)s.kv) & 15) || (s.emb && (((uintptr_t)s.emb) & 15))) return TBX_ERR_ALIGN;
  return TBX_OK;
}

}  // namespace

static int dec_launch(const tbx_dec_mid_t* p, const tbx_dec_layer_t* t, void* stream);
extern "C" int tbx_knarpe_dec_mid(const tbx_dec_mid_t* p, void* stream) { return dec_launch(p, nullptr, stream); }
extern "C" int tbx_knarpe_dec_layer(const tbx_dec_layer_t* t, void* stream) {
  if (!t) return TBX_ERR_ARG;
  if (!t->out_proj2_image || !t->linear1_image || !t->linear2_image || !t->norm2_weight || !t->norm2_bias || !t->src_invalid) return TBX_ERR_ARG;
  if (t->qkv_out && (!t->next_in_proj_image || !t->next_qfold_image || !t->next_norm_weight || !t->next_norm_bias || t->ld_qkv_out < 7 * D ||
                     (t->ld_qkv_out % 4)))
    return TBX_ERR_ARG;
  if ((t->mid.self_seg.kv_bf16 != 0) != (t->kv16_out != nullptr) && t->qkv_out) return TBX_ERR_ARG;  // bf16 tables <=> a bf16 k | v copy for the next layer
  const void* al[] = {t->out_proj2_image, t->linear1_image, t->linear2_image, t->next_in_proj_image, t->next_qfold_image, t->qkv_out};
  for (const void* q : al)
    if (((uintptr_t)q) & 15) return TBX_ERR_ALIGN;
  return dec_launch(&t->mid, t, stream);
}
static int dec_launch(const tbx_dec_mid_t* p, const tbx_dec_layer_t* t, void* stream) {
  if (!p || !p->qkv || !p->x || (!t && (!p->out2 || !p->flag2)) || !p->rpe_k_bias_self || !p->rpe_k_bias_cross || !p->ln_weight || !p->ln_bias ||
      !p->fold_self_image || !p->out_proj_image || !p->q_image || !p->qfold_image || !p->fold_cross_image)
    return TBX_ERR_ARG;
  if (p->n_batch <= 0 || p->n_src <= 0 || p->n_cross < 1 || p->n_cross > 2) return TBX_ERR_ARG;
  if ((p->ld_qkv % 4) || (p->q_off % 4) || (p->qt_off % 4) || (!t && ((p->ld_out2 % 4) || p->ld_out2 < D))) return TBX_ERR_ALIGN;
  const void* al[] = {p->qkv, p->x, p->out2, p->fold_self_image, p->out_proj_image, p->q_image, p->qfold_image, p->fold_cross_image,
                      p->rpe_k_bias_self, p->rpe_k_bias_cross};
  for (const void* q : al)
    if (((uintptr_t)q) & 15) return TBX_ERR_ALIGN;
  MidArgs a;
  int rc = check_seg(p->self_seg, p->freqs_xy, p->freqs_yaw);
  if (rc != TBX_OK) return rc;
  int ktot = 0;
  for (int i = 0; i < p->n_cross; ++i) {
    rc = check_seg(p->cross_seg[i], p->freqs_xy, p->freqs_yaw);
    if (rc != TBX_OK) return rc;
    if ((p->cross_seg[i].kv_bf16 != 0) != (p->self_seg.kv_bf16 != 0)) return TBX_ERR_UNSUPPORTED;  // one table element type per call
    ktot += p->cross_seg[i].k;
  }
  if (ktot > KMAX || p->self_seg.k > KMAX) return TBX_ERR_UNSUPPORTED;
  a.qkv = p->qkv, a.x = p->x;
  a.self.seg[0] = a.self.seg[1] = p->self_seg;
  a.self.n_seg = 1;
  a.cross.seg[0] = p->cross_seg[0];
  a.cross.seg[1] = p->cross_seg[p->n_cross > 1 ? 1 : 0];
  a.cross.n_seg = p->n_cross;
  a.self.scale2 = a.cross.scale2 = 1.4426950408889634f / sqrtf((float)DH);
  a.bias_k1 = p->rpe_k_bias_self, a.bias_k2 = p->rpe_k_bias_cross, a.fxy = p->freqs_xy, a.fyaw = p->freqs_yaw;
  a.fold1 = p->fold_self_image, a.wo = p->out_proj_image, a.wq = p->q_image, a.wkf = p->qfold_image, a.fold2 = p->fold_cross_image;
  a.ln_w = p->ln_weight, a.ln_b = p->ln_bias, a.ln_eps = p->ln_eps;
  a.out2 = p->out2, a.flag2 = p->flag2;
  a.ld_qkv = p->ld_qkv, a.q_off = p->q_off, a.qt_off = p->qt_off, a.ld_out2 = p->ld_out2;
  a.n_rows = p->n_batch * p->n_src, a.n_src = p->n_src;
  a.wo2 = nullptr, a.w1 = a.w2 = a.wqkv = a.wqt = nullptr, a.ln2_w = a.ln2_b = a.ln3_w = a.ln3_b = nullptr;
  a.src_invalid = nullptr, a.qkv_out = nullptr, a.kv16_out = nullptr, a.ln2_eps = a.ln3_eps = 0.f, a.ld_qkv_out = 0;
  for (int i = 0; i < 9; ++i) a.hw[i] = nullptr;
  a.navi_emb = a.latent_emb = nullptr, a.navi_valid = a.latent_invalid = a.type_mask = nullptr, a.action_out = nullptr, a.mask_stride = 0;
  a.fused_tail = 0, a.sim_parts = 0;
  memset(&a.sim, 0, sizeof(a.sim));
  memset(&a.prep, 0, sizeof(a.prep));
  memset(&a.tl, 0, sizeof(a.tl));
  if (t) {
    a.wo2 = t->out_proj2_image, a.w1 = t->linear1_image, a.w2 = t->linear2_image, a.wqkv = t->next_in_proj_image, a.wqt = t->next_qfold_image;
    a.ln2_w = t->norm2_weight, a.ln2_b = t->norm2_bias, a.ln3_w = t->next_norm_weight, a.ln3_b = t->next_norm_bias;
    a.kv16_out = (uint16_t*)t->kv16_out;
    if (t->heads != nullptr) {
      const tbx_heads_tail_t& h = *t->heads;
      if (t->qkv_out != nullptr) return TBX_ERR_ARG;
      for (int i = 0; i < 9; ++i) {
        if (!h.images[i] || (((uintptr_t)h.images[i]) & 15)) return TBX_ERR_ARG;
        a.hw[i] = h.images[i];
      }
      if (!h.navi_emb || !h.latent_emb || !h.navi_valid || !h.latent_invalid || !h.type_mask || !h.action_out || h.mask_stride < a.n_rows)
        return TBX_ERR_ARG;
      a.navi_emb = h.navi_emb, a.latent_emb = h.latent_emb, a.navi_valid = h.navi_valid, a.latent_invalid = h.latent_invalid;
      a.type_mask = h.type_mask, a.action_out = h.action_out, a.mask_stride = h.mask_stride;
      if ((h.sim_state != nullptr) != (h.next_prep != nullptr)) return TBX_ERR_ARG;
      if (h.sim_state != nullptr) {
        if (!t->tail_mfma32) return TBX_ERR_UNSUPPORTED;
        a.sim = *h.sim_state, a.prep = *h.next_prep, a.sim_parts = h.sim_parts, a.fused_tail = 1;
        if ((a.sim_parts & ~TBX_SIM_ADVANCE) != TBX_SIM_AGENTS) return TBX_ERR_ARG;
        if (a.sim.n_batch * a.sim.n_ag != a.n_rows || a.prep.n_tok != a.n_rows || a.sim.action_mean != h.action_out) return TBX_ERR_ARG;
        if (!a.sim.step || !a.prep.hist_valid || !a.prep.tok_pose || !a.prep.attr || !a.prep.pe || !a.prep.row_invalid) return TBX_ERR_ARG;
      }
    }
    a.src_invalid = t->src_invalid, a.qkv_out = t->qkv_out, a.ln2_eps = t->norm2_eps, a.ln3_eps = t->next_norm_eps, a.ld_qkv_out = t->ld_qkv_out;
    if (t->lights != nullptr) {
      const tbx_tl_tail_t& L = *t->lights;
      if (!t->tail_mfma32 || t->qkv_out != nullptr || t->heads != nullptr) return TBX_ERR_ARG;
      for (int i = 0; i < 4; ++i)
        if (!L.kv_images[i] || !L.norm_weight[i] || !L.norm_bias[i] || (((uintptr_t)L.kv_images[i]) & 15)) return TBX_ERR_ARG;
      for (int i = 0; i < 3; ++i)
        if (!L.mlp_images[i] || (((uintptr_t)L.mlp_images[i]) & 15)) return TBX_ERR_ARG;
      if (!L.kv_out || !L.tl_invalid || !L.logits_out || L.ld_kv < 8 * D || (L.ld_kv % 4) || L.n_state <= 0 || L.n_state > 16) return TBX_ERR_ARG;
      if (((uintptr_t)L.kv_out) & 15) return TBX_ERR_ALIGN;
      a.tl = L;
    }
  }
  const size_t lds_bytes = (size_t)(IMG128 + IMGKF + 4 * RED + OUTW + 8 * D) * sizeof(float);
  static_assert((IMG128 + IMGKF + 4 * RED + OUTW + 8 * D) * sizeof(float) <= 160 * 1024, "LDS budget");
  hipStream_t hs = (hipStream_t)stream;
#define TBX_MID_LAUNCH(KV, NWV)                                                                                                    \
  do {                                                                                                                            \
    (void)hipFuncSetAttribute((const void*)dec_mid_kernel<KV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
    hipLaunchKernelGGL((dec_mid_kernel<KV, NWV>), dim3(a.n_rows), dim3(NWV * 64), lds_bytes, hs, a);                              \
  } while (0)
  a.tail_mfma = t ? t->tail_mfma32 : 0;
  bool rel = p->self_seg.emb == nullptr;
  for (int i = 0; i < p->n_cross; ++i) rel = rel && p->cross_seg[i].emb == nullptr;
  if (t && a.tail_mfma && p->self_seg.kv_bf16 != 0) {
    if (rel) hipLaunchKernelGGL((dec_layer_mf_kernel<true, true>), dim3(a.n_rows), dim3(512), 0, hs, a);
    else hipLaunchKernelGGL((dec_layer_mf_kernel<true, false>), dim3(a.n_rows), dim3(512), 0, hs, a);
  } else if (t && a.tail_mfma) {
    if (rel) hipLaunchKernelGGL((dec_layer_mf_kernel<false, true>), dim3(a.n_rows), dim3(512), 0, hs, a);
    else hipLaunchKernelGGL((dec_layer_mf_kernel<false, false>), dim3(a.n_rows), dim3(512), 0, hs, a);
  }
  else if (t && p->self_seg.kv_bf16 != 0)
    TBX_MID_LAUNCH(true, 8);
  else if (t)
    TBX_MID_LAUNCH(false, 8);
  else if (p->self_seg.kv_bf16 != 0)
    TBX_MID_LAUNCH(true, 4);
  else
    TBX_MID_LAUNCH(false, 4);
#undef TBX_MID_LAUNCH
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

#ifdef TBX_STAGE_CLOCK
extern "C" int tbx_debug_mid_dump(unsigned long long* host_out, int max_launches) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_mid_launch), sizeof(n)) != hipSuccess) return -1;
  const int m = (int)n < max_launches ? (int)n : max_launches;
  if (m > 0 && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mid_clk), (size_t)m * 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mid_launch), &z, sizeof(z));
  return m;
}
#endif
