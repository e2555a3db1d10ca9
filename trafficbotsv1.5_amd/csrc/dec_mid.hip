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
  // ... with the row's light stepped behind its logits (tbx_tl_tail_t.sim_state, round 6: the paired launches of a closed-loop step):
  // `sim` / `sim_parts` above hold the LIGHTS' state then (a launch's rows are agents or lights), tl_prep the tbx_tl_prep rows of its new window
  int tl_sim;
  tbx_step::TlPrepArgs tl_prep;
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
#define TBX_MF_NS mf
#define TBX_MF_KERNEL dec_layer_mf_kernel
#define TBX_MF_BODY dec_layer_mf_body
#define TBX_MF_PAIR_KERNEL dec_layer_mf_pair_kernel
#define TBX_MF_PAIR_ARGS MidPair
#define TBX_MF_SINGLE 0
#include "dec_layer_mf.inc"
#undef TBX_MF_NS
#undef TBX_MF_KERNEL
#undef TBX_MF_BODY
#undef TBX_MF_PAIR_KERNEL
#undef TBX_MF_PAIR_ARGS
#undef TBX_MF_SINGLE
#define TBX_MF_NS mf1
#define TBX_MF_KERNEL dec_layer_mf1_kernel
#define TBX_MF_BODY dec_layer_mf1_body
#define TBX_MF_PAIR_KERNEL dec_layer_mf1_pair_kernel
#define TBX_MF_PAIR_ARGS MidPair1
#define TBX_MF_SINGLE 1
#include "dec_layer_mf.inc"
#undef TBX_MF_NS
#undef TBX_MF_KERNEL
#undef TBX_MF_BODY
#undef TBX_MF_PAIR_KERNEL
#undef TBX_MF_PAIR_ARGS
#undef TBX_MF_SINGLE

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
// the launch descriptor of a tbx_knarpe_dec_mid / tbx_knarpe_dec_layer call (validated); rel = every segment as relative poses
static int dec_fill(const tbx_dec_mid_t* p, const tbx_dec_layer_t* t, MidArgs& a, bool& rel) {
  if (!p || !p->qkv || !p->x || (!t && (!p->out2 || !p->flag2)) || !p->rpe_k_bias_self || !p->rpe_k_bias_cross || !p->ln_weight || !p->ln_bias ||
      !p->fold_self_image || !p->out_proj_image || !p->q_image || !p->qfold_image || !p->fold_cross_image)
    return TBX_ERR_ARG;
  if (p->n_batch <= 0 || p->n_src <= 0 || p->n_cross < 1 || p->n_cross > 2) return TBX_ERR_ARG;
  if ((p->ld_qkv % 4) || (p->q_off % 4) || (p->qt_off % 4) || (!t && ((p->ld_out2 % 4) || p->ld_out2 < D))) return TBX_ERR_ALIGN;
  const void* al[] = {p->qkv, p->x, p->out2, p->fold_self_image, p->out_proj_image, p->q_image, p->qfold_image, p->fold_cross_image,
                      p->rpe_k_bias_self, p->rpe_k_bias_cross};
  for (const void* q : al)
    if (((uintptr_t)q) & 15) return TBX_ERR_ALIGN;
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
  a.tl_sim = 0;
  memset(&a.tl_prep, 0, sizeof(a.tl_prep));
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
      if (L.sim_state != nullptr) {  // the light's own step behind its logits
        a.sim = *L.sim_state, a.sim_parts = L.sim_parts, a.tl_sim = 1;
        if ((a.sim_parts & ~(TBX_SIM_ADVANCE | TBX_SIM_NO_APPEND)) != TBX_SIM_LIGHTS) return TBX_ERR_ARG;
        if (a.sim.n_batch * a.sim.n_tl != a.n_rows || a.sim.tl_logits != L.logits_out || L.n_state != 5) return TBX_ERR_ARG;
        if (!a.sim.step || !a.sim.tl_state || !a.sim.hist_tl || !a.sim.tl_gt || !a.sim.out_tl_state) return TBX_ERR_ARG;
        if (!L.prep_attr || !L.prep_row_invalid || L.prep_ld_attr < 16 || (L.prep_ld_attr % 4) || (((uintptr_t)L.prep_attr) & 15)) return TBX_ERR_ARG;
        a.tl_prep = tbx_step::TlPrepArgs{L.tl_invalid, L.prep_attr, L.prep_row_invalid, L.prep_ld_attr};
      }
    }
  }
  a.tail_mfma = t ? t->tail_mfma32 : 0;
  rel = p->self_seg.emb == nullptr;
  for (int i = 0; i < p->n_cross; ++i) rel = rel && p->cross_seg[i].emb == nullptr;
  return TBX_OK;
}

static int dec_launch(const tbx_dec_mid_t* p, const tbx_dec_layer_t* t, void* stream) {
  MidArgs a;
  bool rel = false;
  const int rc_fill = dec_fill(p, t, a, rel);
  if (rc_fill != TBX_OK) return rc_fill;
  const size_t lds_bytes = (size_t)(IMG128 + IMGKF + 4 * RED + OUTW + 8 * D) * sizeof(float);
  static_assert((IMG128 + IMGKF + 4 * RED + OUTW + 8 * D) * sizeof(float) <= 160 * 1024, "LDS budget");
  hipStream_t hs = (hipStream_t)stream;
#define TBX_MID_LAUNCH(KV, NWV)                                                                                                    \
  do {                                                                                                                            \
    (void)hipFuncSetAttribute((const void*)dec_mid_kernel<KV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
    hipLaunchKernelGGL((dec_mid_kernel<KV, NWV>), dim3(a.n_rows), dim3(NWV * 64), lds_bytes, hs, a);                              \
  } while (0)
  if (t && a.tail_mfma == 2 && p->self_seg.kv_bf16 != 0) {  // one bf16 product per LINEAR (bf16 tables only: the bf16-arithmetic schedule)
    if (rel) hipLaunchKernelGGL((dec_layer_mf1_kernel<true, true>), dim3(a.n_rows), dim3(512), 0, hs, a);
    else hipLaunchKernelGGL((dec_layer_mf1_kernel<true, false>), dim3(a.n_rows), dim3(512), 0, hs, a);
  } else if (t && a.tail_mfma == 2) {
    return TBX_ERR_UNSUPPORTED;
  } else if (t && a.tail_mfma && p->self_seg.kv_bf16 != 0) {
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

extern "C" int tbx_knarpe_dec_layer_pair(const tbx_dec_layer_t* ta, const tbx_dec_layer_t* tb, void* stream) {
  if (!ta || !tb) return TBX_ERR_ARG;
  for (const tbx_dec_layer_t* t : {ta, tb}) {  // (tbx_knarpe_dec_layer's own checks)
    if (!t->out_proj2_image || !t->linear1_image || !t->linear2_image || !t->norm2_weight || !t->norm2_bias || !t->src_invalid) return TBX_ERR_ARG;
    if (t->qkv_out && (!t->next_in_proj_image || !t->next_qfold_image || !t->next_norm_weight || !t->next_norm_bias || t->ld_qkv_out < 7 * D ||
                       (t->ld_qkv_out % 4)))
      return TBX_ERR_ARG;
    if ((t->mid.self_seg.kv_bf16 != 0) != (t->kv16_out != nullptr) && t->qkv_out) return TBX_ERR_ARG;
    const void* al[] = {t->out_proj2_image, t->linear1_image, t->linear2_image, t->next_in_proj_image, t->next_qfold_image, t->qkv_out};
    for (const void* q : al)
      if (((uintptr_t)q) & 15) return TBX_ERR_ALIGN;
    if (!t->tail_mfma32) return TBX_ERR_UNSUPPORTED;  // the paired form exists for the matrix-path layer only
  }
  MidPair p;
  static_assert(sizeof(MidPair) == sizeof(MidPair1), "one argument block for both product counts");
  bool rel_a = false, rel_b = false;
  int rc = dec_fill(&ta->mid, ta, p.m[0], rel_a);
  if (rc != TBX_OK) return rc;
  rc = dec_fill(&tb->mid, tb, p.m[1], rel_b);
  if (rc != TBX_OK) return rc;
  const bool kv16 = ta->mid.self_seg.kv_bf16 != 0;
  if (rel_a != rel_b || kv16 != (tb->mid.self_seg.kv_bf16 != 0) || ta->tail_mfma32 != tb->tail_mfma32) return TBX_ERR_UNSUPPORTED;
  if (ta->tail_mfma32 == 2 && !kv16) return TBX_ERR_UNSUPPORTED;
  // The halves on disjoint XCDs (workgroup b runs on XCD b % 8): an XCD's L2 then holds ONE half's weight units and K/V rows instead of
  // both - configs[1] (64 + 128 rows): 491.5 -> 499.4 k agent-steps/s with the agents on 3 XCDs (2: 492, 4: 497). Default: the share of
  // XCDs nearest to the halves' share of rows, if both halves then still have a CU per row (32 CUs per XCD); TBX_PAIR_XCD=0: off, n: forced.
  static const int pair_xcd = [] { const char* e = getenv("TBX_PAIR_XCD"); return (e && *e) ? atoi(e) : -1; }();
  const int na = p.m[0].n_rows, nb = p.m[1].n_rows;
  int xa = pair_xcd;
  if (xa < 0) {
    xa = (16 * na + (na + nb)) / (2 * (na + nb));  // round(8 na / (na + nb))
    xa = xa < 1 ? 1 : (xa > 7 ? 7 : xa);
    if ((na + xa - 1) / xa > 32 || (nb + 7 - xa) / (8 - xa) > 32) xa = 0;
  }
  p.xcd_a = 0;
  unsigned blocks = (unsigned)(na + nb);
  if (xa > 0 && xa < 8) {
    const int sa = (na + xa - 1) / xa, sb = (nb + 7 - xa) / (8 - xa);
    p.xcd_a = xa;
    blocks = 8u * (unsigned)(sa > sb ? sa : sb);
  }
  const dim3 grid(blocks);
  hipStream_t hs = (hipStream_t)stream;
  if (ta->tail_mfma32 == 2) {
    MidPair1 q;
    memcpy(&q, &p, sizeof(q));
    if (rel_a) hipLaunchKernelGGL((dec_layer_mf1_pair_kernel<true, true>), grid, dim3(512), 0, hs, q);
    else hipLaunchKernelGGL((dec_layer_mf1_pair_kernel<true, false>), grid, dim3(512), 0, hs, q);
  } else if (kv16) {
    if (rel_a) hipLaunchKernelGGL((dec_layer_mf_pair_kernel<true, true>), grid, dim3(512), 0, hs, p);
    else hipLaunchKernelGGL((dec_layer_mf_pair_kernel<true, false>), grid, dim3(512), 0, hs, p);
  } else {
    if (rel_a) hipLaunchKernelGGL((dec_layer_mf_pair_kernel<false, true>), grid, dim3(512), 0, hs, p);
    else hipLaunchKernelGGL((dec_layer_mf_pair_kernel<false, false>), grid, dim3(512), 0, hs, p);
  }
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
