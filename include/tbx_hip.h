/*
 * libtbx_hip.so — C ABI of the MI355X-native hot path of TrafficBots V1.5 (HPTR / KNARPE transformer +
 * closed-loop rollout). gfx950 only. Plain pointers and sizes, no torch types.
 *
 * Contract:
 *   - every pointer is a DEVICE pointer to a caller-owned, contiguous buffer unless marked "host";
 *   - nothing is allocated or retained, no global state, no host<->device synchronisation: every entry point only
 *     enqueues kernels on `stream` (a hipStream_t passed as void*), so a whole simulation step is hipGraph-capturable;
 *   - return value: 0 (TBX_OK) or a negative tbx error code; never throws, never exits.
 *
 * The reference (zhejz/TrafficBotsV1.5) has no native layer: each entry point replaces a chain of aten ops of the
 * Python reference, cited per function as `file:line` relative to the reference's src/.
 */
#ifndef TBX_HIP_H
#define TBX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBX_ABI_VERSION 4

enum {
  TBX_OK = 0,
  TBX_ERR_ARG = -1,          /* null pointer / non-positive size */
  TBX_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels are built for */
  TBX_ERR_ALIGN = -3,        /* pointer / leading dimension alignment */
  TBX_ERR_LAUNCH = -4        /* hipGetLastError() != hipSuccess after the launch */
};

int tbx_version(void);
const char* tbx_error_string(int code);

/* ------------------------------------------------------------------------------------------------------------------
 * K1+K2+K4: relative pose, K-nearest selection, pose embedding.
 * Replaces utils/rpe.py:8-37 (get_rel_pose), utils/rpe.py:61-90 (get_tgt_knn_idx: topk(largest=False) + mask +
 * gather) and utils/pose_emb.py:50-55 + utils/positional_emb.py:25,40-41,53 (PoseEmb pe_xy_yaw on the gathered
 * relative poses), as called from agent_encoder.py:321-387, traffic_light.py:118-152, map_encoder.py:82-97.
 *
 *   src_pose [n_batch, n_src, 3] (x,y,yaw)   src_invalid [n_batch, n_src] u8
 *   tgt_pose [n_batch/tgt_batch_div, n_tgt, 3], tgt_invalid likewise (rollouts of one scene may share targets)
 *   idx      [n_batch, n_src, k] i32   k smallest distances (set semantics, order unspecified; ties -> lower index)
 *   invalid  [n_batch, n_src, k] u8    tgt_invalid[idx] | dist > dist_limit   (+inf distances are always invalid)
 *   rel_pose [n_batch, n_src, k, 3]    optional (may be NULL): rotated offset and un-wrapped yaw difference
 *   emb      [n_batch, n_src, k, pe_dim] optional (may be NULL): [cos(x f) sin(x f) cos(y f) sin(y f) cos(k yaw) sin(k yaw)]
 *   freqs_xy [pe_dim/4], freqs_yaw [pe_dim/2]: the reference's repeat-interleaved `freqs` buffers.
 * Limits: 0 < k < n_tgt, k <= 64, n_tgt <= 2048, pe_dim in {64, 128}.
 */
int tbx_knn_embed(const float* src_pose, const uint8_t* src_invalid, const float* tgt_pose, const uint8_t* tgt_invalid,
                  int n_batch, int n_src, int n_tgt, int tgt_batch_div, int k, float dist_limit, int32_t* idx,
                  uint8_t* invalid, float* rel_pose, float* emb, const float* freqs_xy, const float* freqs_yaw,
                  int pe_dim, void* stream);

/* Up to 4 searches of tbx_knn_embed in ONE launch (agent_encoder.py:321-387 runs three per step - agent->agent, agent->map,
 * agent->light - from the same source poses; at a few hundred source rows each search is a latency-bound launch, and three
 * dependent launches on a side stream are three chances for the queue to stall). One workgroup per (job, source row), jobs back
 * to back over the grid in the order given; every job's outputs equal tbx_knn_embed's bit for bit (same device code). */
typedef struct tbx_knn_job {
  const float* src_pose;
  const uint8_t* src_invalid;
  const float* tgt_pose;
  const uint8_t* tgt_invalid;
  int32_t* idx;
  uint8_t* invalid;
  float* rel_pose; /* may be NULL */
  float* emb;      /* may be NULL */
  int32_t n_batch, n_src, n_tgt, tgt_batch_div, k;
  float dist_limit;
} tbx_knn_job_t;
int tbx_knn_embed_multi(const tbx_knn_job_t* jobs /* host */, int n_jobs, const float* freqs_xy, const float* freqs_yaw, int pe_dim,
                        void* stream);
/* ... with a tbx_pose_embed of explicit triples riding on the same launch (its blocks come after the searches'; the agents' destination
 * embedding of a simulation step, navigation.py:65-79, needs the same inputs as the searches and nothing they produce). pe may be NULL. */
typedef struct tbx_pose_embed_job {
  const float* pose3;      /* [n, 3] */
  const float* freqs_xy;
  const float* freqs_yaw;
  float* out;              /* out[i, col_off : col_off + pe_dim] */
  int64_t n;
  int32_t pe_dim, ld_out, col_off, reserved;
} tbx_pose_embed_job_t;
int tbx_knn_embed_multi_pe(const tbx_knn_job_t* jobs /* host */, int n_jobs, const float* freqs_xy, const float* freqs_yaw, int pe_dim,
                           const tbx_pose_embed_job_t* pe /* host, may be NULL */, void* stream);

/* Pose embedding of explicit (x,y,yaw) triples (utils/pose_emb.py:50-55); out[i, col_off : col_off+pe_dim]. */
int tbx_pose_embed(const float* pose3, int64_t n, const float* freqs_xy, const float* freqs_yaw, int pe_dim, float* out,
                   int ld_out, int col_off, void* stream);

/* utils/rpe.py:8-37 (get_rel_pose) / :41-58 (get_rel_dist) as DENSE tensors - the reference's stand-alone utility functions; the
 * hot path never materialises them (tbx_knn_embed ranks and gathers in registers). Same expressions as the search: a dense distance
 * equals the key tbx_knn_embed ranks by, bit for bit.
 *   rel_pose [n_batch, n_src, n_tgt, 3] (may be NULL)   rotated offset into the source frame, un-wrapped yaw difference
 *   rel_dist [n_batch, n_src, n_tgt]    (may be NULL)   |rotated offset|, +inf where source or target is invalid */
int tbx_rel_pose_dense(const float* src_pose, const uint8_t* src_invalid, const float* tgt_pose, const uint8_t* tgt_invalid,
                       int n_batch, int n_src, int n_tgt, int tgt_batch_div, float* rel_pose, float* rel_dist, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K6: fused KNARPE attention (gather + relative-pose bias + ragged masked softmax + weighted sum).
 * Replaces modules/attention_rpe.py:137-190 (rpe branch, apply_q_rpe = False) in the exact factorised form
 *   score[h,t] = q_h . k_h[idx_t] + qt_h . e_t + qb_h          (qt_h = W_rpe_k,h^T q_h ; qb_h = q_h . b_rpe_k,h)
 *   out       = [ sum_t a[h,t] v_h[idx_t]  |  sum_t a[h,t] e_t  (one 128-vector per head) ]
 * with K/V projected BEFORE the gather (per-token tables), mask -> -inf, rows without any valid target un-masked and
 * flagged in `row_no_valid` (the caller zeroes their out-projection, attention_rpe.py:112-118,188-190), softmax of
 * score / sqrt(d_head). d_model = 128, n_head = 4, d_rpe = 128.
 *
 *   qbuf [n_batch*n_src, ldq]: q at q_off (128), qt at qt_off (4*128); rpe_k_bias [128] = linear_rpe.bias[0:128]
 *   (qb_h is formed in-kernel from q and rpe_k_bias)
 *   seg[i]: kv [n_batch/batch_div, n_tgt, ld_kv] with K at k_off and V at v_off; idx / invalid [n_batch, n_src, k];
 *           the pair's pose embedding either materialised (emb [n_batch, n_src, k, 128]) or as its relative pose
 *           (rel_pose [n_batch, n_src, k, 3] + freqs_xy / freqs_yaw = the `pose_rpe` buffers): then the kernel moves exactly
 *           the algorithmic K row + V row + 12 B + 4 B + 1 B per pair
 *   out  [n_batch*n_src, ldo >= 640]; row_no_valid [n_batch*n_src] u8
 * Limits: 1 <= n_seg <= 2, sum of k <= 128, 16-byte aligned rows.
 */
typedef struct tbx_attn_seg {
  const float* kv;
  const int32_t* idx;
  const uint8_t* invalid;
  const float* emb;      /* [n_batch, n_src, k, 128] materialised pose embedding, or NULL */
  const float* rel_pose; /* [n_batch, n_src, k, 3] relative pose (used when emb == NULL: the embedding is rebuilt in registers) */
  int32_t ld_kv, k_off, v_off, n_tgt, batch_div, k;
  int32_t kv_bf16;       /* != 0: `kv` points to a bfloat16 table (ld_kv / k_off / v_off in elements): K and V rows move 2 B per
                            channel (529 B per pair instead of 1041), scores and sums accumulate in fp32. Forward only; every
                            segment of a call must use the same element type. */
} tbx_attn_seg_t;

int tbx_knarpe_attn_fwd(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                        int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, float* out, int ldo,
                        uint8_t* row_no_valid, const float* freqs_xy /* [32] or NULL */, const float* freqs_yaw /* [64] or NULL */,
                        void* stream);

/* The forward of LARGE launches (a wavefront per source row) on the bf16 matrix cores (csrc/attn_mfma.hip): scores and
 * weighted sums as v_mfma_f32_16x16x32_bf16 products per source row (heads padded 4 -> 16), the pair's embedding evaluated in the
 * operand layout, V / embedding rows transposed through LDS (ds_read_b64_tr_b16), online softmax in fp32. Same arguments, `out`
 * layout and row_no_valid as tbx_knarpe_attn_fwd (every segment in the relative-pose form: emb == NULL; the constant q . b_rpe_k
 * per head cancels in the softmax and is not formed); fp32 or bfloat16 K/V tables. Operands are rounded to bfloat16 (q, qt, K, V,
 * e, softmax weights), accumulation and softmax in fp32: the bf16-ARITHMETIC schedule BASELINE configs[1] names (the reference runs
 * at precision 16, configs/trainer/default.yaml:16); tolerance in tests/test_hip_attn_mfma.py. A persistent launch (<= 2 workgroups
 * per CU) whose waves prefetch the next chunk's / the next row's gathers under the current chunk's arithmetic. */
int tbx_knarpe_attn_fwd_mfma(const float* qbuf, int ldq, int q_off, int qt_off, int n_batch, int n_src,
                             const tbx_attn_seg_t* segs /* host */, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                             const float* freqs_xy /* [32] */, const float* freqs_yaw /* [64] */, void* stream);
/* ... with dropout on the attention probabilities (training, attention_rpe.py:171-172) and time-batched keys: the arguments and the
 * mask of tbx_knarpe_attn_fwd_dropout_tb (below) - (seed, call, scene row, step, target slot, head) -, so the fp32 backward entry points
 * regenerate exactly the mask this forward applied. The forward of training's attention under the bf16-autocast-class arithmetic. */
int tbx_knarpe_attn_fwd_mfma_dropout_tb(const float* qbuf, int ldq, int q_off, int qt_off, int n_batch, int n_src,
                                        const tbx_attn_seg_t* segs /* host */, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                                        const float* freqs_xy /* [32] */, const float* freqs_yaw /* [64] */, float p_drop,
                                        const uint64_t* drop_seed /* device */, uint32_t drop_call, int time_batch, int time0,
                                        void* stream);

/* The forward for launches of a few hundred rows (the closed loop at one or a few scenes: 4 wavefronts share a row), with the
 * value half of `linear_rpe` applied in the epilogue (attention_rpe.py:147,181-182: sum a (v + W_v e + b_v), softmax sums to 1):
 *   out [n_batch*n_src, ldo >= 128] = (sum a v)_h + W_rpe_v,h (sum a e)_h + b_rpe_v,h   (zero rows where row_no_valid)
 * fold_image = tbx_pack_weight_gemv(linear_rpe.weight[128:256], linear_rpe.bias[128:256], n = 32, k = 128, groups = 4): 66 KiB,
 * fetched into LDS by LDS-DMA while the targets are swept. 128 floats per row leave the kernel instead of 640, and the grouped
 * LINEAR stage that applied the fold in the following chain disappears; the result is bit-identical to that stage's
 * (same fma order). Inference only (no dropout). */
int tbx_knarpe_attn_fwd_folded(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch, int n_src,
                               const tbx_attn_seg_t* segs /* host */, int n_seg, float* out, int ldo, uint8_t* row_no_valid,
                               const float* freqs_xy, const float* freqs_yaw, const float* fold_image, void* stream);

/* The attention half of a dec_cross_attn layer (transformer_rpe.py:207-233) as ONE launch, for launches of a few hundred rows
 * (one workgroup per row): folded self attention -> x += no valid target ? 0 : out_proj(.) -> LayerNorm_1 -> q = W_q . + b_q ->
 * W_rpe_k^T q per head -> folded cross attention. Replaces [tbx_knarpe_attn_fwd_folded, a row chain, tbx_knarpe_attn_fwd_folded];
 * results are bit-identical to those three launches. All *_image arguments are tbx_pack_weight_gemv images:
 *   fold_self_image / fold_cross_image: linear_rpe.weight[128:256], bias[128:256] of the two attention modules (n 32, k 128, groups 4)
 *   out_proj_image: the self attention's out_proj (n 128, k 128);  q_image: the cross attention's in_proj rows [0, 128) (n 128, k 128)
 *   qfold_image: the cross attention's linear_rpe.weight[0:128] transposed use (n 128, k 32, groups 4, wt = 1), no bias
 * qkv [rows, ld_qkv]: q at q_off, W_k^T q at qt_off of the self attention; x [rows, 128] updated in place;
 * out2 [rows, ld_out2 >= 128] = folded cross-attention output, flag2 [rows] = no valid cross target. Inference only. */
typedef struct tbx_dec_mid {
  const float* qkv;
  float* x;
  tbx_attn_seg_t self_seg;
  tbx_attn_seg_t cross_seg[2];
  const float *rpe_k_bias_self, *rpe_k_bias_cross, *freqs_xy, *freqs_yaw;
  const float *fold_self_image, *out_proj_image, *q_image, *qfold_image, *fold_cross_image;
  const float *ln_weight, *ln_bias;
  float* out2;
  uint8_t* flag2;
  float ln_eps;
  int32_t ld_qkv, q_off, qt_off, ld_out2, n_cross, n_batch, n_src;
} tbx_dec_mid_t;
int tbx_knarpe_dec_mid(const tbx_dec_mid_t* args /* host */, void* stream);

/* A whole dec_cross_attn layer (transformer_rpe.py:207-245) in ONE launch for launches of a few hundred rows: tbx_knarpe_dec_mid
 * followed, in the same workgroup, by what the row chain after it did - x += row without a valid cross target ? 0 : out_proj(.),
 * x += linear2(relu(linear1(norm2(x)))), x[src_invalid] = 0 - and, unless qkv_out is NULL (last layer), the NEXT layer's
 * projections q | k | v = in_proj(norm_src'(x)), W_rpe_k^T q per head -> qkv_out [rows, ld_qkv_out >= 896]. mid.out2 / mid.flag2
 * are not written. Same arithmetic as the chain's stages (bit-identical). All images are tbx_pack_weight_gemv images:
 * out_proj2 (n 128, k 128), linear1 (n 512, k 128), linear2 (n 128, k 512), in_proj rows [0, 384) (n 384, k 128), query-side fold
 * (linear_rpe.weight[0:128] transposed: n 128, k 32, groups 4). */
/* Optional heads of the agents' policy in the LAST layer's launch (qkv_out == NULL), traffic_bots.py:206-221: x += navi_valid ?
 * add_navi.mlp([x | navi_emb]) : 0, x += latent valid ? add_latent.mlp([x | latent_emb]) : 0 (add_navi_latent.py:52-65; both
 * embeddings = mlp_in(.) with their invalid rows already zeroed), then the action head's per-type branches as three stacked /
 * block-diagonal stages and their masked sum (action_head.py:74-100) -> action_out [rows, 2]. images[0..2] = add_navi.mlp
 * (256 -> 128, 128 -> 128, 128 -> 128), [3..5] = add_latent.mlp, [6] = the branches' first layers stacked (128 -> 384), [7] = their
 * second layers (groups 3, 128 -> 128), [8] = their third layers zero-padded to 16 outputs (groups 3, 128 -> 16): gemv images. */
struct tbx_sim_state;       /* defined below (K10) */
struct tbx_agent_prep_args; /* defined below (tbx_agent_prep) */
typedef struct tbx_heads_tail {
  const float* images[9];
  const float *navi_emb, *latent_emb;    /* [rows, 128] */
  const uint8_t *navi_valid, *latent_invalid; /* [rows] */
  const uint8_t* type_mask;              /* [3, mask_stride]: byte set = the agent is not of that type (or not valid). The all-mfma32 layer
                                          * (tail_mfma32) computes only the branches a row's bytes let through - one for a one-hot type,
                                          * none for an invalid agent - and sums them in branch order, as the others do over all three */
  float* action_out;                     /* [rows, 2] */
  int32_t mask_stride;
  /* Fused step tail (tail_mfma32 launches; sim_state and next_prep both or neither; host pointers): behind a row's action the
   * launch runs that agent's tbx_sim_step_parts(sim_state, sim_parts) - sim_parts = TBX_SIM_AGENTS [| TBX_SIM_ADVANCE], the agents'
   * part only, rows = sim_state's n_batch * n_ag agents - and then tbx_agent_prep (*next_prep) of the NEXT step for that agent: an
   * agent's step and its window features read nothing of another agent's, so the two launches that close and open every step of the
   * closed loop are the tail of the launch that produced the action. */
  int32_t sim_parts;
  const struct tbx_sim_state* sim_state;
  const struct tbx_agent_prep_args* next_prep;
} tbx_heads_tail_t;

/* The traffic lights' tail behind their LAST layer (tail_mfma32 launches; traffic_bots.py:188-199): for each of the 4 layers of the
 * agents' block the K/V rows its tl cross-attention reads, k | v = in_proj_kv,l(norm_tgt,l(x)) -> kv_out[row, l * 256 ..] (fp32, or
 * bfloat16 with kv_bf16), and the next-state logits clamp(mlp(x) masked, -3, 3) (traffic_light.py:249-286) -> logits_out [rows, n_state]. */
typedef struct tbx_tl_tail {
  const float* kv_images[4];             /* tbx_pack_weight_mfma32 (n 256, k 128): in_proj rows [128, 384) of layer l */
  const float *norm_weight[4], *norm_bias[4];
  float norm_eps[4];
  void* kv_out;                          /* [rows, ld_kv] */
  const float* mlp_images[3];            /* n 128 k 128, n 128 k 128, n 16 k 128 (the n_state outputs zero-padded to 16) */
  const uint8_t* tl_invalid;             /* [rows] */
  float* logits_out;                     /* [rows, n_state] */
  int32_t ld_kv, kv_bf16, n_state, pad_;
  float clamp_lo, clamp_hi;
  /* sim_state != NULL (round 6; with it `prep_*`): behind its logits the row's workgroup also runs the NEXT step of its light -
   * tbx_sim_step_tl_prep(sim_state, sim_parts = TBX_SIM_LIGHTS [| TBX_SIM_ADVANCE], ...) for that light alone (dynamics.py:143-163,
   * traffic_bots.py:123-143, traffic_light.py:219-226; rows = sim_state's n_batch * n_tl lights, logits_out = its tl_logits): a light's
   * step reads nothing of another light's, so the launch that opened every step of the lights' recurrence is the tail of the launch
   * that produced its logits (the agents' counterpart: tbx_heads_tail_t.sim_state). */
  int32_t sim_parts, prep_ld_attr;
  const struct tbx_sim_state* sim_state;
  float* prep_attr;                      /* [rows * window, prep_ld_attr] */
  uint8_t* prep_row_invalid;             /* [rows * window] */
} tbx_tl_tail_t;

typedef struct tbx_dec_layer {
  tbx_dec_mid_t mid;
  const float *out_proj2_image, *linear1_image, *linear2_image, *next_in_proj_image, *next_qfold_image;
  const float *norm2_weight, *norm2_bias, *next_norm_weight, *next_norm_bias;
  const uint8_t* src_invalid; /* [rows] */
  float* qkv_out;             /* NULL: last layer */
  void* kv16_out;             /* bf16 K/V tables (mid.self_seg.kv_bf16): [rows, 256] bfloat16 copy of the next layer's k | v; else NULL */
  const tbx_heads_tail_t* heads; /* host pointer or NULL; only with qkv_out == NULL */
  float norm2_eps, next_norm_eps;
  int32_t ld_qkv_out;
  int32_t tail_mfma32; /* != 0: EVERY image of the call (mid.fold_self / out_proj / q / qfold / fold_cross, out_proj2 / linear1 / linear2 /
                        * next_in_proj / next_qfold, heads->images) is a tbx_pack_weight_mfma32 image and every LINEAR stage runs on
                        * the split-bf16 matrix path (< 3e-5 of sum |x||w| per output) instead of exact-fp32 fma chains.
                        * 2 (bf16 tables only): the same images, ONE bf16 product per LINEAR - weights and activations rounded to
                        * bfloat16 (2^-9 relative per operand), fp32 accumulation; the lo halves of the weight units are not read */
  const tbx_tl_tail_t* lights; /* host pointer or NULL; only with qkv_out == NULL, heads == NULL and tail_mfma32 */
} tbx_dec_layer_t;
int tbx_knarpe_dec_layer(const tbx_dec_layer_t* args /* host */, void* stream);
/* TWO independent row sets as ONE launch (round 6): layer l of the agents' block and layer l of the lights' block of a closed-loop step
 * (the lights run one step ahead of the agents and read nothing of theirs) - n_a + n_b one-row workgroups on one queue instead of two
 * launches on two queues joined at every step. Both must be tail_mfma32 calls of the same kind (tail_mfma32 value, table element type,
 * relative-pose segments); either may carry its tail (heads / lights). Same results as the two single launches, bit for bit. */
int tbx_knarpe_dec_layer_pair(const tbx_dec_layer_t* a /* host */, const tbx_dec_layer_t* b /* host */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * tbx_layer_tile: the row-local part of a transformer layer for LARGE launches (>= ~1000 rows; 16-row tiles, one launch) -
 * what a tbx_rowchain program does between two attention launches, as one straight-line kernel whose LINEAR stages run on the
 * split-bf16 matrix path (v_mfma_f32_16x16x32_bf16 on bf16 hi + lo halves of activations and weights, three products, fp32
 * accumulation: error < 3e-5 of sum |x||w| per output; the tbx_rowchain / tbx_knarpe_dec_layer paths stay exact fp32).
 * The three optional parts run in this order on the token rows x [n_rows, 128] (updated in place when store_x != 0):
 *   attn_out != NULL   x += row_no_valid ? 0 : out_proj(sum a v + W_rpe_v (sum a e) + b_rpe_v), attn_out [n_rows, ld_attn >= 640]
 *                      = tbx_knarpe_attn_fwd's output (attention_rpe.py:152,182-190; transformer_rpe.py:212-213,233)
 *   linear1_image      x += linear2(relu(linear1(norm2 x))); x[src_invalid] = 0 (transformer_rpe.py:234-237,242)
 *   proj_image         q [| k | v] = in_proj(proj_norm x), qt_h = W_rpe_k,h^T q_h of the next attention call (attention_rpe.py:92-98,
 *                      147) -> proj_out [n_rows, ld_proj]: q | qt (proj_n = 128, ld_proj >= 640) or q | k | v | qt (proj_n = 384,
 *                      ld_proj >= 896); with kv16_out the k | v columns go to that bfloat16 table [n_rows, 256] instead.
 * Weights are tbx_pack_weight_mfma32 images: fold (groups 4, n 32, k 128: linear_rpe's value half), out_proj (128 x 128), linear1
 * (512 x 128), linear2 (128 x 512), proj (proj_n x 128), qfold (linear_rpe.weight[0:128] transposed: groups 4, n 128, k 32, no bias). */
#define TBX_MFMA32_UNIT_FLOATS 2112 /* a wave's unit of a tbx_pack_weight_mfma32 image: 4 groups x (1 KiB hi + 1 KiB lo) + 4 x 16 bias floats */
typedef struct tbx_layer_tile {
  float* x;
  const float* attn_out;
  const uint8_t* row_no_valid;
  const float *fold_image, *out_proj_image;
  const float *norm2_weight, *norm2_bias, *linear1_image, *linear2_image;
  const uint8_t* src_invalid; /* or NULL */
  const float *proj_norm_weight, *proj_norm_bias, *proj_image, *qfold_image;
  float* proj_out;
  void* kv16_out;             /* bfloat16 [n_rows, 256] or NULL */
  float norm2_eps, proj_norm_eps;
  int32_t ld_attn, ld_proj, proj_n, store_x;
  int64_t n_rows;
  /* keyed dropouts of training's stepping pass (tbx_keyed_dropout's mask of (seed, site, step, row, column); drop_thresh = 0: none):
   * site[0] on out_proj(.) before it is added to x (transformer_rpe.py:56-60,212-213), site[1] on the FFN's hidden rows (width 512),
   * site[2] on linear2(.) before it is added (transformer_rpe.py:234-237). A site < 0 is skipped. */
  const uint64_t* drop_seed;
  uint32_t drop_thresh;
  float drop_scale;
  int32_t drop_site[3], drop_step;
  /* rider (rider_rows > 0; first-projection launches only: no attn_out, no linear1_image, proj_n = 384): an independent 4-stage
   * MLP over rider_rows OTHER rows in extra workgroups of the same launch - the heads' navigation embedding (navigation.py:65-79 +
   * add_navi_latent.py:43-50), which depends on nothing the layers produce: a launch of its own on another stream costs a
   * cross-queue signal and a wait (~11 us of idle queue on the stepping stream, measured) -
   *   y = rider_add + (W0 rider_in + b0);  h = relu(W1 y + b1);  h = relu(W2 h + b2);  h = relu(W3 h + b3);
   *   rider_out[row] = rider_valid[row] ? h : 0.   All [rider_rows, 128]; images: tbx_pack_weight_mfma32 (n 128, k 128). */
  const float *rider_in, *rider_add;
  /* rider_pose3 != NULL: rider_in is not read - the stage-0 input is the 128-d pose embedding (tbx_pose_embed's) of rider_pose3
   * [rider_rows, 3], built in the kernel from the frequency tables rider_freqs_xy / rider_freqs_yaw */
  const float *rider_pose3, *rider_freqs_xy, *rider_freqs_yaw;
  const float* rider_images[4];
  const uint8_t* rider_valid;
  float* rider_out;
  int64_t rider_rows;
} tbx_layer_tile_t;
int tbx_layer_tile(const tbx_layer_tile_t* args /* host */, void* stream);
/* The same launch with ONE bf16 product per LINEAR stage (the bf16-arithmetic schedule): weights and activations rounded to bfloat16
 * (2^-9 relative per operand), fp32 accumulation; the same images (their lo halves are not read). Inference only (no dropout sites). */
int tbx_layer_tile_bf16(const tbx_layer_tile_t* args /* host */, void* stream);
/* tbx_tl_tail_tile: the traffic lights' tail (tbx_tl_tail_t above: the 4 K/V tables of the agents' light cross-attention,
 * transformer_rpe.py:220-223 + attention_rpe.py:92-98, and the next-state logits, traffic_light.py:249-286; called at
 * traffic_bots.py:188-199) on the finished light tokens x [n_rows, 128] for LARGE launches, same arithmetic class as tbx_layer_tile
 * (_bf16: one bf16 product per LINEAR). Small launches run the same tail inside tbx_knarpe_dec_layer. */
int tbx_tl_tail_tile(const float* x, int64_t n_rows, const tbx_tl_tail_t* tail /* host */, void* stream);
int tbx_tl_tail_tile_bf16(const float* x, int64_t n_rows, const tbx_tl_tail_t* tail /* host */, void* stream);
/* tbx_heads_tile: the agents' heads (traffic_bots.py:206-221) for large launches, same arithmetic class as tbx_layer_tile:
 * x' = x + (navi_valid ? add_navi.mlp([x | navi_emb]) : 0); x'' = x' + (latent_invalid ? 0 : add_latent.mlp([x' | latent_emb]));
 * action_out [n_rows, 2] = sum over the branches g with type_mask[g, row] == 0 of branch_g(x'') (action_head.py:74-100). x is not
 * written. images (tbx_pack_weight_mfma32): [0..2] add_navi.mlp (n 128 k 256, n 128 k 128, n 128 k 128), [3..5] add_latent.mlp,
 * [6] the branches' first layers stacked (n 384 k 128), [7] their second layers (groups 3, n 128, k 128), [8] their third layers
 * zero-padded to 16 outputs (groups 3, n 16, k 128). Both embeddings = mlp_in(.) with their invalid rows already zeroed. */
typedef struct tbx_heads_tile {
  const float* x;
  const float *navi_emb, *latent_emb;          /* [n_rows, 128] */
  const uint8_t *navi_valid, *latent_invalid;  /* [n_rows] */
  const uint8_t* type_mask;                    /* [3, mask_stride] */
  const float* images[9];
  float* action_out;
  int32_t mask_stride;
  /* raw != 0 (training's stepping pass): the two embeddings are made in the launch, with the keyed dropouts of the adders' MLPs -
   *   navi_emb = navi_valid ? mlp_in(dest_feature + mlp_pe(navi_pe)) : 0,  latent_emb = latent_invalid ? 0 : mlp_in(latent_z)
   * (navigation.py:65-79, add_navi_latent.py:43-50; navi_emb / latent_emb are not read). raw_images: mlp_pe (n 128 k 128), the
   * navigation adder's mlp_in (3 x n 128 k 128), the latent adder's mlp_in (n 128 k 32 - the 16-wide weight zero-padded -, n 128 k 128
   * twice). Dropout (drop_thresh != 0): site ids of the 12 relu outputs in the row chain's order - navi mlp_in 0..2, navi mlp 0..2,
   * latent mlp_in 0..2, latent mlp 0..2 (-1: none) - masks of tbx_keyed_dropout(seed, site, step, row, column of 128). */
  int32_t raw;
  int64_t n_rows;
  const float *navi_pe, *dest_feature; /* [n_rows, 128] */
  const float* latent_z;               /* [n_rows, ld_z >= 16] */
  const float* raw_images[7];
  const uint64_t* drop_seed;
  uint32_t drop_thresh;
  float drop_scale;
  int32_t drop_site[12], drop_step, ld_z;
} tbx_heads_tile_t;
int tbx_heads_tile(const tbx_heads_tile_t* args /* host */, void* stream);
/* The same launch with ONE bf16 product per LINEAR stage (the bf16-arithmetic schedule): weights and activations rounded to bfloat16
 * (2^-9 relative per operand), fp32 accumulation; the same images (their lo halves are not read). Inference only (no dropout sites). */
int tbx_heads_tile_bf16(const tbx_heads_tile_t* args /* host */, void* stream);

/* tbx_window_tile: the temporal PointNet of the agents' windows for large launches (agent_encoder.py:130-159: input encoder in
 * "cat" mode; polyline_encoder.py:49-61; pooling.py:18-19,38): per row f = [mlp(attr) | pe], three layers of
 * { h = relu(W f + b); f = [h | max of h over the window's valid rows] (invalid rows 0) }, out[window] = max of f over its valid
 * rows (0 for a window without one). attr [n_groups * window, ld_attr >= 32] (the first 32 columns are read: the producer zero-pads),
 * pe [n_groups * window, 64], row_invalid u8 [n_groups * window], out [n_groups, 128], window <= 16.
 * in_images: the input MLP as (n 64, k 32 - the weight zero-padded to 32 columns), (n 64, k 64), (n 64, k 64); pn_images: the
 * three PointNet layers (n 64, k 128): tbx_pack_weight_mfma32 images. */
typedef struct tbx_window_tile {
  const float *attr, *pe;
  const uint8_t* row_invalid;
  const float* in_images[3];
  const float* pn_images[3];
  float* out;
  int32_t window, ld_attr;
  int64_t n_groups;
  int32_t attr_cols; /* columns of attr that are read (<= 32, % 4 == 0; the rest of the 32-wide first layer sees zeros) */
  int32_t d_mlp;     /* 64: "cat" mode (agents); 128: "add" mode (traffic lights: traffic_light.py:219-226 - the input MLP is
                      * 32 -> 128 -> 128 -> 128 and `pe` is ONE feature row per window [n_groups, 128] added to its output) */
  int32_t add_mode, pad_;
  /* keyed dropouts of training's stepping pass on the three PointNet layers' relu outputs (polyline_encoder.py:52-58 through
   * mlp.py:60-61; tbx_keyed_dropout's mask of (seed, site, step, row, column of 64)); drop_thresh = 0: none */
  const uint64_t* drop_seed;
  uint32_t drop_thresh;
  float drop_scale;
  int32_t drop_site[3], drop_step;
} tbx_window_tile_t;
int tbx_window_tile(const tbx_window_tile_t* args /* host */, void* stream);
/* The same launch with ONE bf16 product per LINEAR stage (the bf16-arithmetic schedule): weights and activations rounded to bfloat16
 * (2^-9 relative per operand), fp32 accumulation; the same images (their lo halves are not read). Inference only (no dropout sites). */
int tbx_window_tile_bf16(const tbx_window_tile_t* args /* host */, void* stream);
/* tbx_front: everything between the feature preparation and a block's first decoder layer as ONE launch (small launches: the
 * closed loop at a few scenes), instead of tbx_window_tile -> tbx_knn_embed_multi_pe -> tbx_layer_tile one after the other:
 *   win    the block's window PointNet (tbx_window_tile's arguments; no dropout);
 *   layer  the block's FIRST PROJECTION as a tbx_layer_tile_t: x = win.out, n_rows = win.n_groups, no attn_out / linear1_image,
 *          proj_n = 384 (q | k | v | W_k^T q into proj_out, + kv16_out), and its rider_* (or rider_rows = 0);
 *   jobs   n_jobs (0..4) K-nearest searches in the relative-pose form (emb = NULL) + the pose-embedding job `pe` (or NULL), as
 *          tbx_knn_embed_multi_pe takes them.
 * The parts run in separate workgroups of the one launch; results are those of the three launches (the projection of a window's
 * pooled row on the split-bf16 matrix path as tbx_layer_tile's). */
struct tbx_knn_job;
struct tbx_pose_embed_job;
typedef struct tbx_front {
  tbx_window_tile_t win;
  tbx_layer_tile_t layer;
  const struct tbx_knn_job* jobs; /* host array */
  const struct tbx_pose_embed_job* pe; /* host pointer or NULL */
  const float *freqs_xy, *freqs_yaw;  /* (only read by jobs with emb != NULL: unused here) */
  int32_t n_jobs, pe_dim;
} tbx_front_t;
int tbx_front(const tbx_front_t* args /* host */, void* stream);
/* The agents' tbx_front (d_mlp 64, "cat") and the lights' (d_mlp 128, "add") of one closed-loop step as ONE launch (round 6; with
 * tbx_knarpe_dec_layer_pair the step runs on one queue). Same results as the two launches, bit for bit. */
int tbx_front_pair(const tbx_front_t* agents /* host */, const tbx_front_t* lights /* host */, void* stream);

/* tbx_tall_linear: y [m, n] = x [m, k] W^T (+ b) (optionally relu) over very many rows - the forward / input-gradient products of
 * training's time-batched pass - on the split-bf16 matrix path (< 3e-5 of sum |x||w| per output): image = tbx_pack_weight_mfma32 of
 * W [n x k] (wt = 1 for a [k x n] weight: the input gradient dx = dy W), k and n multiples of 128 (<= 1024), ldx / ldy % 4 == 0,
 * 16-byte aligned. has_bias: add the image's bias. k or n = 64 mod 128 (the 64-wide PointNet layers, polyline_encoder.py:49-61): the
 * image is that of W zero-padded to the next multiples of 128; x rows hold k, y rows n valid columns (loads / stores masked). */
int tbx_tall_linear(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                    void* stream);
/* ... with ONE bf16 product per term (x and W rounded to bfloat16, fp32 accumulation; the lo halves of the image are not read):
 * F.linear under torch.autocast(bfloat16). Same image, same arguments. */
int tbx_tall_linear_bf16(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                         void* stream);
/* ... with a second output: the same rows rounded to bfloat16, y16 [m, ldy16] (the K/V table of attention_rpe.py:92-98 in the element type
 * tbx_knarpe_attn_fwd_mfma gathers fastest, written by the producing LINEAR instead of a conversion pass; the fp32 rows stay for the
 * backward). Both arithmetic classes. */
int tbx_tall_linear_dual(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                         uint16_t* y16, int ldy16, void* stream);
int tbx_tall_linear_dual_bf16(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, int relu, float* y, int ldy,
                              uint16_t* y16, int ldy16, void* stream);
/* y = dropout(relu(x W^T + b)) in the same launch: the hidden activation of a transformer layer's FFN (transformer_rpe.py:119-131:
 * linear1 -> relu -> dropout) / an MLP layer (mlp.py:56-61) over the time-batched rows; the mask is tbx_keyed_dropout's for the [m, n]
 * output (p_drop = 0: relu only), so tbx_relu_drop_bwd reads relu' and the mask off y exactly as behind tbx_relu_drop_fwd.
 * Bit-identical to tbx_tall_linear(relu = 1) followed by tbx_keyed_dropout. _bf16: one bf16 product per term. */
int tbx_tall_linear_relu_drop(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, float* y, int ldy,
                              float p_drop, const uint64_t* drop_seed, uint32_t site, int rows_per_scene, int time_batch, int time0,
                              void* stream);
int tbx_tall_linear_relu_drop_bf16(const float* x, int64_t m, int k, int ldx, const float* image, int n, int has_bias, float* y, int ldy,
                                   float p_drop, const uint64_t* drop_seed, uint32_t site, int rows_per_scene, int time_batch, int time0,
                                   void* stream);

/* Image for the tbx_*_tile kernels of W_g [n x k] (g < groups; stored [k x n] per group if wt), bias [groups * n] or NULL. k = 32, 64 or a multiple
 * of 128, n % 16 == 0. Size in floats (negative: error code). Layout: csrc/tile_layer.hip. */
int64_t tbx_pack_weight_mfma32_size(int n, int k, int groups);
int tbx_pack_weight_mfma32(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out, void* stream);
/* n_jobs images in ONE launch (a training step re-packs ~300 images of <= 640 x 128 weights after every optimizer step: the launches,
 * not the bytes, were the cost). jobs is a HOST array; every job as tbx_pack_weight_mfma32's arguments. */
typedef struct tbx_pack_job {
  const float* w;
  const float* bias; /* or NULL */
  float* out;
  int32_t n, k, ld, groups, wt, pad_;
} tbx_pack_job_t;
int tbx_pack_weight_mfma32_multi(const tbx_pack_job_t* jobs /* host */, int n_jobs, void* stream);

/* Backward of tbx_knarpe_attn_fwd (training; autograd of modules/attention_rpe.py:137-190 in the factorised form).
 *   dout   [n_batch*n_src, ldo >= 640] = d(sum a v) | d(sum a e per head)
 *   dqbuf  [n_batch*n_src, ldq]  : dq written at q_off, dqt at qt_off (other columns untouched)
 *   dkv[i] : gradient of seg[i].kv, same shape / leading dimension; dK, dV are ACCUMULATED with atomicAdd (zero it first)
 *   dbias_k [n_batch*n_src, 128]: per-row gradient of rpe_k_bias, overwritten (the parameter's gradient is the sum over
 *   rows: one shared 128-float accumulator serialises ~10^3-deep at the L2 atomic units). May be NULL (every backward entry point):
 *   the gradient is identically zero in exact arithmetic - q_h . bk_h shifts all scores of a row, which its softmax ignores - and what
 *   the accumulation returns is round-off; the training step passes NULL and hands the parameter a zero gradient.
 * Probabilities are recomputed from the forward inputs; the pose embeddings carry no gradient (utils/rpe.py:7). */
int tbx_knarpe_attn_bwd(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch, int n_src,
                        const tbx_attn_seg_t* segs /* host */, int n_seg, const float* dout, int ldo, float* dqbuf,
                        float* const* dkv /* host array of n_seg device pointers */, float* dbias_k,
                        const float* freqs_xy, const float* freqs_yaw, void* stream);

/* The same pair with attention-probability dropout (training; modules/attention_rpe.py:171-172: dropout on the softmax
 * output, scaled by 1 / (1 - p)). The mask bit of (row, target slot, head) is a counter-based hash of the 64-bit seed at
 * *drop_seed (DEVICE memory, read by the kernels) and of drop_call (distinguishes the calls of one training step), so
 * the backward regenerates the forward's mask from the same (seed, call) and a captured graph draws fresh masks when the
 * host rewrites the seed between replays. p_drop = 0 is exactly tbx_knarpe_attn_fwd / _bwd (drop_seed may be NULL). */
int tbx_knarpe_attn_fwd_dropout(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, float* out, int ldo,
                                uint8_t* row_no_valid, const float* freqs_xy, const float* freqs_yaw, float p_drop,
                                const uint64_t* drop_seed /* device */, uint32_t drop_call, void* stream);
int tbx_knarpe_attn_bwd_dropout(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, const float* dout, int ldo,
                                float* dqbuf, float* const* dkv /* host array of n_seg device pointers */, float* dbias_k,
                                const float* freqs_xy, const float* freqs_yaw, float p_drop,
                                const uint64_t* drop_seed /* device */, uint32_t drop_call, void* stream);

/* The backward without dK / dV atomics. tbx_knn_inverse turns a K-nearest set into per-table inverse lists (for every
 * target token the un-masked (source row, slot) pairs that selected it; pair id = row * k + slot; order unspecified):
 *   inv_ptr [n_batch / tgt_batch_div, n_tgt + 1] i32 (offsets into the table's list), inv_list [n_tables, n_src * div * k] i32.
 * tbx_knarpe_attn_bwd_gather = tbx_knarpe_attn_bwd_dropout, but the row kernel only stores 8 coefficients per pair into
 * `coef` [n_batch*n_src, sum k, 8] (scratch) and a second kernel sums every target token's dK / dV row through the lists
 * (the K and V columns of every token row of dkv are OVERWRITTEN - no pre-zeroing, no atomics: 23 M float atomics per launch
 * at 1024 rows x 89 pairs were 2/3 of the launch).
 * n_tgt <= 2048. */
int tbx_knn_inverse(const int32_t* idx, const uint8_t* invalid, int n_batch, int n_src, int k, int n_tgt, int tgt_batch_div,
                    int32_t* inv_ptr, int32_t* inv_list, void* stream);
int tbx_knarpe_attn_bwd_gather(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                               int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, const float* dout, int ldo,
                               float* dqbuf, float* const* dkv /* host array */, float* dbias_k, const float* freqs_xy,
                               const float* freqs_yaw, float p_drop, const uint64_t* drop_seed /* device */, uint32_t drop_call,
                               const int32_t* const* inv_ptr /* host array of n_seg device pointers */,
                               const int32_t* const* inv_list /* host array */, float* coef, void* stream);

/* Time-batched forms (training, train_graph.training_rollout_batched): the reference's training rollout detaches the policy
 * inputs of every closed-loop step (waymo_motion.py:206-311 with training=True: the only cross-step gradient path is the
 * dynamics chain), so once the states of the T steps are known the T policy evaluations of a scene are independent and are
 * evaluated - and differentiated - as T consecutive batch entries: batch entry b is step time0 + b % time_batch of scene
 * b / time_batch. The dropout mask is then keyed by (seed, call, scene row, step, slot, head): the batched call draws exactly
 * the masks of time_batch per-step calls made with (time_batch = 1, time0 = step). (1, 0) = the entry points above. */
int tbx_knarpe_attn_fwd_dropout_tb(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                   int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, float* out, int ldo,
                                   uint8_t* row_no_valid, const float* freqs_xy, const float* freqs_yaw, float p_drop,
                                   const uint64_t* drop_seed /* device */, uint32_t drop_call, int time_batch, int time0,
                                   void* stream);
int tbx_knarpe_attn_bwd_dropout_tb(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                   int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, const float* dout, int ldo,
                                   float* dqbuf, float* const* dkv /* host array */, float* dbias_k, const float* freqs_xy,
                                   const float* freqs_yaw, float p_drop, const uint64_t* drop_seed /* device */,
                                   uint32_t drop_call, int time_batch, int time0, void* stream);
int tbx_knarpe_attn_bwd_gather_tb(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                  int n_src, const tbx_attn_seg_t* segs /* host */, int n_seg, const float* dout, int ldo,
                                  float* dqbuf, float* const* dkv /* host array */, float* dbias_k, const float* freqs_xy,
                                  const float* freqs_yaw, float p_drop, const uint64_t* drop_seed /* device */,
                                  uint32_t drop_call, int time_batch, int time0, const int32_t* const* inv_ptr /* host array */,
                                  const int32_t* const* inv_list /* host array */, float* coef, void* stream);

/* Elementwise dropout with the same kind of key (replaces F.dropout at modules/mlp.py:60-61, transformer_rpe.py:56-60,
 * 93-131 in training): y[row, c] = x[row, c] * keep / (1 - p), keep = hash(seed, site, step, scene row, c) >= p * 2^32.
 * x, y [rows, cols] contiguous (y may alias x); rows_per_scene rows per batch entry; batch entry b = row / rows_per_scene is
 * step time0 + b % time_batch of scene b / time_batch (see above), scene row = (b / time_batch) * rows_per_scene + row %
 * rows_per_scene. Its own backward (the gradient takes the same mask). `site` distinguishes the dropout sites of a step. */
int tbx_keyed_dropout(const float* x, float* y, int64_t rows, int cols, int rows_per_scene, float p_drop,
                      const uint64_t* drop_seed /* device */, uint32_t site, int time_batch, int time0, void* stream);

/* The elementwise glue of a transformer layer over the time-batched rows, one pass per tensor (training; autograd of
 * transformer_rpe.py:93-131: masked_fill / dropout / add / masked_fill and relu / dropout, each an HBM pass of its own in aten).
 * The dropout is tbx_keyed_dropout's for the tensor's [rows, cols] view (p_drop = 0: none); cols % 4 == 0, 16-byte aligned.
 *   tbx_residual_drop_fwd: out = zero_out[row] ? 0 : x + dropout(zero_y[row] ? 0 : y)   (zero_y / zero_out: u8 per row, may be NULL)
 *   tbx_residual_drop_bwd: dy = (zero_out | zero_y)[row] ? 0 : dout * mask / (1 - p); dx (may be NULL: then dx = dout) = zero_out[row] ? 0 : dout
 *   tbx_relu_drop_fwd:     h = dropout(relu(z));   tbx_relu_drop_bwd: dz = h > 0 ? dh / (1 - p) : 0 */
int tbx_residual_drop_fwd(const float* x, const float* y, const uint8_t* zero_y, const uint8_t* zero_out, int64_t rows, int cols, float p_drop,
                          const uint64_t* drop_seed /* device */, uint32_t site, int rows_per_scene, int time_batch, int time0, float* out,
                          void* stream);
int tbx_residual_drop_bwd(const float* dout, const uint8_t* zero_y, const uint8_t* zero_out, int64_t rows, int cols, float p_drop,
                          const uint64_t* drop_seed /* device */, uint32_t site, int rows_per_scene, int time_batch, int time0, float* dy,
                          float* dx, void* stream);
int tbx_relu_drop_fwd(const float* z, int64_t rows, int cols, float p_drop, const uint64_t* drop_seed /* device */, uint32_t site,
                      int rows_per_scene, int time_batch, int time0, float* h, void* stream);
int tbx_relu_drop_bwd(const float* dh, const float* h, int64_t rows, int cols, float p_drop, float* dz, void* stream);
/* h [n_batch, n_a, n_m, cols] += pa [n_batch, n_a, cols] (per agent) + pm [n_batch, n_m, cols] (per polyline), then relu if `relu`, in
 * place and in one pass: the broadcast terms of NaviPredictor's first Linear over agent x polyline pairs (navigation.py:245-262 - the
 * reference concatenates [f_a | f_m | e] per pair and multiplies by W [128, 384]; here W_a f_a and W_m f_m + b are computed per agent /
 * per polyline and added onto W_e e). cols % 4 == 0, 16-byte aligned. */
int tbx_pair_bias_relu(float* h, const float* pa, const float* pm, int64_t n_batch, int n_a, int n_m, int cols, int relu, void* stream);

/* Weight gradient of a LINEAR over very many rows (training; autograd of F.linear at modules/mlp.py:69-72,
 * attention_rpe.py:95-120,190, transformer_rpe.py:119-131 in the time-batched pass): dw[n,k] = dy[rows,n]^T x[rows,k],
 * db[n] = sum_rows dy (db may be NULL). dy / x row-major with leading dimensions ld_dy / ld_x; n, k and both ld multiples of 4,
 * 16-byte aligned. Exact-fp32 MFMA; `scratch` holds splits x (n*k + n) floats (splits from tbx_linear_wgrad_splits, any value
 * >= 1 is legal), summed by a second kernel in a fixed order (deterministic). */
int tbx_linear_wgrad_splits(int64_t rows, int n, int k);
int tbx_linear_wgrad(const float* dy, int ld_dy, const float* x, int ld_x, int64_t rows, int n, int k, float* dw, float* db,
                     float* scratch, int splits, void* stream);
/* The same with ONE bf16 product per term: dy and x rounded to bfloat16 in registers, fp32 accumulation over the rows (db stays an
 * exact fp32 sum) - the weight gradient of torch.autocast(bfloat16); the reference trains at precision 16
 * (configs/trainer/default.yaml:16). Same arguments, scratch and determinism. */
int tbx_linear_wgrad_bf16(const float* dy, int ld_dy, const float* x, int ld_x, int64_t rows, int n, int k, float* dw, float* db,
                          float* scratch, int splits, void* stream);

/* The folded GEMM weights of ONE AttentionRPE module and their backward (training; DESIGN.md 3: the exact algebra that takes
 * linear_rpe's per-pair projection, attention_rpe.py:137-164, out of the pair loop; autograd of the slices / batched products /
 * concatenations that built them - ~35 launches per module and training step - as one launch each):
 *   w [384,128] / b [384] in_proj, wr [256,128] / br [256] linear_rpe (key half rows 0..127, value half 128..255), wo [128,128] / bo [128]
 *   out_proj  ->  w_in [640,128] = [W_q ; B_k^T W_q], b_in [640], w_kv [256,128] = in_proj rows 128.., b_kv [256], bias_k [128] = br[:128],
 *   w_out [128,640] = [W_o | W_o B_v^T], b_out [128] = W_o br[128:] + b_o    (B_k / B_v: per-head 32 x 128 blocks of wr's halves)
 * Backward: g_* = the gradients that arrived for the seven outputs (any may be NULL = zero); d_* overwritten. fp32, fixed order. */
int tbx_attn_fold_fwd(const float* w, const float* b, const float* wr, const float* br, const float* wo, const float* bo, float* w_in,
                      float* b_in, float* w_kv, float* b_kv, float* bias_k, float* w_out, float* b_out, void* stream);
int tbx_attn_fold_bwd(const float* w, const float* b, const float* wr, const float* br, const float* wo, const float* g_w_in,
                      const float* g_b_in, const float* g_w_kv, const float* g_b_kv, const float* g_bias_k, const float* g_w_out,
                      const float* g_b_out, float* d_w, float* d_b, float* d_wr, float* d_br, float* d_wo, float* d_bo, void* stream);

/* Forward and backward of a LayerNorm over rows of 128 (training; autograd of F.layer_norm at modules/transformer_rpe.py:207-245 - norm1 / norm2 /
 * norm_src / norm_tgt - over the time-batched rows): x, dy, dx [rows, 128] contiguous, gamma [128], mean / rstd [rows] as the forward
 * (torch.native_layer_norm) produced them; dgamma / dbeta [128]. x and dy are read once, dx written once; `scratch` holds
 * tbx_layernorm_bwd_partials(rows) x 256 floats, summed by a second kernel in a fixed order (deterministic). cols != 128:
 * TBX_ERR_UNSUPPORTED. */
int tbx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int cols, float* y, float* mean,
                      float* rstd, void* stream); /* y = (x - mean) * rstd * gamma + beta, rstd = 1 / sqrtf(var + eps) (biased variance,
                                                     two passes): the arithmetic of the row chains' TBX_OP_LAYERNORM stage */
int tbx_layernorm_bwd_partials(int64_t rows);
int tbx_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, int64_t rows, int cols,
                      float* dx, float* dgamma, float* dbeta, float* scratch, void* stream);
/* ... with the gradient of the residual branch that forks off x added in the same pass: dx = add + (the LayerNorm's input gradient); add
 * [rows, 128] or NULL (= tbx_layernorm_bwd). In x_{i+1} = x_i + f(LayerNorm(x_i)) (transformer_rpe.py:207-245) autograd sums the two
 * gradients of x_i with a kernel of its own: three more passes over [rows, 128] per LayerNorm. */
int tbx_layernorm_bwd_add(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, int64_t rows, int cols,
                          const float* add, float* dx, float* dgamma, float* dbeta, float* scratch, void* stream);

/* The glue of a PointNet layer over the time-batched windows, forward and backward (training; autograd of polyline_encoder.py:49-61 and
 * pooling.py:18-19,38: relu / dropout / masked_fill / amax / expand / cat and their backward kernels). A wavefront per group of
 * group_rows <= 32 rows (two register footprints: <= 16 - the windows of 11 - and <= 32 - the map's polylines of 20 nodes); every tensor is read or
 * written once.
 *   tbx_pointnet_tail_fwd: z [n_groups, group_rows, 64] = the layer's Linear output, invalid [n_groups, group_rows] u8 ->
 *     out [n_groups, group_rows, 128] = [h | max over the group's valid rows of h], invalid rows zeroed, h = dropout(relu(z)) with
 *     tbx_keyed_dropout's mask for the [n_groups * group_rows, 64] view (p_drop = 0: none; site / rows_per_scene / time_batch / time0
 *     as there).
 *   tbx_pointnet_tail_bwd: dz from dout and the forward's `out` (the maximum's gradient split evenly among tied rows, as aten's
 *     amax backward does; relu' and the dropout mask are read off h > 0).
 *   tbx_masked_maxpool_fwd / _bwd: y [n_groups, 128] = max over the valid rows of x [n_groups, group_rows, 128] (0 for a group without
 *     one); dx from dy and x, ties split evenly.
 * cols must be 64 (tail) / 128 (pool): anything else is TBX_ERR_UNSUPPORTED. */
int tbx_pointnet_tail_fwd(const float* z, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols, float p_drop,
                          const uint64_t* drop_seed /* device */, uint32_t site, int rows_per_scene, int time_batch, int time0, float* out,
                          void* stream);
int tbx_pointnet_tail_bwd(const float* dout, const float* out, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols,
                          float p_drop, float* dz, void* stream);
int tbx_masked_maxpool_fwd(const float* x, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols, float* y, void* stream);
int tbx_masked_maxpool_bwd(const float* dy, const float* x, const uint8_t* invalid, int64_t n_groups, int group_rows, int cols, float* dx,
                           void* stream);

/* The per-step state machine of the TRAINING rollout (waymo_motion.py:206-311 with training=True; utils/dynamics.py:66-204,
 * 237-274, utils/teacher_forcing.py:108-167, utils/traffic_rule_checker.py:109-120,300-330, utils/rewards.py:35-85,
 * utils/buffer.py:39-78) for steps t in [t0, t1) (step number s = t + 1 = ground-truth index): action = tanh(mean) * lim,
 * kinematic step, teacher-forcing override (tf_mask & ~disabled), outside-map / destination-reached flags on the prediction,
 * the differentiable reward -(w_pos SmoothL1(xy) + w_rot (1 - cos dyaw) / 2 + w_spd SmoothL1(speed)) on pred & gt valid, and
 * the disabling of agents that left the map. A thread per agent; the state persists in the struct's buffers between calls
 * (the stepping pass calls one step at a time, the differentiated pass all steps at once). All arrays are caller-owned
 * device memory; `mean` element (b, t, a, c) is mean[b * stride_n + (t - t0) * stride_t + a * 2 + c].
 *   rec_*: the state BEFORE every step, time-major with window - 1 leading slots the caller zero-fills: slot window - 1 + t
 *   holds the state before step t + 1 (slot window - 1 is written by the caller from the initial state), so the W-step history
 *   window of step t + 1 is slots [t, t + window). [n, T + window, A(,3)].
 * tbx_train_chain_bwd: d_mean [n,T,A,2] from d_reward [n,T,A] after a forward over [0, T) (reads rec_*, ov, reward_valid,
 * pred_*): the reverse walk with a 4-float adjoint (x, y, yaw, speed) per agent. */
typedef struct tbx_train_chain {
  int32_t n_batch, n_ag, n_step, n_step_gt, n_node, window;
  float dt, w_pos, w_rot, w_spd;
  const uint8_t* gt_valid;   /* [n,A,Tg] */
  const float* gt_pose;      /* [n,A,Tg,3] */
  const float* gt_motion;    /* [n,A,Tg,3] */
  const uint8_t* tf_mask;    /* [n,A,Tg] TeacherForcing.ag_teacher_forcing */
  const float* lim;          /* [n,A,2] max acceleration / yaw rate of the agent's type */
  const float* dest_pos;     /* [n,A,N,2] */
  const float* dest_dir;     /* [n,A,N,2] unit */
  const uint8_t* dest_invalid; /* [n,A,N] */
  const float* dest_thresh;  /* [n,A] */
  const uint8_t* dest_kind;  /* [n,A] bit 0: lane-like destination (position and heading), bit 1: type 4 (position only) */
  const float* boundary;     /* [n,4] x0 x1 y0 y1 */
  uint8_t *valid, *disabled, *navi_valid, *outside, *reached; /* state [n,A] */
  float *pose, *motion;      /* state [n,A,3] */
  uint8_t* rec_valid;        /* [n,T+W,A] */
  float* rec_pose;           /* [n,T+W,A,3] */
  float* rec_motion;         /* [n,T+W,A,3] */
  uint8_t* rec_navi_valid;   /* [n,T+W,A] */
  uint8_t *pred_valid, *tf, *ov, *reward_valid; /* [n,T,A]: tf = logged forcing mask, ov = override applied */
  float *pred_pose, *pred_motion; /* [n,T,A,3] */
  float* reward;             /* [n,T,A] */
} tbx_train_chain_t;
int tbx_train_chain_fwd(const tbx_train_chain_t* c /* host */, const float* mean, int64_t mean_stride_n, int64_t mean_stride_t, int t0,
                        int t1, void* stream);
int tbx_train_chain_bwd(const tbx_train_chain_t* c /* host */, const float* mean, int64_t mean_stride_n, int64_t mean_stride_t,
                        const float* d_reward, float* d_mean, void* stream);
/* tbx_train_chain_fwd over steps [t0, t1) (t0 == t1: no step, `mean` may be NULL) and then the policy inputs of step t1 + 1 in the
 * layout TrafficBots.agent_policy reads (agent_encoder.py:130-159's windows): hv [n,A,W] u8, hp / hm [n,A,W,3] oldest first, valid /
 * navi_valid [n,A] bytes - by the thread that owns the agent (the stepping pass's six strided copies per step, waymo_motion.py:206-311
 * with traffic_bots.py:123-143's history append, in the step's own launch). */
int tbx_train_chain_fwd_windows(const tbx_train_chain_t* c /* host */, const float* mean, int64_t mean_stride_n, int64_t mean_stride_t, int t0,
                                int t1, uint8_t* hv, float* hp, float* hm, uint8_t* valid, uint8_t* navi_valid, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K5/K7/K8/K9 + every dense contraction: a row-tile "chain" interpreter. One workgroup owns a tile of rows and runs
 * a short program of stages over it with the activations resident in LDS (two ping-pong buffers of `ldw` floats per
 * row + one 260-float auxiliary buffer) and the weights streamed once from L2/HBM into exact-fp32 MFMA
 * (v_mfma_f32_16x16x4_f32). Replaces the F.linear / layer_norm / relu / masked_fill / amax / cat chains of
 * modules/mlp.py:69-72, modules/input_encoder.py:51-59, modules/polyline_encoder.py:49-61 + utils/pooling.py:18-19,38,
 * modules/transformer_rpe.py:207-245 (LN, projections, out-proj + residual, FFN, row masking),
 * modules/add_navi_latent.py:43-65, modules/action_head.py:74-100, traffic_light.py:279-286, navigation.py:65-79.
 *
 * Row addressing: flat (group_rows == 0): tile t owns global rows [t*tile_rows, (t+1)*tile_rows);
 * grouped (group_rows = W > 0): tile t owns the W rows of group t (polyline / track), padded to tile_rows.
 */
enum {
  TBX_OP_LOAD = 1,      /* dst[:, dst_col:+n] (=|+=) p0[row_of(g) * ld + c]; cols [n, k) zero-filled (k = padded width) */
  TBX_OP_LINEAR = 2,    /* dst[:, dst_col:+n] (=|+=) act(src[:, src_col:+k] @ W^T + b); W=p0 [n,k] ld (or [k,n] if TBX_F_WT), b=p1.
                         * Grouped (block-diagonal) form: reserved = G > 1, div = (src_col_stride << 16) | dst_col_stride;
                         * group g uses the next n (or k if WT) weight rows and the next n bias entries. */
  TBX_OP_LAYERNORM = 3, /* dst[:, dst_col:+n] = LN(src[:, src_col:+n]) * p0 + p1, eps = f0 */
  TBX_OP_ADD = 4,       /* dst[:, dst_col:+n] += src[:, src_col:+n] */
  TBX_OP_COPY = 5,      /* dst[:, dst_col:+n]  = src[:, src_col:+n] */
  TBX_OP_ROWMASK = 6,   /* rows with p0[row_of(g)] != 0 (and padding rows): dst[:, dst_col:+n] = f0 */
  TBX_OP_GROUPMAX = 7,  /* dst[r, dst_col + c] = max_r' src[r', src_col + c]  (r' over the rows of r's group; flat mode: the tile).
                           p1 != NULL (a byte per global row): rows whose byte is set stay out of the maximum and come out as 0 in
                           the src and the dst columns - [ROWMASK(-inf), GROUPMAX, ROWMASK(0)] of a PointNet layer in one stage */
  TBX_OP_POOLMAX = 8,   /* p0[group * ld + dst_col + c] = max over un-masked rows (mask p1) of src[:, src_col + c]; none -> 0 */
  TBX_OP_STORE = 9,     /* p0[g * ld + dst_col + c] = src[:, src_col + c] for valid rows */
  TBX_OP_CLAMP = 10,    /* dst[:, dst_col:+n] = clamp(dst, f0, f1) */
  TBX_OP_DROPOUT = 11   /* dst[g, dst_col + c] *= keep ? f0 : 0 with tbx_keyed_dropout's mask for (seed = *(u64*)p0, site = div, step = k,
                           scene row = global row g, column c of n): keep = hash >= threshold = (uint32_t)reserved (p * 2^32),
                           f0 = 1 / (1 - p). The stepping (no-grad) pass of training runs its layers as chains with these stages. */
};
enum { TBX_ACT_NONE = 0, TBX_ACT_RELU = 1 };
enum {
  TBX_F_ACCUM = 1,    /* LOAD / LINEAR: accumulate into dst */
  TBX_F_WT = 2,       /* LINEAR: weight stored [k, n] (x @ W instead of x @ W^T) */
  TBX_F_ROW_DIV = 4,  /* row_of(g) = g / div   (per-group / per-agent broadcast) */
  TBX_F_ROW_MOD = 8,  /* row_of(g) = g % div */
  TBX_F_ROW_IDX = 16, /* row_of(g) = ((const int32_t*)p1)[g]  (LOAD gather) */
  TBX_F_ROW_BATCH_MOD = 32, /* row_of(g) = (g / div2) * div + g % div   with div2 packed in k (LOAD only) */
  TBX_F_WPACK = 64,   /* LINEAR: p0 is the tbx_pack_weight() image of the weight (ld / TBX_F_WT are then ignored) */
  TBX_F_MASK_INV = 128, /* ROWMASK: p0 holds a validity byte: rows with p0[row_of(g)] == 0 are filled */
  TBX_F_POOL_KEEP = 256, /* POOLMAX: the pooled rows also stay in LDS (row j of buffer `dst` != src, columns [k, k + n)) and the tile goes on
                           as a flat tile of its groups - global row = group index: the stages after it run on the pooled rows (the first
                           projection of the transformer that consumes them, in the launch that pooled them) */
  TBX_F_WSPLIT = 512,   /* LINEAR + TBX_F_WPACK: p0 is a tbx_pack_weight_split() image: split-bf16, three products on the bf16 MFMA */
  TBX_F_LOAD2 = 2048,   /* LOAD: a second source in the same stage (one memory round trip for both): buffer `src`, column `src_col`
                           (=) p2[row_of(g) * ld2 + c], c < reserved; whole float4s on both sides (n, ld, reserved, ld2 % 4 == 0,
                           16-byte aligned), reserved <= 256, no TBX_F_ACCUM */
  TBX_F_ROWSKIP = 1024, /* LINEAR + TBX_F_WPACK / TBX_F_WGEMV (LDS destination): p1 holds a byte per global row; rows whose byte is set
                           (clear with TBX_F_MASK_INV) and padding rows are NOT written - with TBX_F_ACCUM into the residual buffer
                           this is x += mask ? 0 : linear(...) in one stage (the attention / FFN output folded into the token row) */
  TBX_F_OUT_BF16 = 8192, /* STORE, and LINEAR with dst = TBX_BUF_GLOBAL: the global destination holds bfloat16 (round to nearest even);
                           ld / ld2 and dst_col count bf16 elements. For K/V tables read by tbx_knarpe_attn_fwd with seg.kv_bf16. */
  TBX_F_WGEMV = 4096,   /* LINEAR of a tbx_rowchain_live program: p0 is a tbx_pack_weight_gemv() image; a thread per output column,
                           v_fma chains in the MFMA path's k order (bit-identical results, a fraction of the latency at 1-4 rows) */
  TBX_F_ROWZERO = 32768, /* with TBX_F_ROWSKIP: the skipped rows (and padding rows) are written as 0 instead of keeping their content:
                           [LINEAR (+ accumulate)] followed by [ROWMASK fill 0] of the destination columns in one stage */
  TBX_F_MASKED_SUM = 16384 /* STORE (fp32 destination): p0[g * ld + dst_col + c] = sum over the `reserved` groups i whose byte
                           p1[i * k + g] is CLEAR of src[:, src_col + i * div + c] (0 when all are set): the masked sum over per-type
                           branches of action_head.py:89-96 in the storing stage (was ROWMASK + COPY/ADD per branch + STORE) */
};
enum { TBX_BUF0 = 0, TBX_BUF1 = 1, TBX_BUF_AUX = 2,
       TBX_BUF_GLOBAL = 3 /* LINEAR only: dst is global memory: p2[g * ld2 + dst_col + c] (valid rows), nothing staged in LDS */ };
#define TBX_MAX_STAGES 44
#define TBX_AUX_LD 260

typedef struct tbx_stage {
  int32_t op, src, dst, src_col, dst_col, k, n, act, flags, ld, div, reserved, ld2, pad; /* pad: set by the library */
  float f0, f1;
  const void* p0;
  const void* p1;
  const void* p2;
} tbx_stage_t;

/* Fragment-order image of a LINEAR weight + bias (nn.Linear [n,k] row-major with row stride ld, or [k,n] if wt != 0;
 * `groups` blocks stacked along dim 0 as LINEAR's grouped mode expects; bias [groups*n] or NULL = zeros): per 16-column
 * output tile, ceil(k/16) k-blocks stored as the 64 lanes x float4 an MFMA 16x16x4 B operand sequence consumes (zero-
 * padded to whole blocks) followed by the tile's bias replicated per lane, so every wavefront-wide weight load is one
 * contiguous 1 KiB and a stage's first fragments can be requested while the previous stage still computes. out holds
 * tbx_pack_weight_size(n, k, groups) floats. Pack once per weight update; the image replaces p0 in stages flagged
 * TBX_F_WPACK (their p1 / ld / TBX_F_WT are ignored). */
int64_t tbx_pack_weight_size(int n, int k, int groups);
int tbx_pack_weight(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out, void* stream);
/* Same image size and layout, but the 16 bytes of a (lane, k-block) hold the four weights as bf16 hi (8 B) + bf16 lo = bf16(w -
 * hi) (8 B). Stages flagged TBX_F_WPACK | TBX_F_WSPLIT split the activations the same way in registers and form
 * a_hi w_hi + a_hi w_lo + a_lo w_hi on v_mfma_f32_16x16x16_bf16 (fp32 accumulation): ~1e-5 relative error instead of the
 * exact-fp32 MFMA's ~1e-7, at 1/16 of the matrix-core time (the fp32 MFMA is the floor of a 16-row stage: 0.85 of 2.9 us). */
int tbx_pack_weight_split(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out, void* stream);

/* Column-stream image of the same weight + bias for TBX_F_WGEMV stages: per block of 128 output columns (outputs = groups*n,
 * group-major), 1 + ceil(k/16)*4 float4 rows of [128 columns][4]: row 0 = (bias, 0, 0, 0), row 1 + q = the four weights the MFMA
 * sequence multiplies in its k-block q / 4, step q % 4: k = (q/4)*16 + {0,4,8,12} + q%4 (zero-padded). The kernel streams it
 * through LDS in contiguous chunks of <= 66 KiB by LDS-DMA. tbx_pack_weight_gemv_size floats. */
int64_t tbx_pack_weight_gemv_size(int n, int k, int groups);
int tbx_pack_weight_gemv(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out, void* stream);

/* tile_rows in {16, 32, 48}; ldw % 4 == 0; LDS = (2*ldw + 260) * tile_rows * 4 bytes <= 160 KiB. */
int tbx_rowchain(const tbx_stage_t* stages /* host */, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                 int ldw, void* stream);

/* Same, with one LDS row width per buffer (BUF0 / BUF1 / AUX): programs whose wide intermediates live in BUF0 only (and
 * whose wide outputs go straight to global memory, TBX_BUF_GLOBAL) fit 32-row tiles or two workgroups per CU.
 * LDS = (ldw0 + ldw1 + ld_aux) * tile_rows * 4 bytes <= 160 KiB. */
int tbx_rowchain_ex(const tbx_stage_t* stages /* host */, int n_stages, int64_t n_rows, int group_rows, int tile_rows,
                    int ldw0, int ldw1, int ld_aux, void* stream);

/* The same interpreter on tiles of `live_rows` (1, 2 or 4) rows - for launches of a few dozen to a few hundred rows in all (one
 * 64-agent scene: the closed loop of BASELINE config 2), where a stage's latency, not its throughput, sets the step time: 4-16x
 * more workgroups than 16-row tiles, and every LINEAR stage must be TBX_F_WGEMV (a thread per output element running the MFMA
 * sequence's k-ordered fma chain on weights streamed through LDS, instead of MFMA tiles whose rows would be 3/4 .. 15/16 padding).
 * Flat programs only (no GROUPMAX / POOLMAX / DROPOUT). LDS = 4 rows x (ldw0 + ldw1 + ld_aux) floats + 132 KiB of weight slots +
 * the program <= 160 KiB. Results are bit-identical to the 16-row MFMA programs. */
int tbx_rowchain_live(const tbx_stage_t* stages /* host */, int n_stages, int64_t n_rows, int live_rows, int ldw0, int ldw1,
                      int ld_aux, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-step feature preparation (token frames, attribute rows, input pose embeddings).
 */

/* agent_encoder.py:130-159 (+ navigation.py:69-78 for the destination's relative pose, action_head.py:76 masks):
 * hist_* hold the W-step sliding window, oldest first, missing steps invalid.
 *   tok_pose [n,A,3] last-valid pose (0 if never valid), tok_invalid [n,A] u8,
 *   attr [n*A*W, 32]: size3 type3 | spd acc yaw_rate | one-hot_W(position) | 0-pad   (W <= 23)
 *   pe   [n*A*W, pe_dim]: pe_xy_yaw of the step's pose in the token frame, row_invalid [n*A*W] u8
 *   type_mask [3, n*A] u8: ~(type_i & valid_now)
 *   navi_pose3 [n*A, 3]: dest token pose relative to the agent's current pose; navi_row [n*A] i32 = b*M + dest
 */
/* tbx_agent_prep's arguments as a structure (tbx_heads_tail_t.next_prep): n_tok = n_batch * n_ag */
typedef struct tbx_agent_prep_args {
  const uint8_t* hist_valid;
  const float *hist_pose, *hist_motion, *ag_attr6;
  const uint8_t* ag_type_idx;
  const float *freqs_xy, *freqs_yaw;
  float* tok_pose;
  uint8_t* tok_invalid;
  float *attr, *pe;
  uint8_t *row_invalid, *type_mask;
  const int64_t* dest;
  const float* mp_tok_pose;
  float* navi_pose3;
  int32_t* navi_row;
  int32_t n_tok, n_ag, window, pe_dim, n_mp, mp_batch_div;
} tbx_agent_prep_args_t;
int tbx_agent_prep(const uint8_t* hist_valid, const float* hist_pose, const float* hist_motion, const float* ag_attr6,
                   const uint8_t* ag_type_idx, int n_batch, int n_ag, int window, const float* freqs_xy,
                   const float* freqs_yaw, int pe_dim, float* tok_pose, uint8_t* tok_invalid, float* attr, float* pe,
                   uint8_t* row_invalid, uint8_t* type_mask, const int64_t* dest, const float* mp_tok_pose, int n_mp,
                   int mp_batch_div, float* navi_pose3, int32_t* navi_row, void* stream);

/* traffic_light.py:219-226: attr [n*L*W, ld_attr] = one-hot5(state) | one-hot_W(position) | 0-pad; rows of missing
 * steps (hist_tl == 0xFF) and of invalid lights are flagged in row_invalid. hist_tl holds the 5-bit state mask per step. */
int tbx_tl_prep(const uint8_t* hist_tl, const uint8_t* tl_invalid, int n_batch, int n_tl, int window, int ld_attr,
                float* attr, uint8_t* row_invalid, void* stream);

/* map_encoder.py:64-77: per polyline node, attr [n*M*N, 32] = type11 | one-hot_N(node) | 0-pad, pe [n*M*N, 8] = the
 * 7-d MultiPath++ polyline feature of the node in the token (first node) frame (pose_emb.py:58-89) | 0. */
int tbx_map_prep(const uint8_t* mp_valid, const float* mp_type11, const float* mp_pose, int n_batch, int n_mp, int n_node,
                 float* attr, float* pe, uint8_t* row_invalid, float* tok_pose, uint8_t* tok_invalid, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K10: one closed-loop simulation step, all agents / lights of all scenes, no host round trip.
 * Replaces dynamics.py:66-163 (update_ag / MultiPathPP / override_ag / override_tl), dynamics.py:165-204
 * (disable_ag / disable_navi), teacher_forcing.py:108-167 (get), traffic_rule_checker.py:109-120,300-330
 * (outside-map and destination-reached checks: the two that feed back into the rollout) and
 * traffic_bots.py:123-143 (_append_hist), buffer.py:39-78 (add).
 * The step index is read from and incremented in `st->step` (device), so one captured graph replays every step.
 */
typedef struct tbx_sim_state {
  /* sizes */
  int32_t n_batch, n_ag, n_tl, window, n_step_gt, n_step_tl_gt, n_step_out, n_node;
  /* device step counter: step[0] = 1-based step being simulated, step[1] = workgroup arrival counter used when the
   * advance is fused into a part's kernel (zero-initialised by the caller, left at zero) */
  int32_t* step;
  /* agent state, updated in place */
  uint8_t* ag_valid;    /* [n,A] */
  uint8_t* ag_disabled; /* [n,A] */
  float* ag_pose;       /* [n,A,3] */
  float* ag_motion;     /* [n,A,3] */
  uint8_t* navi_valid;  /* [n,A] */
  uint8_t* outside_map; /* [n,A] accumulated */
  uint8_t* dest_reached;/* [n,A] accumulated */
  uint8_t* tl_state;    /* [n,L] 5-bit one-hot mask */
  /* sliding windows (oldest first) */
  uint8_t* hist_valid;  /* [n,A,W] */
  float* hist_pose;     /* [n,A,W,3] */
  float* hist_motion;   /* [n,A,W,3] */
  uint8_t* hist_tl;     /* [n,L,W] state mask, 0xFF = missing */
  /* static per rollout */
  const uint8_t* ag_type_idx; /* [n,A] 0 veh 1 ped 2 cyc */
  const uint8_t* tf_mask;     /* [n,A,n_step_gt] teacher-forcing / spawn mask */
  const uint8_t* gt_valid;    /* [n,A,n_step_gt] */
  const float* gt_pose;       /* [n,A,n_step_gt,3] */
  const float* gt_motion;     /* [n,A,n_step_gt,3] */
  const uint8_t* tl_gt;       /* [n,L,n_step_tl_gt] state masks */
  const float* boundary;      /* [n,4] xmin xmax ymin ymax */
  const float* dest_pos;      /* [n,A,n_node,2] */
  const float* dest_dir;      /* [n,A,n_node,2] unit */
  const uint8_t* dest_invalid;/* [n,A,n_node] */
  const uint8_t* dest_kind;   /* [n,A] bit0 lane (type<=3) bit1 road-edge boundary (type 4) */
  const float* dest_thresh;   /* [n,A] */
  /* model outputs of this step */
  const float* action_mean;   /* [n,A,2] unbounded */
  const float* tl_logits;     /* [n,L,5] */
  /* rollout log, written at [.., step-1, ..] */
  uint8_t* out_valid;         /* [n,A,T] validity BEFORE this step's override */
  float* out_pose;            /* [n,A,T,3] */
  float* out_motion;          /* [n,A,T,3] */
  float* out_action;          /* [n,A,T,2] */
  uint8_t* out_tl_state;      /* [n,L,T] */
  uint8_t* out_outside_map;   /* [n,A,T] */
  uint8_t* out_dest_reached;  /* [n,A,T] */
  float max_acc[3], max_yaw_rate[3]; /* per type idx (veh, ped, cyc) */
  float dt;
  /* ---- optional (NULL = off): the rest of RolloutBuffer.add (buffer.py:39-78, filled at waymo_motion.py:250-300) */
  float* out_reward;          /* [n,A,T,4] DifferentiableReward.get (rewards.py:35-85, loss.py:9-36 "cosine"):
                                 r_imitation_pos, r_imitation_rot, r_imitation_spd, diffbar_reward = (pos + rot) + spd */
  uint8_t* out_reward_valid;  /* [n,A,T] diffbar_reward_valid: pred_valid & gt_valid while ground truth lasts, else pred_valid */
  uint8_t* out_tf;            /* [n,A,T] mask_teacher_forcing = ag_override["valid"] of the step */
  float* out_tl_nll;          /* [n,L,T] -Categorical(logits).log_prob(argmax gt state), 0 once the light ground truth ended
                                 (waymo_motion.py:276-283) */
  float w_pos, w_rot, w_spd;  /* reward weights (l_pos / l_rot / l_spd .weight) */
  /* ---- optional: step-wise drivers (WaymoMotion.forward, waymo_motion.py:118-204) */
  const uint8_t* player_valid;  /* [n,A]   player_override["valid"]: the action below replaces the policy's (dynamics.py:104-107) */
  const float* player_action;   /* [n,A,2] physical (acc, yaw rate) */
  const uint8_t* ov_valid;      /* [n,A]   ag_override of THIS step given explicitly instead of tf_mask / gt_* at the step index */
  const float* ov_pose;         /* [n,A,3] */
  const float* ov_motion;       /* [n,A,3] */
  const uint8_t* ov_tl_valid;   /* [n,L]   tl_override["valid"] (with ov_valid) */
  const uint8_t* ov_tl_state;   /* [n,L]   5-bit state mask */
  uint8_t* now_outside;         /* [n,A]   outside_map_this_step  (for Dynamics.disable_ag by the caller, TBX_SIM_NO_DISABLE) */
  uint8_t* now_reached;         /* [n,A]   dest_reached_this_step */
} tbx_sim_state_t;

int tbx_sim_step(const tbx_sim_state_t* st /* host */, void* stream);

/* The same step in separately launchable parts. The traffic lights' recurrence (tl_state -> tl encoder -> tl_logits ->
 * tl_state, traffic_bots.py:188-199 + dynamics.py:143-163) never reads an agent, so a rollout may advance the lights on
 * one stream while the agents of the same step run on another; both parts read *step, TBX_SIM_ADVANCE bumps it and
 * must be ordered after both. With TBX_SIM_ADVANCE next to a part, the last workgroup of that part's kernel to arrive
 * bumps the counter (every workgroup has read it by then): no extra launch - for grids of up to 256 workgroups (the call uses
 * workgroups of up to 1024 threads to stay there: 8192 agents); a larger grid's arrivals on the one counter would serialise in L2
 * (~10 ns each), so there the call makes a second, one-thread launch on the same stream that bumps it. tbx_sim_step == all three,
 * one kernel (two for such a grid). */
enum { TBX_SIM_AGENTS = 1, TBX_SIM_LIGHTS = 2, TBX_SIM_ADVANCE = 4,
       /* Step-wise drivers split a step where the reference's Python does (waymo_motion.py:118-204 is `forward`, :250-275 the
        * caller's rule check + disable_ag / disable_navi, traffic_bots.py:123-143 appends the windows at the NEXT forward):
        * modifiers of TBX_SIM_AGENTS / TBX_SIM_LIGHTS ... */
       TBX_SIM_NO_DISABLE = 8,  /* flags are computed, logged and left in now_outside / now_reached; nothing is disabled */
       TBX_SIM_NO_APPEND = 16,  /* the sliding windows are left alone */
       /* ... and a part of its own: append the current agent / light state to the windows (no step is simulated) */
       TBX_SIM_APPEND = 32 };
int tbx_sim_step_parts(const tbx_sim_state_t* st /* host */, int parts, void* stream);
/* tbx_sim_step_parts with TBX_SIM_LIGHTS (appending), and tbx_tl_prep of the lights' new windows in the same launch: a light's thread
 * writes the ld_attr-wide one-hot rows and the row mask of its own window (traffic_light.py:219-226) right after shifting it.
 * tl_invalid [n_batch * n_tl] u8, attr [n_batch * n_tl * window, ld_attr], row_invalid [n_batch * n_tl * window]. tl_invalid NULL:
 * tbx_sim_step_parts. */
int tbx_sim_step_tl_prep(const tbx_sim_state_t* st /* host */, int parts, const uint8_t* tl_invalid, int ld_attr, float* attr,
                         uint8_t* row_invalid, void* stream);

/* utils/rewards.py:35-85 (DifferentiableReward.get, default configuration: the three imitation terms; w_collision = 0) for ONE step
 * on caller-supplied tensors - what tbx_sim_step logs into out_reward, as a call of its own. n = n_sc * n_ag rows.
 *   out4 [n, 4] = r_imitation_pos, r_imitation_rot ("cosine", metrics/loss.py:9-36), r_imitation_spd, diffbar_reward = (pos + rot) + spd
 *   out_valid [n] = pred_valid & gt_valid  (gt_valid NULL: pred_valid, all terms 0 - rewards.py:49-57) */
int tbx_diffbar_reward(const uint8_t* pred_valid, const float* pred_pose, const float* pred_motion, const uint8_t* gt_valid,
                       const float* gt_pose, const float* gt_motion, int64_t n, float w_pos, float w_rot, float w_spd, float* out4,
                       uint8_t* out_valid, void* stream);


/* ------------------------------------------------------------------------------------------------------------------
 * Metric-only traffic-rule checks of a rollout, batched over the (rollout, step) frames of the
 * rollout log. Replaces TrafficRuleChecker.check as called once per Python step (pl_modules/waymo_motion.py:250;
 * utils/traffic_rule_checker.py:342-451) for the five violations that do not feed back into the simulation:
 *   collided (:122-156), collided_wosac (utils/wosac_collision.py:211-257), run_road_edge (:158-181, 453-484),
 *   run_red_light (:183-233), passive (:235-298, 486-501).
 * Boolean results, bit-exact against the reference formulation (same fp32 operation order).
 */
enum { TBX_RULE_COLLIDED = 1, TBX_RULE_COLLIDED_WOSAC = 2, TBX_RULE_RUN_ROAD_EDGE = 4, TBX_RULE_RUN_RED_LIGHT = 8,
       TBX_RULE_PASSIVE = 16 };

/* Static tables of a scene (TrafficRuleChecker._get_road_edge / _get_lane_center): node segments (x0,y0,x1,y1 = pos,
 * pos + dir) of polylines of type 4 / 5 / 7 and node positions of polylines of type 0 / 1 / 2, valid nodes only,
 * compacted in unspecified order (every consumer is an any()).
 *   mp_valid [n_scene, n_mp, n_node] u8, mp_type_idx [n_scene, n_mp] u8 (argmax of the one-hot type),
 *   mp_pos / mp_dir [n_scene, n_mp, n_node, ld_xy] (ld_xy = 2 or 3: x, y first)
 *   seg [n_scene, n_mp*n_node, 4] (16-byte aligned), lane [n_scene, n_mp*n_node, 2], n_seg / n_lane [n_scene] i32 */
int tbx_rule_tables(const uint8_t* mp_valid, const uint8_t* mp_type_idx, const float* mp_pos, const float* mp_dir, int ld_xy,
                    int n_scene, int n_mp, int n_node, float* seg, int32_t* n_seg, float* lane, int32_t* n_lane, void* stream);

typedef struct tbx_rule_ctx {
  int32_t n_batch, n_ag, n_tl;
  int32_t map_batch_div;      /* rollouts [b*div, (b+1)*div) share scene b's tables */
  int32_t cap;                /* n_mp*n_node: rows per scene of seg / lane */
  const float* seg;
  const int32_t* n_seg;
  const float* lane;
  const int32_t* n_lane;
  const float* ag_size;       /* [n, A, 3] length width height, as handed to the reference constructor (un-scaled) */
  const uint8_t* ag_type_idx; /* [n, A] 0 veh 1 ped 2 cyc */
  const uint8_t* tl_valid;    /* [n, L] */
  const float* tl_pose;       /* [n, L, 3] */
  float collision_size_scale; /* 1.1 */
  /* optional (all NULL: every (frame, vehicle) scans the whole tables): tbx_rule_grid's outputs - seg / lane are then the SORTED
   * tables, *_start [n_scene, tbx_rule_grid_cells() + 1] the cells' first rows, *_grid [n_scene, 4] = (x0, y0, cells per metre,
   * half of the longest segment | 0). Bit-identical flags (the same predicates on a superset of the elements that can hold). */
  const int32_t* seg_start;
  const float* seg_grid;
  const int32_t* lane_start;
  const float* lane_grid;
} tbx_rule_ctx_t;

/* The scene tables of tbx_rule_tables sorted into the cells of a uniform raster over their bounding box (segments by midpoint): the
 * road-edge test (traffic_rule_checker.py:159-172) and the lane test of `passive` (:243-246) then visit the few cells a vehicle's
 * box / its 2 m disc overlaps instead of the scene's ~6,400 rows. seg_sorted / lane_sorted have the shapes of seg / lane. */
int tbx_rule_grid(const float* seg, const int32_t* n_seg, const float* lane, const int32_t* n_lane, int n_scene, int cap,
                  float* seg_sorted, int32_t* seg_start, float* seg_grid, float* lane_sorted, int32_t* lane_start, float* lane_grid,
                  void* stream);
int tbx_rule_grid_cells(void);

/* Raw per-frame flags for steps [t0, t0 + n_t) of a log with ld_t steps per agent (ld_t = 1, t0 = 0 for a single step):
 *   valid [n, A, ld_t] u8, pose / motion [n, A, ld_t, 3], tl_state [n, L, ld_t] u8 5-bit state masks
 *   flags [n, A, ld_t] u8: TBX_RULE_* bits of this step; TBX_RULE_PASSIVE here is the un-counted condition
 *   (traffic_rule_checker.py:291), the counter lives in tbx_rule_accumulate. n_ag <= 256. */
int tbx_rule_check(const tbx_rule_ctx_t* ctx /* host */, const uint8_t* valid, const float* pose, const float* motion,
                   const uint8_t* tl_state, int ld_t, int t0, int n_t, uint8_t* flags, void* stream);

/* traffic_rule_checker.py:293-296 + the running ORs of check(): acc_state [n_rows] u8 and passive_counter [n_rows] f32 are
 * read and updated (zero them before the first step); out_now / out_acc [n_rows, ld_t] u8 receive the `*_this_step` and the
 * accumulated flags of every step in the range. raw may alias out_now. */
int tbx_rule_accumulate(const uint8_t* raw, int n_rows, int ld_t, int t0, int n_t, uint8_t* acc_state, float* passive_counter,
                        uint8_t* out_now, uint8_t* out_acc, void* stream);

/* The checks of TrafficRuleChecker.check that feed back into the simulation, for ONE step (utils/traffic_rule_checker.py:109-120
 * _check_outside_map, :277-288 _check_goal_reached, :290-330 _check_dest_reached; called at pl_modules/waymo_motion.py:250 between
 * two `forward`s). The rollout engine's own loop evaluates outside-map / destination-reached inside tbx_sim_step; this entry point
 * serves a step-wise driver written against the reference.
 *   valid [n, A] u8, pose [n, A, 3], boundary [n / map_batch_div, 4] (xmin, xmax, ymin, ymax)
 *   dest_invalid [n, A, n_node] u8, dest_pos / dest_dir [n, A, n_node, 2] (dir = unit vectors), dest_kind [n, A] u8 (bit 0: lane
 *   destination - position within dest_thresh AND heading within 30 deg of a valid node -, bit 1: road-edge destination - position
 *   only), dest_thresh [n, A] f32: TrafficRuleChecker._get_dest (:87-107); all five NULL = no destinations (dest_reached stays 0)
 *   goal [n, A, 4] (x, y, yaw, v), goal_thresh [n, A] (8 agent lengths, :66); both NULL = no goals
 *   acc [3, n, A] u8: the running ORs outside_map, dest_reached, goal_reached - read and updated (zero them before the first step)
 *   out_now [3, n, A] u8: outside_map_this_step, dest_reached_this_step, goal_reached_this_step */
int tbx_rule_navi_check(const uint8_t* valid, const float* pose, const float* boundary, int map_batch_div, const uint8_t* dest_invalid,
                        const float* dest_pos, const float* dest_dir, const uint8_t* dest_kind, const float* dest_thresh, const float* goal,
                        const float* goal_thresh, int n_batch, int n_ag, int n_node, uint8_t* acc, uint8_t* out_now, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * WOSAC rollout filter. Replaces WOSACPostProcessing._filter_futures
 * (data_modules/wosac_post_processing.py:31-64): score_k = sum_a role_a * any_{t >= t_start} col[k,a,t]
 *                                                        + w_road_edge * sum_a role_a * any_{t >= t_start} edge[k,a,t],
 * keep the n_keep rollouts of each scene with the smallest score (the reference's topk(largest=False, sorted=False) leaves
 * ties unspecified; here ties go to the lower rollout index and idx is in ascending (score, index) order).
 *   flags [n_scene*n_k, n_ag, ld_t] u8 TBX_RULE_* bits (e.g. tbx_rule_accumulate's out_acc); col_bit = TBX_RULE_COLLIDED or
 *   TBX_RULE_COLLIDED_WOSAC (use_wosac_col); ag_role_any [n_scene, n_ag] u8 = ag_role.any(-1)
 *   score [n_scene, n_k] f32, idx [n_scene, n_keep] i32 (outputs)
 *   pred_pose [n_scene*n_k, n_ag, ld_t, 3] -> trajs [n_scene, n_keep, n_ag, ld_t - t_start, 3] (both NULL: selection only)
 * n_k <= 1024. */
int tbx_filter_futures(const uint8_t* flags, int col_bit, const uint8_t* ag_role_any, int n_scene, int n_k, int n_ag, int ld_t,
                       int t_start, float w_road_edge, int n_keep, float* score, int32_t* idx, const float* pred_pose,
                       float* trajs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TBX_HIP_H */
