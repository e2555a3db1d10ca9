// The K-nearest search of one source row (see csrc/knn.hip for the design) and the several-searches-per-launch dispatch as device
// functions, shared by tbx_knn_embed*'s kernels and the fused front launch (csrc/front.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace tbx_knn {

struct KnnArgs {
  const float* src_pose;
  const uint8_t* src_invalid;
  const float* tgt_pose;
  const uint8_t* tgt_invalid;
  int32_t* idx;
  uint8_t* invalid;
  float* rel_pose;
  float* emb;
  const float* fxy;
  const float* fyaw;
  int n_rows, n_src, n_tgt, tgt_batch_div, k, pe_dim;
  float dist_limit;
};

__device__ __forceinline__ void rel_xy(float x1, float y1, float c, float s, float x2, float y2, float& rx, float& ry) {
  const float dx = __fsub_rn(x2, x1), dy = __fsub_rn(y2, y1);
  rx = __fadd_rn(__fmul_rn(dx, c), __fmul_rn(dy, s));
  ry = __fadd_rn(__fmul_rn(dx, -s), __fmul_rn(dy, c));
}

// WPR = wavefronts per source row: the selection is done by the row's first wave, the K embeddings are split over all
// WPR waves (4 on the small grids of a few scenes, where the kernel is latency-bound; 1 on large grids).
template <int MAXC, int WPR>
__device__ __forceinline__ void knn_rows(const KnnArgs& a, int block) {
  constexpr int RPB = 4 / WPR;
  __shared__ float rel_s[RPB][64][3];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int wave = wv / WPR;  // row slot in the workgroup
  const int wir = wv % WPR;   // wave within the row
  const int row = block * RPB + wave;
  if (row >= a.n_rows) return;  // uniform per row (per workgroup when WPR == 4)
  const int b = row / a.n_src;
  const int bt = b / a.tgt_batch_div;
  const float x1 = a.src_pose[row * 3 + 0], y1 = a.src_pose[row * 3 + 1], yaw1 = a.src_pose[row * 3 + 2];
  const bool inv1 = a.src_invalid[row] != 0;
  const float c = cosf(yaw1), s = sinf(yaw1);
  const float* tp = a.tgt_pose + (int64_t)bt * a.n_tgt * 3;
  const uint8_t* ti = a.tgt_invalid + (int64_t)bt * a.n_tgt;

  if (wir == 0) {
    // Candidate keys: the fp32 bits of the distance (non-negative, so unsigned order == float order; +inf for masked
    // pairs), 0xffffffff for slots past n_tgt. Lane l owns targets l, l+64, ...
    // The candidates' relative poses stay in registers from here to the output phase (rx, ry, yaw difference, the target's own
    // invalid bit): the winners are written without touching the target tables again (a dependent L2 round trip per candidate
    // slot otherwise - 16 of them in a row were most of this kernel's time on the latency-bound grids of a few scenes).
    uint32_t key[MAXC];
    float crx[MAXC], cry[MAXC], cyaw[MAXC];
    uint32_t tinv = 0;
    {
      float tx[MAXC], ty[MAXC];
      uint8_t tb[MAXC];
#pragma unroll
      for (int q = 0; q < MAXC; ++q) {  // all loads first, on clamped indices (no branch between them: one latency, not MAXC)
        const int j = min(lane + 64 * q, a.n_tgt - 1);
        tx[q] = tp[j * 3 + 0];
        ty[q] = tp[j * 3 + 1];
        cyaw[q] = tp[j * 3 + 2];
        tb[q] = ti[j];
      }
#pragma unroll
      for (int q = 0; q < MAXC; ++q) {
        const bool tin = tb[q] != 0;
        rel_xy(x1, y1, c, s, tx[q], ty[q], crx[q], cry[q]);
        cyaw[q] = __fsub_rn(cyaw[q], yaw1);
        tinv |= tin ? (1u << q) : 0u;
        const float dist = __fsqrt_rn(__fmaf_rn(cry[q], cry[q], __fmul_rn(crx[q], crx[q])));
        key[q] = (lane + 64 * q < a.n_tgt) ? __float_as_uint((inv1 || tin) ? INFINITY : dist) : 0xffffffffu;
      }
    }
    // K-th smallest key by bisection over its 31 value bits: count(key < cand) is a sum of wave ballots' popcounts, so
    // it lands in a scalar register and the pivot update is scalar too. ~50 instructions per bit instead of the ~150
    // per extracted neighbour of a K-round argmin (K = 64 for agent -> map).
    uint32_t kth = 0;
    for (int bit = 30; bit >= 0; --bit) {
      const uint32_t cand = kth | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < MAXC; ++q) cnt += __popcll(__ballot(key[q] < cand));
      if (cnt < a.k) kth = cand;
    }
    // Everything below the K-th key is in; keys equal to it fill the remaining slots in ascending target index (the
    // reference's topk leaves the order among equal distances open; masked +inf pairs are don't-cares). Output slots
    // are assigned in ascending target index: position = number of chosen targets with a smaller index.
    int n_less = 0;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) n_less += __popcll(__ballot(key[q] < kth));
    const int n_ties = a.k - n_less;  // >= 1
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    int ties_before = 0, chosen_before = 0;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
      const bool tie = key[q] == kth;
      const uint64_t tie_m = __ballot(tie);
      const bool take = key[q] < kth || (tie && ties_before + __popcll(tie_m & lt_mask) < n_ties);
      const uint64_t take_m = __ballot(take);
      if (take) {
        const int pos = chosen_before + __popcll(take_m & lt_mask);
        const int j = lane + 64 * q;
        const int64_t o = (int64_t)row * a.k + pos;
        const float dist = __uint_as_float(key[q]);
        a.idx[o] = j;
        a.invalid[o] = (((tinv >> q) & 1u) != 0 || dist > a.dist_limit) ? 1 : 0;
        const float rx = crx[q], ry = cry[q], ryaw = cyaw[q];
        rel_s[wave][pos][0] = rx;
        rel_s[wave][pos][1] = ry;
        rel_s[wave][pos][2] = ryaw;
        if (a.rel_pose != nullptr) {
          a.rel_pose[o * 3 + 0] = rx;
          a.rel_pose[o * 3 + 1] = ry;
          a.rel_pose[o * 3 + 2] = ryaw;
        }
      }
      ties_before += __popcll(tie_m);
      chosen_before += __popcll(take_m);
    }
  }  // wir == 0
  if (a.emb == nullptr) return;
  if constexpr (WPR > 1)
    __syncthreads();
  else
    __builtin_amdgcn_wave_barrier();  // LDS is in-order per wave; this only pins the compiler's ordering
  for (int t = wir; t < a.k; t += WPR) {
    const float x = rel_s[wave][t][0], y = rel_s[wave][t][1], yaw = rel_s[wave][t][2];
    tbx::pose_emb_write(a.emb + ((int64_t)row * a.k + t) * a.pe_dim, a.pe_dim, x, y, yaw, a.fxy, a.fyaw, lane, 64);
  }
}

// Several searches in one launch (the agents' three K-nearest sets of a simulation step: one grid instead of three dependent
// launches on the auxiliary stream). A workgroup = one source row of one job (the 4-waves-per-row form); jobs are laid out
// back to back over blockIdx.x in the order given (the longest search first, so that its rows are dispatched first).
constexpr int KNN_MAX_JOBS = 4;
struct PoseEmbedArgs {  // tbx_pose_embed riding on the searches' launch (tbx_knn_embed_multi_pe): the blocks past the last search
  const float* pose3;
  const float *fxy, *fyaw;
  float* out;
  int64_t n;
  int pe_dim, ld, col_off;
};
struct KnnMulti {
  KnnArgs job[KNN_MAX_JOBS];
  int first_block[KNN_MAX_JOBS + 1];
  int n_jobs;
  PoseEmbedArgs pe;  // pe.n == 0: none
};
// `bid` = the workgroup's index among the launch's search / embedding workgroups, by 256 threads
__device__ __forceinline__ void knn_multi_body(const KnnMulti& m, const int bid) {
  if (bid >= m.first_block[KNN_MAX_JOBS]) {  // one thread per (pose, argument) as pose_embed_kernel
    const int half = m.pe.pe_dim >> 1;
    const int64_t e = (int64_t)(bid - m.first_block[KNN_MAX_JOBS]) * 256 + threadIdx.x;
    if (e >= m.pe.n * half) return;
    const int64_t i = e / half;
    const int c = (int)(e - i * half);
    tbx::pose_emb_write(m.pe.out + i * m.pe.ld + m.pe.col_off, m.pe.pe_dim, m.pe.pose3[i * 3], m.pe.pose3[i * 3 + 1], m.pe.pose3[i * 3 + 2],
                        m.pe.fxy, m.pe.fyaw, c, half);
    return;
  }
  int j = 0;
  while (j + 1 < m.n_jobs && bid >= m.first_block[j + 1]) ++j;
  const KnnArgs& a = m.job[j];
  const int block = bid - m.first_block[j];
  if (a.n_tgt <= 128)
    knn_rows<2, 4>(a, block);
  else if (a.n_tgt <= 1024)
    knn_rows<16, 4>(a, block);
  else
    knn_rows<32, 4>(a, block);
}


// host: the launch descriptor of tbx_knn_embed_multi_pe's arguments; blocks = its workgroups (256 threads each)
inline int knn_multi_fill(const tbx_knn_job_t* jobs, int n_jobs, const float* freqs_xy, const float* freqs_yaw, int pe_dim,
                          const tbx_pose_embed_job_t* pe, KnnMulti& m, int& blocks) {
  if (!jobs || n_jobs <= 0 || n_jobs > KNN_MAX_JOBS) return TBX_ERR_ARG;
  m.n_jobs = n_jobs;
  m.pe = PoseEmbedArgs{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  if (pe != nullptr) {
    if (!pe->pose3 || !pe->freqs_xy || !pe->freqs_yaw || !pe->out || pe->n <= 0) return TBX_ERR_ARG;
    if ((pe->pe_dim != 64 && pe->pe_dim != 128) || pe->ld_out < pe->col_off + pe->pe_dim) return TBX_ERR_UNSUPPORTED;
    m.pe = PoseEmbedArgs{pe->pose3, pe->freqs_xy, pe->freqs_yaw, pe->out, pe->n, pe->pe_dim, pe->ld_out, pe->col_off};
  }
  blocks = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const tbx_knn_job_t& q = jobs[j];
    if (!q.src_pose || !q.src_invalid || !q.tgt_pose || !q.tgt_invalid || !q.idx || !q.invalid) return TBX_ERR_ARG;
    if (q.n_batch <= 0 || q.n_src <= 0 || q.n_tgt <= 0 || q.tgt_batch_div <= 0 || q.n_batch % q.tgt_batch_div != 0) return TBX_ERR_ARG;
    if (q.k <= 0 || q.k >= q.n_tgt || q.k > 64 || q.n_tgt > 2048) return TBX_ERR_UNSUPPORTED;
    if (q.emb != nullptr && (!freqs_xy || !freqs_yaw || (pe_dim != 64 && pe_dim != 128))) return TBX_ERR_UNSUPPORTED;
    m.job[j] = KnnArgs{q.src_pose, q.src_invalid, q.tgt_pose, q.tgt_invalid, q.idx, q.invalid, q.rel_pose, q.emb, freqs_xy, freqs_yaw,
                       q.n_batch * q.n_src, q.n_src, q.n_tgt, q.tgt_batch_div, q.k, pe_dim, q.dist_limit};
    m.first_block[j] = blocks;
    blocks += q.n_batch * q.n_src;
  }
  for (int j = n_jobs; j <= KNN_MAX_JOBS; ++j) m.first_block[j] = blocks;
  for (int j = n_jobs; j < KNN_MAX_JOBS; ++j) m.job[j] = m.job[0];
  if (m.pe.n > 0) blocks += (int)((m.pe.n * (m.pe.pe_dim >> 1) + 255) / 256);  // the pose embedding's blocks come last
  return TBX_OK;
}

}  // namespace tbx_knn
