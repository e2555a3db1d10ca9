// tbx_knn_embed / tbx_pose_embed: relative pose + K-nearest selection + pose embedding (see include/tbx_hip.h).
//
// One wavefront per source token. Each lane keeps n_tgt/64 candidate distances in registers; the K-th smallest distance is
// found by bisection over the key bits (wave ballots), the K winners are everything below it plus the lowest-index ties.
// Distances follow the reference's operation order without FMA contraction (SURVEY.md Appx A.2):
//   rx = dx*c + dy*s ; ry = dy*c - dx*s ; dist = sqrt(fma(ry, ry, rx*rx)) - torch's CPU 2-norm accumulates its squares
//   with a fused multiply-add (measured: 100 % bit-equal to this form, 91 % to the unfused one); +inf if either side is invalid.
// The embedding of the K selected relative poses is written by the same wave (lanes = channels), so the
// [S, T, 3] relative-pose tensor of the reference is never materialised.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

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

template <int MAXC, int WPR>
__global__ __launch_bounds__(256) void knn_embed_kernel(const KnnArgs a) {
  knn_rows<MAXC, WPR>(a, blockIdx.x);
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
__global__ __launch_bounds__(256) void knn_multi_kernel(const KnnMulti m) {
  if ((int)blockIdx.x >= m.first_block[KNN_MAX_JOBS]) {  // one thread per (pose, argument) as pose_embed_kernel
    const int half = m.pe.pe_dim >> 1;
    const int64_t e = (int64_t)((int)blockIdx.x - m.first_block[KNN_MAX_JOBS]) * blockDim.x + threadIdx.x;
    if (e >= m.pe.n * half) return;
    const int64_t i = e / half;
    const int c = (int)(e - i * half);
    tbx::pose_emb_write(m.pe.out + i * m.pe.ld + m.pe.col_off, m.pe.pe_dim, m.pe.pose3[i * 3], m.pe.pose3[i * 3 + 1], m.pe.pose3[i * 3 + 2],
                        m.pe.fxy, m.pe.fyaw, c, half);
    return;
  }
  int j = 0;
  while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first_block[j + 1]) ++j;
  const KnnArgs& a = m.job[j];
  const int block = blockIdx.x - m.first_block[j];
  if (a.n_tgt <= 128)
    knn_rows<2, 4>(a, block);
  else if (a.n_tgt <= 1024)
    knn_rows<16, 4>(a, block);
  else
    knn_rows<32, 4>(a, block);
}

__global__ void pose_embed_kernel(const float* __restrict__ pose3, int64_t n, const float* __restrict__ fxy,
                                  const float* __restrict__ fyaw, int pe_dim, float* __restrict__ out, int ld, int col_off) {
  const int half = pe_dim >> 1;  // one thread per (pose, argument): writes a cos and a sin channel
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * half) return;
  const int64_t i = e / half;
  const int c = (int)(e - i * half);
  tbx::pose_emb_write(out + i * ld + col_off, pe_dim, pose3[i * 3], pose3[i * 3 + 1], pose3[i * 3 + 2], fxy, fyaw, c, half);
}

}  // namespace

extern "C" int tbx_knn_embed(const float* src_pose, const uint8_t* src_invalid, const float* tgt_pose,
                             const uint8_t* tgt_invalid, int n_batch, int n_src, int n_tgt, int tgt_batch_div, int k,
                             float dist_limit, int32_t* idx, uint8_t* invalid, float* rel_pose, float* emb,
                             const float* freqs_xy, const float* freqs_yaw, int pe_dim, void* stream) {
  if (!src_pose || !src_invalid || !tgt_pose || !tgt_invalid || !idx || !invalid) return TBX_ERR_ARG;
  if (n_batch <= 0 || n_src <= 0 || n_tgt <= 0 || tgt_batch_div <= 0 || n_batch % tgt_batch_div != 0) return TBX_ERR_ARG;
  if (k <= 0 || k >= n_tgt || k > 64 || n_tgt > 2048) return TBX_ERR_UNSUPPORTED;
  if (emb != nullptr && (!freqs_xy || !freqs_yaw || (pe_dim != 64 && pe_dim != 128))) return TBX_ERR_UNSUPPORTED;
  KnnArgs a{src_pose, src_invalid, tgt_pose, tgt_invalid, idx, invalid, rel_pose, emb, freqs_xy, freqs_yaw,
            n_batch * n_src, n_src, n_tgt, tgt_batch_div, k, pe_dim, dist_limit};
  const dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool small = a.n_rows < 4096;
  const dim3 grid(small ? a.n_rows : (a.n_rows + 3) / 4);
#define TBX_KNN_LAUNCH(MAXC)                                                   \
  do {                                                                         \
    if (small)                                                                 \
      hipLaunchKernelGGL((knn_embed_kernel<MAXC, 4>), grid, block, 0, s, a);   \
    else                                                                       \
      hipLaunchKernelGGL((knn_embed_kernel<MAXC, 1>), grid, block, 0, s, a);   \
  } while (0)
  if (n_tgt <= 128)
    TBX_KNN_LAUNCH(2);
  else if (n_tgt <= 1024)
    TBX_KNN_LAUNCH(16);
  else
    TBX_KNN_LAUNCH(32);
#undef TBX_KNN_LAUNCH
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_knn_embed_multi(const tbx_knn_job_t* jobs, int n_jobs, const float* freqs_xy, const float* freqs_yaw, int pe_dim,
                                   void* stream) {
  return tbx_knn_embed_multi_pe(jobs, n_jobs, freqs_xy, freqs_yaw, pe_dim, nullptr, stream);
}

extern "C" int tbx_knn_embed_multi_pe(const tbx_knn_job_t* jobs, int n_jobs, const float* freqs_xy, const float* freqs_yaw, int pe_dim,
                                      const tbx_pose_embed_job_t* pe, void* stream) {
  if (!jobs || n_jobs <= 0 || n_jobs > KNN_MAX_JOBS) return TBX_ERR_ARG;
  KnnMulti m;
  m.n_jobs = n_jobs;
  m.pe = PoseEmbedArgs{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  if (pe != nullptr) {
    if (!pe->pose3 || !pe->freqs_xy || !pe->freqs_yaw || !pe->out || pe->n <= 0) return TBX_ERR_ARG;
    if ((pe->pe_dim != 64 && pe->pe_dim != 128) || pe->ld_out < pe->col_off + pe->pe_dim) return TBX_ERR_UNSUPPORTED;
    m.pe = PoseEmbedArgs{pe->pose3, pe->freqs_xy, pe->freqs_yaw, pe->out, pe->n, pe->pe_dim, pe->ld_out, pe->col_off};
  }
  int blocks = 0;
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
  hipLaunchKernelGGL(knn_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, m);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

namespace {
// Inverse of a K-nearest set, per target table: for every target token the (source row, slot) pairs that selected it
// (masked pairs left out). One workgroup per table: counts in LDS, an exclusive scan, then a fill through LDS cursors (the
// order inside a target's list is unspecified). pair id = global row * k + slot.
constexpr int INV_MAX_TGT = 2048;
__global__ __launch_bounds__(1024) void knn_inverse_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ invalid,
                                                           int n_src, int k, int n_tgt, int batch_div,
                                                           int32_t* __restrict__ inv_ptr, int32_t* __restrict__ inv_list) {
  __shared__ int cnt[INV_MAX_TGT + 1];
  __shared__ int part[1024];
  const int table = blockIdx.x;
  const int rows = n_src * batch_div;  // consecutive batches share the table
  const int64_t row0 = (int64_t)table * rows;
  const int n_pairs = rows * k;
  for (int j = threadIdx.x; j <= n_tgt; j += blockDim.x) cnt[j] = 0;
  __syncthreads();
  for (int p = threadIdx.x; p < n_pairs; p += blockDim.x) {
    const int64_t g = row0 * k + p;
    if (invalid[g] == 0) atomicAdd(&cnt[idx[g]], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0..n_tgt): each thread owns a contiguous chunk
  const int chunk = (n_tgt + blockDim.x - 1) / blockDim.x;
  const int j0 = threadIdx.x * chunk, j1 = min(j0 + chunk, n_tgt);
  int sum = 0;
  for (int j = j0; j < j1; ++j) sum += cnt[j];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < (int)blockDim.x; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
  }
  __syncthreads();
  int run = part[threadIdx.x];
  int32_t* ptr = inv_ptr + (int64_t)table * (n_tgt + 1);
  for (int j = j0; j < j1; ++j) {
    const int v = cnt[j];
    ptr[j] = run;
    cnt[j] = run;  // becomes the fill cursor
    run += v;
  }
  if (j1 == n_tgt && j0 < n_tgt) ptr[n_tgt] = run;
  if (n_tgt == 0 && threadIdx.x == 0) ptr[0] = 0;
  __syncthreads();
  int32_t* list = inv_list + (int64_t)table * n_pairs;
  for (int p = threadIdx.x; p < n_pairs; p += blockDim.x) {
    const int64_t g = row0 * k + p;
    if (invalid[g] == 0) list[atomicAdd(&cnt[idx[g]], 1)] = (int32_t)g;
  }
}
}  // namespace

extern "C" int tbx_knn_inverse(const int32_t* idx, const uint8_t* invalid, int n_batch, int n_src, int k, int n_tgt,
                               int tgt_batch_div, int32_t* inv_ptr, int32_t* inv_list, void* stream) {
  if (!idx || !invalid || !inv_ptr || !inv_list) return TBX_ERR_ARG;
  if (n_batch <= 0 || n_src <= 0 || k <= 0 || n_tgt <= 0 || tgt_batch_div <= 0 || n_batch % tgt_batch_div) return TBX_ERR_ARG;
  if (n_tgt > INV_MAX_TGT || (int64_t)n_batch * n_src * k > 0x7fffffff) return TBX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(knn_inverse_kernel, dim3(n_batch / tgt_batch_div), dim3(1024), 0, (hipStream_t)stream, idx, invalid, n_src, k,
                     n_tgt, tgt_batch_div, inv_ptr, inv_list);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_pose_embed(const float* pose3, int64_t n, const float* freqs_xy, const float* freqs_yaw, int pe_dim,
                              float* out, int ld_out, int col_off, void* stream) {
  if (!pose3 || !freqs_xy || !freqs_yaw || !out || n <= 0) return TBX_ERR_ARG;
  if ((pe_dim != 64 && pe_dim != 128) || ld_out < col_off + pe_dim) return TBX_ERR_UNSUPPORTED;
  const int64_t total = n * (pe_dim / 2);
  hipLaunchKernelGGL(pose_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pose3, n,
                     freqs_xy, freqs_yaw, pe_dim, out, ld_out, col_off);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
