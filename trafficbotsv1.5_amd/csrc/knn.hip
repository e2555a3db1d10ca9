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
__global__ __launch_bounds__(256) void knn_embed_kernel(const KnnArgs a) {
  constexpr int RPB = 4 / WPR;
  __shared__ float rel_s[RPB][64][3];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int wave = wv / WPR;  // row slot in the workgroup
  const int wir = wv % WPR;   // wave within the row
  const int row = blockIdx.x * RPB + wave;
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
    uint32_t key[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
      const int j = lane + 64 * q;
      key[q] = 0xffffffffu;
      if (j < a.n_tgt) {
        float rx, ry;
        rel_xy(x1, y1, c, s, tp[j * 3 + 0], tp[j * 3 + 1], rx, ry);
        const float dist = __fsqrt_rn(__fmaf_rn(ry, ry, __fmul_rn(rx, rx)));
        key[q] = __float_as_uint((inv1 || ti[j] != 0) ? INFINITY : dist);
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
        a.invalid[o] = (ti[j] != 0 || dist > a.dist_limit) ? 1 : 0;
        float rx, ry;
        rel_xy(x1, y1, c, s, tp[j * 3 + 0], tp[j * 3 + 1], rx, ry);
        const float ryaw = __fsub_rn(tp[j * 3 + 2], yaw1);
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
