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
#include "knn_core.h"

namespace {

using namespace tbx_knn;

template <int MAXC, int WPR>
__global__ __launch_bounds__(256) void knn_embed_kernel(const KnnArgs a) {
  knn_rows<MAXC, WPR>(a, blockIdx.x);
}

__global__ __launch_bounds__(256) void knn_multi_kernel(const KnnMulti m) { knn_multi_body(m, (int)blockIdx.x); }

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
  KnnMulti m;
  int blocks = 0;
  const int rc = tbx_knn::knn_multi_fill(jobs, n_jobs, freqs_xy, freqs_yaw, pe_dim, pe, m, blocks);
  if (rc != TBX_OK) return rc;
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

// utils/rpe.py:8-58 as dense tensors (the stand-alone `get_rel_pose` / `get_rel_dist` of the reference): one thread per (source,
// target) pair, the SAME expressions as the K-nearest search above (tbx_knn::rel_xy, un-fused distance), so a dense distance equals
// the key the search ranks by, bit for bit.
namespace {
__global__ __launch_bounds__(256) void rel_pose_dense_kernel(const float* __restrict__ src_pose, const uint8_t* __restrict__ src_invalid,
                                                             const float* __restrict__ tgt_pose, const uint8_t* __restrict__ tgt_invalid,
                                                             int64_t n_pairs, int n_src, int n_tgt, int tgt_batch_div,
                                                             float* __restrict__ rel_pose, float* __restrict__ rel_dist) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const int64_t row = p / n_tgt;
  const int j = (int)(p - row * n_tgt);
  const int64_t bt = (row / n_src) / tgt_batch_div;
  const float x1 = src_pose[row * 3], y1 = src_pose[row * 3 + 1], yaw1 = src_pose[row * 3 + 2];
  const float* t = tgt_pose + (bt * n_tgt + j) * 3;
  float rx, ry;
  tbx_knn::rel_xy(x1, y1, cosf(yaw1), sinf(yaw1), t[0], t[1], rx, ry);
  if (rel_pose != nullptr) {
    rel_pose[p * 3] = rx;
    rel_pose[p * 3 + 1] = ry;
    rel_pose[p * 3 + 2] = __fsub_rn(t[2], yaw1);  // not wrapped (rpe.py:32, cast=False)
  }
  if (rel_dist != nullptr) {
    const bool inv = src_invalid[row] != 0 || tgt_invalid[bt * n_tgt + j] != 0;
    rel_dist[p] = inv ? INFINITY : __fsqrt_rn(__fmaf_rn(ry, ry, __fmul_rn(rx, rx)));
  }
}
}  // namespace

extern "C" int tbx_rel_pose_dense(const float* src_pose, const uint8_t* src_invalid, const float* tgt_pose, const uint8_t* tgt_invalid,
                                  int n_batch, int n_src, int n_tgt, int tgt_batch_div, float* rel_pose, float* rel_dist, void* stream) {
  if (!src_pose || !src_invalid || !tgt_pose || !tgt_invalid || (!rel_pose && !rel_dist)) return TBX_ERR_ARG;
  if (n_batch <= 0 || n_src <= 0 || n_tgt <= 0 || tgt_batch_div <= 0 || n_batch % tgt_batch_div != 0) return TBX_ERR_ARG;
  const int64_t n_pairs = (int64_t)n_batch * n_src * n_tgt;
  if ((n_pairs + 255) / 256 > 0x7fffffff) return TBX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(rel_pose_dense_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_pose,
                     src_invalid, tgt_pose, tgt_invalid, n_pairs, n_src, n_tgt, tgt_batch_div, rel_pose, rel_dist);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
