// tbx_attn_fold_fwd / _bwd: the folded GEMM weights of ONE AttentionRPE module (the exact algebra that moves
// linear_rpe's per-pair projection out of the pair loop), forward and backward, as one launch each.
//   [q | qt] = x W_in^T + b_in      W_in  = [W_q ; B_k^T W_q]   (640 x 128)    b_in  = [b_q ; B_k^T b_q]
//   y = [sum a v | sum a e] W_out^T + b_out     W_out = [W_o | W_o B_v^T] (128 x 640)   b_out = W_o b_rv + b_o
// with B_k / B_v the per-head blocks of linear_rpe's key / value halves (attention_rpe.py:92-98,137-164,181-190). In torch this was
// 3 batched GEMMs, 3 concatenations, 6 slices, a matrix-vector product and an add per module forward, and ~25 kernels backward
// (slice backward = zero fill + strided add per slice, bmm backward, cat backward) - x 40 attention modules x 2 (the no-grad stepping
// pass has its own copies): ~2,000 launches of 2-5 us in a training step whose tensors hold 64 K floats at most.
// Sums are fp32 in a fixed order (deterministic). One thread per output element; dot products of 32 or 128 terms.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

constexpr int D = 128, NH = 4, DH = 32;

struct FoldArgs {
  const float *w, *b;    // in_proj [384,128], [384]
  const float *wr, *br;  // linear_rpe [256,128], [256]: key half rows 0..127, value half rows 128..255 (row = h * 32 + i)
  const float *wo, *bo;  // out_proj [128,128], [128]
  float *w_in, *b_in, *w_kv, *b_kv, *bias_k, *w_out, *b_out;  // [640,128], [640], [256,128], [256], [128], [128,640], [128]
};

// blocks: 0..639 rows of w_in (+ b_in), 640..767 rows of w_out (+ b_out), 768..1023 rows of w_kv (+ b_kv, bias_k); 128 threads
__global__ __launch_bounds__(128) void fold_fwd_kernel(const FoldArgs a) {
  const int r = blockIdx.x, j = threadIdx.x;
  if (r < 5 * D) {
    if (r < D) {
      a.w_in[r * D + j] = a.w[r * D + j];
      if (j == 0) a.b_in[r] = a.b[r];
      return;
    }
    // row 128 + h * 128 + c: sum_i wr[h*32+i, c] * Wq[h*32+i, j]
    const int h = (r - D) / D, c = (r - D) % D;
    float acc = 0.f, accb = 0.f;
    for (int i = 0; i < DH; ++i) {
      const float wk = a.wr[(h * DH + i) * D + c];
      acc = fmaf(wk, a.w[(h * DH + i) * D + j], acc);
      accb = fmaf(wk, a.b[h * DH + i], accb);
    }
    a.w_in[r * D + j] = acc;
    if (j == 0) a.b_in[r] = accb;
    return;
  }
  if (r < 6 * D) {
    const int ro = r - 5 * D;  // output row of W_out; thread j covers columns j, 128 + j, .., 512 + j
    a.w_out[ro * 5 * D + j] = a.wo[ro * D + j];
    for (int h = 0; h < NH; ++h) {  // column 128 + h * 128 + c (c = j): sum_i wo[ro, h*32+i] * wr[128 + h*32+i, c]
      float acc = 0.f;
      for (int i = 0; i < DH; ++i) acc = fmaf(a.wo[ro * D + h * DH + i], a.wr[(D + h * DH + i) * D + j], acc);
      a.w_out[ro * 5 * D + D + h * D + j] = acc;
    }
    if (j == 0) {
      float acc = 0.f;
      for (int q = 0; q < D; ++q) acc = fmaf(a.wo[ro * D + q], a.br[D + q], acc);
      a.b_out[ro] = acc + a.bo[ro];
    }
    return;
  }
  const int rk = r - 6 * D;  // 0..255
  a.w_kv[rk * D + j] = a.w[(D + rk) * D + j];
  if (j == 0) a.b_kv[rk] = a.b[D + rk];
  if (rk == 0) a.bias_k[j] = a.br[j];
}

struct FoldBwdArgs {
  FoldArgs f;  // inputs as the forward (outputs unused)
  const float *g_w_in, *g_b_in, *g_w_kv, *g_b_kv, *g_bias_k, *g_w_out, *g_b_out;  // any may be NULL (no gradient arrived)
  float *d_w, *d_b, *d_wr, *d_br, *d_wo, *d_bo;  // overwritten
};

__device__ __forceinline__ float ld0(const float* p, int64_t i) { return p == nullptr ? 0.f : p[i]; }

// blocks: 0..127 rows of d W_q (+ d b_q), 128..383 rows of d W_kv (+ d b_kv), 384..511 key-half rows of d wr (+ d br key = g_bias_k),
// 512..639 value-half rows of d wr (+ d br value), 640..767 rows of d wo (+ d bo); 128 threads
__global__ __launch_bounds__(128) void fold_bwd_kernel(const FoldBwdArgs a) {
  const FoldArgs& f = a.f;
  const int r = blockIdx.x, j = threadIdx.x;
  if (r < D) {
    // d Wq[r, j] = G_in[r, j] + sum_c wr[r, c] G_in[128 + h*128 + c, j],  h = r / 32
    const int h = r / DH;
    float acc = ld0(a.g_w_in, r * D + j);
    if (a.g_w_in != nullptr)
      for (int c = 0; c < D; ++c) acc = fmaf(f.wr[r * D + c], a.g_w_in[(D + h * D + c) * D + j], acc);
    a.d_w[r * D + j] = acc;
    if (j == 0) {
      float ab = ld0(a.g_b_in, r);
      if (a.g_b_in != nullptr)
        for (int c = 0; c < D; ++c) ab = fmaf(f.wr[r * D + c], a.g_b_in[D + h * D + c], ab);
      a.d_b[r] = ab;
    }
    return;
  }
  if (r < 3 * D) {
    a.d_w[r * D + j] = ld0(a.g_w_kv, (r - D) * D + j);
    if (j == 0) a.d_b[r] = ld0(a.g_b_kv, r - D);
    return;
  }
  if (r < 4 * D) {
    // key half: d wr[rk, c = j] = sum_q Wq[rk, q] G_in[128 + h*128 + c, q] + b[rk] g_b_in[128 + h*128 + c]
    const int rk = r - 3 * D, h = rk / DH;
    float acc = 0.f;
    if (a.g_w_in != nullptr)
      for (int q = 0; q < D; ++q) acc = fmaf(f.w[rk * D + q], a.g_w_in[(D + h * D + j) * D + q], acc);
    if (a.g_b_in != nullptr) acc = fmaf(f.b[rk], a.g_b_in[D + h * D + j], acc);
    a.d_wr[rk * D + j] = acc;
    if (j == 0) a.d_br[rk] = ld0(a.g_bias_k, rk);
    return;
  }
  if (r < 5 * D) {
    // value half: d wr[128 + rv, c = j] = sum_ro wo[ro, rv] G_out[ro, 128 + h*128 + c]
    const int rv = r - 4 * D, h = rv / DH;
    float acc = 0.f;
    if (a.g_w_out != nullptr)
      for (int ro = 0; ro < D; ++ro) acc = fmaf(f.wo[ro * D + rv], a.g_w_out[ro * 5 * D + D + h * D + j], acc);
    a.d_wr[(D + rv) * D + j] = acc;
    if (j == 0) {
      float ab = 0.f;
      if (a.g_b_out != nullptr)
        for (int ro = 0; ro < D; ++ro) ab = fmaf(f.wo[ro * D + rv], a.g_b_out[ro], ab);
      a.d_br[D + rv] = ab;
    }
    return;
  }
  // d wo[ro, j] = G_out[ro, j] + sum_c G_out[ro, 128 + h*128 + c] wr[128 + j, c] + g_b_out[ro] br[128 + j],  h = j / 32
  const int ro = r - 5 * D, h = j / DH;
  float acc = ld0(a.g_w_out, ro * 5 * D + j);
  if (a.g_w_out != nullptr)
    for (int c = 0; c < D; ++c) acc = fmaf(a.g_w_out[ro * 5 * D + D + h * D + c], f.wr[(D + j) * D + c], acc);
  if (a.g_b_out != nullptr) acc = fmaf(a.g_b_out[ro], f.br[D + j], acc);
  a.d_wo[ro * D + j] = acc;
  if (j == 0) a.d_bo[ro] = ld0(a.g_b_out, ro);
}

}  // namespace

extern "C" int tbx_attn_fold_fwd(const float* w, const float* b, const float* wr, const float* br, const float* wo, const float* bo,
                                 float* w_in, float* b_in, float* w_kv, float* b_kv, float* bias_k, float* w_out, float* b_out, void* stream) {
  if (!w || !b || !wr || !br || !wo || !bo || !w_in || !b_in || !w_kv || !b_kv || !bias_k || !w_out || !b_out) return TBX_ERR_ARG;
  FoldArgs a{w, b, wr, br, wo, bo, w_in, b_in, w_kv, b_kv, bias_k, w_out, b_out};
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(8 * D), dim3(D), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_attn_fold_bwd(const float* w, const float* b, const float* wr, const float* br, const float* wo, const float* g_w_in,
                                 const float* g_b_in, const float* g_w_kv, const float* g_b_kv, const float* g_bias_k, const float* g_w_out,
                                 const float* g_b_out, float* d_w, float* d_b, float* d_wr, float* d_br, float* d_wo, float* d_bo, void* stream) {
  if (!w || !b || !wr || !br || !wo || !d_w || !d_b || !d_wr || !d_br || !d_wo || !d_bo) return TBX_ERR_ARG;
  FoldBwdArgs a;
  a.f = FoldArgs{w, b, wr, br, wo, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a.g_w_in = g_w_in, a.g_b_in = g_b_in, a.g_w_kv = g_w_kv, a.g_b_kv = g_b_kv, a.g_bias_k = g_bias_k, a.g_w_out = g_w_out, a.g_b_out = g_b_out;
  a.d_w = d_w, a.d_b = d_b, a.d_wr = d_wr, a.d_br = d_br, a.d_wo = d_wo, a.d_bo = d_bo;
  hipLaunchKernelGGL(fold_bwd_kernel, dim3(6 * D), dim3(D), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
