// tbx_knarpe_attn_fwd: fused KNARPE attention (see include/tbx_hip.h for the math and the layouts).
//
// One wavefront per source token, 4 tokens per 256-thread workgroup.
//   phase 1 (scores): 8 lanes per target, 8 targets per pass. The 8 lanes of a group read one full 128-B line of the
//     target's K row and of its embedding row per step (coalesced gathers), keep the query side (q, qt = W_rpe_k^T q)
//     in registers, and reduce with three xor-shuffles. Raw scores go to LDS.
//   softmax: lanes = targets (<= 128), masked -inf unless the whole row is masked (then un-masked and flagged).
//   phase 2 (weighted sums): lanes = channel pairs; V rows and embedding rows are streamed fully coalesced, the
//     probabilities are LDS broadcasts; fully-masked targets are skipped (wave-uniform branch).
// HBM-bound by construction: per pair it reads the K row, the V row, the embedding row (second pass from L2),
// 4 B of index and 1 B of mask. d_model 128, 4 heads of 32, d_rpe 128.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

constexpr int D = 128, NH = 4, DH = 32, DR = 128, KMAX = 128;

struct AttnArgs {
  const float* qbuf;
  const float* rpe_k_bias;
  float* out;
  uint8_t* row_no_valid;
  tbx_attn_seg_t seg[2];
  int ldq, q_off, qt_off, ldo, n_rows, n_src, n_seg;
  float scale;
};

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__global__ __launch_bounds__(256) void knarpe_attn_kernel(const AttnArgs a) {
  __shared__ float p_s[4][NH][KMAX];
  __shared__ uint8_t inv_s[4][KMAX];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= a.n_rows) return;
  const int b = row / a.n_src;
  const int k0 = a.seg[0].k;
  const int ktot = k0 + (a.n_seg > 1 ? a.seg[1].k : 0);
  const int s8 = lane & 7, tg = lane >> 3;

  // ---- query side in registers
  const float* qrow = a.qbuf + (int64_t)row * a.ldq;
  float4 qv[NH], qtv[NH][4];
  float qb[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
    const float4 bk = *(const float4*)(a.rpe_k_bias + h * DH + s8 * 4);
    qb[h] = tbx::group8_sum(dot4(qv[h], bk));
#pragma unroll
    for (int st = 0; st < 4; ++st) qtv[h][st] = *(const float4*)(qrow + a.qt_off + h * DR + st * 32 + s8 * 4);
  }

  // ---- phase 1: raw scores
  bool any_valid = false;
  for (int base = 0; base < ktot; base += 8) {
    const int t = base + tg;
    const bool active = t < ktot;
    const int sg = (active && t >= k0) ? 1 : 0;
    const tbx_attn_seg_t& S = a.seg[sg];
    const int kk = sg ? t - k0 : t;
    float acc[NH] = {0.f, 0.f, 0.f, 0.f};
    bool inv = true;
    if (active) {
      const int64_t pi = (int64_t)row * S.k + kk;
      const int j = S.idx[pi];
      inv = S.invalid[pi] != 0;
      const float* krow = S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt + j) * S.ld_kv + S.k_off;
      const float* erow = S.emb + pi * DR;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const float4 kq = *(const float4*)(krow + st * 32 + s8 * 4);
        const float4 e = *(const float4*)(erow + st * 32 + s8 * 4);
        acc[st] += dot4(kq, qv[st]);
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[h] += dot4(e, qtv[h][st]);
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = tbx::group8_sum(acc[h]) + qb[h];
    if (active && s8 == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h) p_s[wave][h][t] = acc[h];
      inv_s[wave][t] = inv ? 1 : 0;
    }
    any_valid = any_valid || (__ballot(active && !inv) != 0ull);
  }
  __builtin_amdgcn_wave_barrier();

  // ---- softmax over targets (lanes = targets), scale applied after masking as the reference does
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float sv[2];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      float s = -INFINITY;
      if (t < ktot && !(any_valid && inv_s[wave][t] != 0)) s = p_s[wave][h][t] * a.scale;
      sv[q] = s;
      m = fmaxf(m, s);
    }
    m = tbx::wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sv[q] = (sv[q] == -INFINITY) ? 0.f : expf(sv[q] - m);
      sum += sv[q];
    }
    sum = tbx::wave_sum(sum);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      if (t < ktot) p_s[wave][h][t] = sv[q] / sum;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- phase 2: out = [sum a v | sum a e per head], lanes = channel pairs
  const int c2 = lane * 2;
  const int myh = lane >> 4;
  float2 o = make_float2(0.f, 0.f);
  float2 eb[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) eb[h] = make_float2(0.f, 0.f);
  for (int t = 0; t < ktot; ++t) {
    const float a0 = p_s[wave][0][t], a1 = p_s[wave][1][t], a2 = p_s[wave][2][t], a3 = p_s[wave][3][t];
    if (a0 == 0.f && a1 == 0.f && a2 == 0.f && a3 == 0.f) continue;
    const int sg = t >= k0 ? 1 : 0;
    const tbx_attn_seg_t& S = a.seg[sg];
    const int64_t pi = (int64_t)row * S.k + (sg ? t - k0 : t);
    const int j = __builtin_amdgcn_readfirstlane(S.idx[pi]);
    const float* vrow = S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt + j) * S.ld_kv + S.v_off;
    const float2 v = *(const float2*)(vrow + c2);
    const float2 e = *(const float2*)(S.emb + pi * DR + c2);
    const float am = myh == 0 ? a0 : (myh == 1 ? a1 : (myh == 2 ? a2 : a3));
    o.x += am * v.x;
    o.y += am * v.y;
    eb[0].x += a0 * e.x; eb[0].y += a0 * e.y;
    eb[1].x += a1 * e.x; eb[1].y += a1 * e.y;
    eb[2].x += a2 * e.x; eb[2].y += a2 * e.y;
    eb[3].x += a3 * e.x; eb[3].y += a3 * e.y;
  }
  float* orow = a.out + (int64_t)row * a.ldo;
  *(float2*)(orow + c2) = o;
#pragma unroll
  for (int h = 0; h < NH; ++h) *(float2*)(orow + D + h * DR + c2) = eb[h];
  if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
}

}  // namespace

extern "C" int tbx_knarpe_attn_fwd(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                   int n_src, const tbx_attn_seg_t* segs, int n_seg, float* out, int ldo,
                                   uint8_t* row_no_valid, void* stream) {
  if (!qbuf || !rpe_k_bias || !segs || !out || !row_no_valid || n_batch <= 0 || n_src <= 0) return TBX_ERR_ARG;
  if (n_seg < 1 || n_seg > 2 || ldo < D + NH * DR) return TBX_ERR_UNSUPPORTED;
  if ((ldq % 4) || (q_off % 4) || (qt_off % 4) || (ldo % 4) || (((uintptr_t)qbuf) & 15) || (((uintptr_t)out) & 15) ||
      (((uintptr_t)rpe_k_bias) & 15))
    return TBX_ERR_ALIGN;
  AttnArgs a;
  int ktot = 0;
  for (int i = 0; i < n_seg; ++i) {
    const tbx_attn_seg_t& s = segs[i];
    if (!s.kv || !s.idx || !s.invalid || !s.emb || s.k <= 0 || s.n_tgt <= 0 || s.batch_div <= 0) return TBX_ERR_ARG;
    if ((s.ld_kv % 4) || (s.k_off % 4) || (s.v_off % 4) || (((uintptr_t)s.kv) & 15) || (((uintptr_t)s.emb) & 15))
      return TBX_ERR_ALIGN;
    ktot += s.k;
    a.seg[i] = s;
  }
  if (n_seg == 1) a.seg[1] = a.seg[0];
  if (ktot > KMAX) return TBX_ERR_UNSUPPORTED;
  a.qbuf = qbuf;
  a.rpe_k_bias = rpe_k_bias;
  a.out = out;
  a.row_no_valid = row_no_valid;
  a.ldq = ldq;
  a.q_off = q_off;
  a.qt_off = qt_off;
  a.ldo = ldo;
  a.n_rows = n_batch * n_src;
  a.n_src = n_src;
  a.n_seg = n_seg;
  a.scale = 1.0f / sqrtf((float)DH);
  hipLaunchKernelGGL(knarpe_attn_kernel, dim3((a.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
