// tbx_knarpe_attn_fwd: fused KNARPE attention (see include/tbx_hip.h for the math and the layouts).
//
// One wavefront per source token, 4 tokens per 256-thread workgroup (large grids), or 4 wavefronts per token (small grids).
// Forward: single pass with an online softmax (see knarpe_attn_kernel below): per pair it reads the K row, the V row and
// the embedding row exactly once (full 128-B lines per 8-lane group), 4 B of index and 1 B of mask; masked targets
// contribute nothing, rows without any valid target are written as zeros and flagged.
// d_model 128, 4 heads of 32, d_rpe 128.
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

// WPR = wavefronts cooperating on one source row: 1 for large grids (a wave per row, 4 rows per workgroup), 4 for small
// grids (the closed loop at a few scenes is latency-bound: 4 waves split a row's targets and combine through LDS).
//
// Single pass, online softmax: 8 lanes per target, 8 targets per wave per pass. The 8 lanes of a group read one full
// 128-B line of the target's K row, V row and embedding row per step (coalesced gathers) - every row exactly once - and
// keep, for THEIR target slot, a running (max, sum) per head and the un-normalised partial sums
//   O_slot[c] += p[h(c)] v[c] ,  E_slot[h][c] += p[h] e[c]      (their 16-channel slice c)
// rescaled when the slot's running max grows. The 8 slots (and the WPR waves) are merged once per row:
//   out = sum_slots exp(m_slot - M) acc_slot / sum_slots exp(m_slot - M) l_slot.
// No LDS traffic and no barrier inside the target loop.
template <int WPR>
__global__ __launch_bounds__(256) void knarpe_attn_kernel(const AttnArgs a) {
  constexpr int OUTW = D + NH * DR;  // 640
  __shared__ float red_s[WPR > 1 ? WPR : 1][WPR > 1 ? (OUTW + 2 * NH) : 1];
  constexpr int RPB = 4 / WPR;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int rib = wave / WPR;
  const int wir = wave % WPR;
  const int row = blockIdx.x * RPB + rib;
  if (row >= a.n_rows) return;  // uniform per row group (and per workgroup when WPR == 4)
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

  // ---- per-slot online softmax state and partial sums (this lane's 16-channel slice: channels st*32 + s8*4 .. +3)
  float m_run[NH], l_run[NH];
  float4 oacc[4], eacc[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    m_run[h] = -INFINITY;
    l_run[h] = 0.f;
    oacc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int st = 0; st < 4; ++st) eacc[h][st] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto scale4 = [](float4& v, float f) { v.x *= f; v.y *= f; v.z *= f; v.w *= f; };
  auto fma4 = [](float4& acc, float p, const float4 v) { acc.x += p * v.x; acc.y += p * v.y; acc.z += p * v.z; acc.w += p * v.w; };

  for (int base = wir * 8; base < ktot; base += 8 * WPR) {
    const int t = base + tg;
    const bool active = t < ktot;
    const int sg = (active && t >= k0) ? 1 : 0;
    const tbx_attn_seg_t& S = a.seg[sg];
    const int kk = sg ? t - k0 : t;
    float acc[NH] = {0.f, 0.f, 0.f, 0.f};
    bool inv = true;
    float4 v[4], e[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) v[st] = e[st] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
      const int64_t pi = (int64_t)row * S.k + kk;
      const int j = S.idx[pi];
      inv = S.invalid[pi] != 0;
      const float* trow = S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt + j) * S.ld_kv;
      const float* erow = S.emb + pi * DR;
      float4 kq[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        kq[st] = *(const float4*)(trow + S.k_off + st * 32 + s8 * 4);
        e[st] = *(const float4*)(erow + st * 32 + s8 * 4);
        v[st] = *(const float4*)(trow + S.v_off + st * 32 + s8 * 4);
      }
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        acc[st] += dot4(kq[st], qv[st]);
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[h] += dot4(e[st], qtv[h][st]);
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float sc = (tbx::group8_sum(acc[h]) + qb[h]) * a.scale;  // scale applied after masking as the reference does
      if (active && !inv) {  // uniform within the 8-lane group
        const float m_new = fmaxf(m_run[h], sc);
        const float alpha = expf(m_run[h] - m_new);  // exp(-inf) = 0 on the slot's first valid target
        const float pr = expf(sc - m_new);
        l_run[h] = l_run[h] * alpha + pr;
        m_run[h] = m_new;
        scale4(oacc[h], alpha);      // K/V channel block st == h belongs to head h
        fma4(oacc[h], pr, v[h]);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          scale4(eacc[h][st], alpha);
          fma4(eacc[h][st], pr, e[st]);
        }
      }
    }
  }

  // ---- merge the 8 target slots of this wave (lanes with equal s8: xor 8, 16, 32)
  float M[NH], L[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float mm = m_run[h];
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) mm = fmaxf(mm, __shfl_xor(mm, off, 64));
    M[h] = mm;
    const float f = (m_run[h] == -INFINITY) ? 0.f : expf(m_run[h] - mm);
    float ll = l_run[h] * f;
    scale4(oacc[h], f);
#pragma unroll
    for (int st = 0; st < 4; ++st) scale4(eacc[h][st], f);
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) ll += __shfl_xor(ll, off, 64);
    L[h] = ll;
  }
  auto red4 = [](float4 v) {
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
      v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    return v;
  };
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    oacc[h] = red4(oacc[h]);
#pragma unroll
    for (int st = 0; st < 4; ++st) eacc[h][st] = red4(eacc[h][st]);
  }
  float* orow = a.out + (int64_t)row * a.ldo;
  if constexpr (WPR == 1) {
    const bool any_valid = M[0] > -INFINITY;  // masks are per target, so every head sees the same validity
    if (tg == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float inv_l = any_valid ? 1.0f / L[h] : 0.f;
        float4 o = oacc[h];
        scale4(o, inv_l);
        *(float4*)(orow + h * DH + s8 * 4) = o;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          float4 ev = eacc[h][st];
          scale4(ev, inv_l);
          *(float4*)(orow + D + h * DR + st * 32 + s8 * 4) = ev;
        }
      }
    }
    if (lane == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
  } else {
    // per-wave (M, L, un-normalised sums) -> LDS, then all threads combine the WPR waves
    if (tg == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        *(float4*)(&red_s[wir][h * DH + s8 * 4]) = oacc[h];
#pragma unroll
        for (int st = 0; st < 4; ++st) *(float4*)(&red_s[wir][D + h * DR + st * 32 + s8 * 4]) = eacc[h][st];
      }
    }
    if (lane < NH) {
      red_s[wir][OUTW + lane] = M[lane];
      red_s[wir][OUTW + NH + lane] = L[lane];
    }
    __syncthreads();
    float Mx[NH], inv_l[NH], fw[WPR][NH];
    bool any_valid = false;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      float mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < WPR; ++w) mm = fmaxf(mm, red_s[w][OUTW + h]);
      Mx[h] = mm;
      float ll = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) {
        const float mw = red_s[w][OUTW + h];
        fw[w][h] = (mw == -INFINITY) ? 0.f : expf(mw - mm);
        ll += fw[w][h] * red_s[w][OUTW + NH + h];
      }
      any_valid = any_valid || mm > -INFINITY;
      inv_l[h] = (mm > -INFINITY) ? 1.0f / ll : 0.f;
    }
    for (int c = threadIdx.x; c < OUTW; c += 256) {
      const int h = c < D ? c / DH : (c - D) / DR;
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < WPR; ++w) acc += fw[w][h] * red_s[w][c];
      orow[c] = acc * inv_l[h];
    }
    if (threadIdx.x == 0) a.row_no_valid[row] = any_valid ? 0 : 1;
  }
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
  if (a.n_rows >= 4096)
    hipLaunchKernelGGL(knarpe_attn_kernel<1>, dim3((a.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(knarpe_attn_kernel<4>, dim3(a.n_rows), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

// =====================================================================================================================
// Backward of the fused KNARPE attention (training). Same factorised math as the forward:
//   s[h,t] = q_h.(k_h[idx_t] + bk_h) + qt_h.e_t ;  a = softmax(s * scale) (masked) ;  out = [sum_t a v_h | sum_t a e_t]
// Given dout = [dO (128) | dE (4 x 128)] per row it produces
//   dq, dqt (written to dqbuf at q_off / qt_off), dK / dV scattered with atomicAdd into the K/V-table-shaped gradient
//   of each segment, and d(rpe_k_bias) (atomicAdd, 128 floats).
// The pose embeddings carry no gradient (relative poses are computed under no_grad in the reference, utils/rpe.py:7).
// One wavefront per source row; probabilities are recomputed (nothing but the inputs is saved by the forward).
namespace {

struct AttnBwdArgs {
  AttnArgs f;            // forward arguments (qbuf, segs, ...); f.out is unused
  const float* dout;     // [rows, ldo] = dO | dE
  float* dqbuf;          // [rows, ldq]: dq at q_off, dqt at qt_off (overwritten)
  float* dkv[2];         // per segment, same [.., ld_kv] layout as seg.kv (accumulated)
  float* dbias_k;        // [128] (accumulated)
};

__global__ __launch_bounds__(256) void knarpe_attn_bwd_kernel(const AttnBwdArgs b) {
  const AttnArgs& a = b.f;
  __shared__ float p_s[4][NH][KMAX];   // probabilities a[h,t]
  __shared__ float d_s[4][NH][KMAX];   // da[h,t], then dS[h,t]
  __shared__ uint8_t inv_s[4][KMAX];
  const int lane = threadIdx.x & 63;
  const int rib = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + rib;
  if (row >= a.n_rows) return;
  const int bidx = row / a.n_src;
  const int k0 = a.seg[0].k;
  const int ktot = k0 + (a.n_seg > 1 ? a.seg[1].k : 0);
  const int s8 = lane & 7, tg = lane >> 3;
  const float* qrow = a.qbuf + (int64_t)row * a.ldq;
  float4 qv[NH], bkv[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    qv[h] = *(const float4*)(qrow + a.q_off + h * DH + s8 * 4);
    bkv[h] = *(const float4*)(a.rpe_k_bias + h * DH + s8 * 4);
  }
  // ---- recompute raw scores
  bool any_valid = false;
  {
    float4 qtv[NH][4];
    float qb[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      qb[h] = tbx::group8_sum(dot4(qv[h], bkv[h]));
#pragma unroll
      for (int st = 0; st < 4; ++st) qtv[h][st] = *(const float4*)(qrow + a.qt_off + h * DR + st * 32 + s8 * 4);
    }
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
        const float* krow = S.kv + ((int64_t)(bidx / S.batch_div) * S.n_tgt + j) * S.ld_kv + S.k_off;
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
        for (int h = 0; h < NH; ++h) p_s[rib][h][t] = acc[h];
        inv_s[rib][t] = inv ? 1 : 0;
      }
      any_valid = any_valid || (__ballot(active && !inv) != 0ull);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- softmax (probabilities back into p_s)
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float sv[2];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      float sc = -INFINITY;
      if (t < ktot && !(any_valid && inv_s[rib][t] != 0)) sc = p_s[rib][h][t] * a.scale;
      sv[q] = sc;
      m = fmaxf(m, sc);
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
      if (t < ktot) p_s[rib][h][t] = sv[q] / sum;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- pass B: da[h,t] = dO_h . v_h[idx_t] + dE_h . e_t
  const float* drow = b.dout + (int64_t)row * a.ldo;
  float4 dov[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) dov[h] = *(const float4*)(drow + h * DH + s8 * 4);
  {
    float4 dev[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int st = 0; st < 4; ++st) dev[h][st] = *(const float4*)(drow + D + h * DR + st * 32 + s8 * 4);
    for (int base = 0; base < ktot; base += 8) {
      const int t = base + tg;
      const bool active = t < ktot;
      const int sg = (active && t >= k0) ? 1 : 0;
      const tbx_attn_seg_t& S = a.seg[sg];
      const int kk = sg ? t - k0 : t;
      float acc[NH] = {0.f, 0.f, 0.f, 0.f};
      if (active) {
        const int64_t pi = (int64_t)row * S.k + kk;
        const int j = S.idx[pi];
        const float* vrow = S.kv + ((int64_t)(bidx / S.batch_div) * S.n_tgt + j) * S.ld_kv + S.v_off;
        const float* erow = S.emb + pi * DR;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const float4 v = *(const float4*)(vrow + st * 32 + s8 * 4);
          const float4 e = *(const float4*)(erow + st * 32 + s8 * 4);
          acc[st] += dot4(v, dov[st]);
#pragma unroll
          for (int h = 0; h < NH; ++h) acc[h] += dot4(e, dev[h][st]);
        }
      }
#pragma unroll
      for (int h = 0; h < NH; ++h) acc[h] = tbx::group8_sum(acc[h]);
      if (active && s8 == 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) d_s[rib][h][t] = acc[h];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- softmax backward: dS = a (da - sum_t a da) * scale
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      if (t < ktot) part += p_s[rib][h][t] * d_s[rib][h][t];
    }
    const float dotv = tbx::wave_sum(part);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = lane + 64 * q;
      if (t < ktot) d_s[rib][h][t] = p_s[rib][h][t] * (d_s[rib][h][t] - dotv) * a.scale;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- pass C: dq, dqt, d bias_k (registers, reduced over the 8 target slots at the end); dK, dV scattered
  float4 dq[NH], dbk[NH], dqt[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    dq[h] = dbk[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int st = 0; st < 4; ++st) dqt[h][st] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int base = 0; base < ktot; base += 8) {
    const int t = base + tg;
    if (t >= ktot) continue;
    const int sg = t >= k0 ? 1 : 0;
    const tbx_attn_seg_t& S = a.seg[sg];
    const int64_t pi = (int64_t)row * S.k + (sg ? t - k0 : t);
    const int j = S.idx[pi];
    const int64_t trow = ((int64_t)(bidx / S.batch_div) * S.n_tgt + j) * S.ld_kv;
    const float* krow = S.kv + trow + S.k_off;
    const float* erow = S.emb + pi * DR;
    float* dk = b.dkv[sg] + trow + S.k_off;
    float* dv = b.dkv[sg] + trow + S.v_off;
    float ds[NH], pa[NH];
    bool any = false;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      ds[h] = d_s[rib][h][t];
      pa[h] = p_s[rib][h][t];
      any = any || ds[h] != 0.f || pa[h] != 0.f;
    }
    if (!any) continue;
#pragma unroll
    for (int st = 0; st < 4; ++st) {  // st doubles as the head of the K/V channel block
      const float4 kq = *(const float4*)(krow + st * 32 + s8 * 4);
      const float4 e = *(const float4*)(erow + st * 32 + s8 * 4);
      const float g = ds[st];
      dq[st].x += g * (kq.x + bkv[st].x); dq[st].y += g * (kq.y + bkv[st].y);
      dq[st].z += g * (kq.z + bkv[st].z); dq[st].w += g * (kq.w + bkv[st].w);
      dbk[st].x += g * qv[st].x; dbk[st].y += g * qv[st].y; dbk[st].z += g * qv[st].z; dbk[st].w += g * qv[st].w;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        dqt[h][st].x += ds[h] * e.x; dqt[h][st].y += ds[h] * e.y; dqt[h][st].z += ds[h] * e.z; dqt[h][st].w += ds[h] * e.w;
      }
      const int c0 = st * 32 + s8 * 4;
      atomicAdd(dk + c0 + 0, g * qv[st].x); atomicAdd(dk + c0 + 1, g * qv[st].y);
      atomicAdd(dk + c0 + 2, g * qv[st].z); atomicAdd(dk + c0 + 3, g * qv[st].w);
      const float pv = pa[st];
      atomicAdd(dv + c0 + 0, pv * dov[st].x); atomicAdd(dv + c0 + 1, pv * dov[st].y);
      atomicAdd(dv + c0 + 2, pv * dov[st].z); atomicAdd(dv + c0 + 3, pv * dov[st].w);
    }
  }
  // reduce over the 8 target slots (lanes with equal s8)
  auto red4 = [](float4 v) {
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
      v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    return v;
  };
  float* dqrow = b.dqbuf + (int64_t)row * a.ldq;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float4 r = red4(dq[h]);
    const float4 rb = red4(dbk[h]);
    if (tg == 0) {
      *(float4*)(dqrow + a.q_off + h * DH + s8 * 4) = r;
      float* db = b.dbias_k + h * DH + s8 * 4;
      atomicAdd(db + 0, rb.x); atomicAdd(db + 1, rb.y); atomicAdd(db + 2, rb.z); atomicAdd(db + 3, rb.w);
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float4 rt = red4(dqt[h][st]);
      if (tg == 0) *(float4*)(dqrow + a.qt_off + h * DR + st * 32 + s8 * 4) = rt;
    }
  }
}

}  // namespace

extern "C" int tbx_knarpe_attn_bwd(const float* qbuf, int ldq, int q_off, int qt_off, const float* rpe_k_bias, int n_batch,
                                   int n_src, const tbx_attn_seg_t* segs, int n_seg, const float* dout, int ldo, float* dqbuf,
                                   float* const* dkv, float* dbias_k, void* stream) {
  if (!qbuf || !rpe_k_bias || !segs || !dout || !dqbuf || !dkv || !dbias_k || n_batch <= 0 || n_src <= 0) return TBX_ERR_ARG;
  if (n_seg < 1 || n_seg > 2 || ldo < D + NH * DR) return TBX_ERR_UNSUPPORTED;
  if ((ldq % 4) || (q_off % 4) || (qt_off % 4) || (ldo % 4) || (((uintptr_t)qbuf) & 15) || (((uintptr_t)dout) & 15) ||
      (((uintptr_t)dqbuf) & 15) || (((uintptr_t)rpe_k_bias) & 15))
    return TBX_ERR_ALIGN;
  AttnBwdArgs b;
  int ktot = 0;
  for (int i = 0; i < n_seg; ++i) {
    const tbx_attn_seg_t& s = segs[i];
    if (!s.kv || !s.idx || !s.invalid || !s.emb || !dkv[i] || s.k <= 0 || s.n_tgt <= 0 || s.batch_div <= 0) return TBX_ERR_ARG;
    if ((s.ld_kv % 4) || (s.k_off % 4) || (s.v_off % 4) || (((uintptr_t)s.kv) & 15) || (((uintptr_t)s.emb) & 15)) return TBX_ERR_ALIGN;
    ktot += s.k;
    b.f.seg[i] = s;
    b.dkv[i] = dkv[i];
  }
  if (n_seg == 1) {
    b.f.seg[1] = b.f.seg[0];
    b.dkv[1] = b.dkv[0];
  }
  if (ktot > KMAX) return TBX_ERR_UNSUPPORTED;
  b.f.qbuf = qbuf;
  b.f.rpe_k_bias = rpe_k_bias;
  b.f.out = nullptr;
  b.f.row_no_valid = nullptr;
  b.f.ldq = ldq;
  b.f.q_off = q_off;
  b.f.qt_off = qt_off;
  b.f.ldo = ldo;
  b.f.n_rows = n_batch * n_src;
  b.f.n_src = n_src;
  b.f.n_seg = n_seg;
  b.f.scale = 1.0f / sqrtf((float)DH);
  b.dout = dout;
  b.dqbuf = dqbuf;
  b.dbias_k = dbias_k;
  hipLaunchKernelGGL(knarpe_attn_bwd_kernel, dim3((b.f.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, b);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
