// Device code shared by the KNARPE attention kernels (attn.hip) and the fused decoder-layer kernel (dec_mid.hip): the per-lane
// channel slices, the in-register pose embedding, and the target sweep of one wavefront with its 8-slot merge. One definition =
// the same arithmetic in the same order in every kernel (bit-identical rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace tbx_attn {

constexpr int D = 128, NH = 4, DH = 32, DR = 128, KMAX = 128;

// Every multiply-add of the shared arithmetic is written out (contraction off, explicit fmaf): which product of "a*b + c*d" the
// compiler fuses is otherwise its choice per kernel, and two kernels that inline the same source would round differently.
#pragma clang fp contract(off)
__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ void scale4(float4& v, float f) { v.x *= f; v.y *= f; v.z *= f; v.w *= f; }
__device__ __forceinline__ void fma4(float4& acc, float p, const float4 v) {
  acc.x = __builtin_fmaf(p, v.x, acc.x); acc.y = __builtin_fmaf(p, v.y, acc.y);
  acc.z = __builtin_fmaf(p, v.z, acc.z); acc.w = __builtin_fmaf(p, v.w, acc.w);
}
__device__ __forceinline__ void scale2(float2& v, float f) { v.x *= f; v.y *= f; }
__device__ __forceinline__ void fma2(float2& acc, float p, const float2 v) {
  acc.x = __builtin_fmaf(p, v.x, acc.x); acc.y = __builtin_fmaf(p, v.y, acc.y);
}

// A lane's 16-channel slice of a 128-d embedding-space vector (see the header comment for the channel set).
struct ESlice {
  float2 xc, xs, yc, ys;
  float4 wc, ws;
  __device__ __forceinline__ void zero() {
    xc = xs = yc = ys = make_float2(0.f, 0.f);
    wc = ws = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void load(const float* __restrict__ p, int s8) {
    xc = *(const float2*)(p + 2 * s8);
    xs = *(const float2*)(p + 16 + 2 * s8);
    yc = *(const float2*)(p + 32 + 2 * s8);
    ys = *(const float2*)(p + 48 + 2 * s8);
    wc = *(const float4*)(p + 64 + 4 * s8);
    ws = *(const float4*)(p + 96 + 4 * s8);
  }
  __device__ __forceinline__ void store(float* __restrict__ p, int s8) const {
    *(float2*)(p + 2 * s8) = xc;
    *(float2*)(p + 16 + 2 * s8) = xs;
    *(float2*)(p + 32 + 2 * s8) = yc;
    *(float2*)(p + 48 + 2 * s8) = ys;
    *(float4*)(p + 64 + 4 * s8) = wc;
    *(float4*)(p + 96 + 4 * s8) = ws;
  }
  __device__ __forceinline__ float dot(const ESlice& o) const {
    float s = xc.x * o.xc.x;
    s = __builtin_fmaf(xc.y, o.xc.y, s); s = __builtin_fmaf(xs.x, o.xs.x, s); s = __builtin_fmaf(xs.y, o.xs.y, s);
    s = __builtin_fmaf(yc.x, o.yc.x, s); s = __builtin_fmaf(yc.y, o.yc.y, s); s = __builtin_fmaf(ys.x, o.ys.x, s);
    s = __builtin_fmaf(ys.y, o.ys.y, s);
    return (s + dot4(wc, o.wc)) + dot4(ws, o.ws);
  }
  __device__ __forceinline__ void scale(float f) {
    scale2(xc, f); scale2(xs, f); scale2(yc, f); scale2(ys, f);
    scale4(wc, f); scale4(ws, f);
  }
  __device__ __forceinline__ void fma(float p, const ESlice& o) {
    fma2(xc, p, o.xc); fma2(xs, p, o.xs); fma2(yc, p, o.yc); fma2(ys, p, o.ys);
    fma4(wc, p, o.wc); fma4(ws, p, o.ws);
  }
  __device__ __forceinline__ void reduce_slots() {  // sum over the 8 target slots (lanes with equal s8)
    using tbx::slot_sum;
    xc.x = slot_sum(xc.x); xc.y = slot_sum(xc.y); xs.x = slot_sum(xs.x); xs.y = slot_sum(xs.y);
    yc.x = slot_sum(yc.x); yc.y = slot_sum(yc.y); ys.x = slot_sum(ys.x); ys.y = slot_sum(ys.y);
    wc.x = slot_sum(wc.x); wc.y = slot_sum(wc.y); wc.z = slot_sum(wc.z); wc.w = slot_sum(wc.w);
    ws.x = slot_sum(ws.x); ws.y = slot_sum(ws.y); ws.z = slot_sum(ws.z); ws.w = slot_sum(ws.w);
  }
};

// sin / cos of an fp32 angle through the hardware v_sin_f32 / v_cos_f32 (argument in revolutions, |rev| <= 256), with a
// two-constant 1/(2 pi) and an fma-exact reduction to [-0.5, 0.5]: the reduction error is ~1e-9 rev even for the
// ~500 rad arguments of x * f_0, so the result is within the hardware's ~1e-6 absolute error of sin/cos of the SAME fp32
// product the reference feeds to torch.sin / torch.cos. ~10 instructions instead of ~100 for the libm sincosf.
__device__ __forceinline__ void sincos_rev(float arg, float* sn, float* cs) {
  constexpr float INV2PI_HI = 0.15915494f;          // float(1 / 2pi)
  constexpr float INV2PI_LO = 6.4206395e-09f;       // 1 / 2pi - INV2PI_HI
  const float n = rintf(arg * INV2PI_HI);
  float f = fmaf(arg, INV2PI_HI, -n);
  f = fmaf(arg, INV2PI_LO, f);
  *sn = __builtin_amdgcn_sinf(f);
  *cs = __builtin_amdgcn_cosf(f);
}

// Dropout mask bit of (row, global target slot t < 128, head): lowbias32 finaliser over a counter keyed by the seed.
struct DropKey {
  uint32_t lo, hi, krow;
  // row: the wave's source row (uniform); b = row / n_src
  template <class A>
  __device__ __forceinline__ void init(const A& a, int row, int b) {
    const uint64_t sd = *a.drop_seed;
    const int sc = b / a.drop_time_batch;
    const uint32_t ts = (uint32_t)(a.drop_time0 + (b - sc * a.drop_time_batch));
    krow = (uint32_t)(sc * a.n_src + (row - b * a.n_src));
    lo = (uint32_t)sd ^ (a.drop_call * 0x85EBCA6Bu) ^ (ts * 0x27D4EB2Fu);
    hi = (uint32_t)(sd >> 32) + a.drop_call * 0xC2B2AE35u + ts * 0x165667B1u;
  }
  __device__ __forceinline__ bool keep(uint32_t t, uint32_t h, uint32_t thresh) const {
    uint32_t x = ((krow * 128u + t) * 4u + h) ^ lo;
    x *= 0x9E3779B1u;
    x ^= hi;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x >= thresh;
  }
};

// The lane's frequencies for rebuilding its embedding slice from a relative pose.
struct EFreq {
  float fx[2], fw[4];
  __device__ __forceinline__ void init(const float* __restrict__ fxy, const float* __restrict__ fyaw, int s8) {
    fx[0] = fx[1] = 0.f;
    fw[0] = fw[1] = fw[2] = fw[3] = 0.f;
    if (fxy == nullptr) return;
    // the reference's buffers are repeat-interleaved [f0,f0,f1,f1,..]; the even entry serves the (cos, sin) pair
    fx[0] = fxy[2 * (2 * s8)];
    fx[1] = fxy[2 * (2 * s8 + 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) fw[i] = fyaw[2 * (4 * s8 + i)];
  }
  __device__ __forceinline__ void embed(const float* __restrict__ rel3, ESlice& e) const {
    const float x = rel3[0], y = rel3[1], w = rel3[2];
    sincos_rev(x * fx[0], &e.xs.x, &e.xc.x);
    sincos_rev(x * fx[1], &e.xs.y, &e.xc.y);
    sincos_rev(y * fx[0], &e.ys.x, &e.yc.x);
    sincos_rev(y * fx[1], &e.ys.y, &e.yc.y);
    sincos_rev(w * fw[0], &e.ws.x, &e.wc.x);
    sincos_rev(w * fw[1], &e.ws.y, &e.wc.y);
    sincos_rev(w * fw[2], &e.ws.z, &e.wc.z);
    sincos_rev(w * fw[3], &e.ws.w, &e.wc.w);
  }
};

__device__ __forceinline__ void load_e(const tbx_attn_seg_t& S, int64_t pi, int s8, const EFreq& fq, ESlice& e) {
  if (S.emb != nullptr)
    e.load(S.emb + pi * DR, s8);
  else
    fq.embed(S.rel_pose + pi * 3, e);
}

// K / V channels [c, c + 4) of a table row: fp32 tables, or bfloat16 tables widened to fp32 (a bf16 is the upper half of the float)
template <bool KV16>
__device__ __forceinline__ float4 kv_load4(const float* __restrict__ table_row, int c) {
  if constexpr (KV16) {
    const uint2 r = *(const uint2*)((const uint16_t*)table_row + c);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
  } else {
    return *(const float4*)(table_row + c);
  }
}


// q_h . k_h + qt_h . e of one (pair, head) for the lane's channel slices: 20 products as TWO interleaved fma chains (even / odd
// elements) on v_pk_fma_f32 - 10 packed instructions + one add instead of a 20-long scalar chain (the sweep is VALU-issue bound:
// profiles/r02_attn_counters.json). Shared by every kernel that forms scores (forward, ring, fused decoder layer, backward), so
// they all round alike.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pk2(const float2 v) { return (f32x2_t){v.x, v.y}; }
__device__ __forceinline__ float pair_score(const float4 k, const float4 q, const ESlice& e, const ESlice& t) {
  f32x2_t s = (f32x2_t){k.x, k.y} * (f32x2_t){q.x, q.y};
  s = __builtin_elementwise_fma((f32x2_t){k.z, k.w}, (f32x2_t){q.z, q.w}, s);
  s = __builtin_elementwise_fma(pk2(e.xc), pk2(t.xc), s);
  s = __builtin_elementwise_fma(pk2(e.xs), pk2(t.xs), s);
  s = __builtin_elementwise_fma(pk2(e.yc), pk2(t.yc), s);
  s = __builtin_elementwise_fma(pk2(e.ys), pk2(t.ys), s);
  s = __builtin_elementwise_fma((f32x2_t){e.wc.x, e.wc.y}, (f32x2_t){t.wc.x, t.wc.y}, s);
  s = __builtin_elementwise_fma((f32x2_t){e.wc.z, e.wc.w}, (f32x2_t){t.wc.z, t.wc.w}, s);
  s = __builtin_elementwise_fma((f32x2_t){e.ws.x, e.ws.y}, (f32x2_t){t.ws.x, t.ws.y}, s);
  s = __builtin_elementwise_fma((f32x2_t){e.ws.z, e.ws.w}, (f32x2_t){t.ws.z, t.ws.w}, s);
  return s[0] + s[1];
}

// Per-slot online softmax state and partial sums of one wavefront's share of a source row's targets.
struct RowAcc {
  float m_run[NH], l_run[NH];
  float4 oacc[NH];
  ESlice eacc[NH];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      m_run[h] = -INFINITY;
      l_run[h] = 0.f;
      oacc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      eacc[h].zero();
    }
  }
};

#ifdef TBX_ATTN_CLOCK
// Profiling build only (libtbx_hip_clk.so, tools/attn_clock.py): wave 0 of workgroup 0 sums the 100 MHz s_memtime ticks it spends in
// each phase of a pass: [0] index / mask / pose loads -> embedding, [1] -> scores (K rows), [2] -> softmax, [3] -> accumulate (V rows), [4] passes
extern __device__ unsigned long long g_attn_clk[8];
#define TBX_ACLK(VAR, DEP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(VAR) : "v"(DEP) : "memory")
#endif

// The target sweep of wavefront `wir` of the WPR that share source row `row` (batch entry b): segment by segment, 8 targets per
// pass, starting at target wir * 8 with stride 8 * WPR. `a` supplies seg[], n_seg, scale2 and (DROP) the dropout fields.
template <int WPR, bool DROP, bool KV16, class A>
__device__ __forceinline__ void sweep(const A& a, int row, int b, int wir, int s8, int tg, const float4 (&qv)[NH],
                                      const ESlice (&qt)[NH], const float (&qb)[NH], const EFreq& fq, RowAcc& st) {
  float(&m_run)[NH] = st.m_run;
  float(&l_run)[NH] = st.l_run;
  float4(&oacc)[NH] = st.oacc;
  ESlice(&eacc)[NH] = st.eacc;
  DropKey dk;
  if constexpr (DROP) dk.init(a, row, b);
  int t_off = 0;  // global slot of the segment's first target (the dropout counter and the backward index targets 0..ktot)
  for (int sg = 0; sg < a.n_seg; t_off += a.seg[sg].k, ++sg) {
    const tbx_attn_seg_t& S = a.seg[sg];
    // (bf16 tables: element offsets, half the bytes - the pointer is kept as float* and scaled by hand)
    constexpr int ES = KV16 ? 2 : 1;  // table elements per float slot
    const float* kvb = (const float*)((const char*)S.kv + ((int64_t)(b / S.batch_div) * S.n_tgt * S.ld_kv) * (4 / ES));
    const int64_t pbase = (int64_t)row * S.k;
    for (int base = wir * 8; base < S.k; base += 8 * WPR) {
#ifdef TBX_ATTN_CLOCK
      unsigned long long c0, c1, c2, c3, c4;
      TBX_ACLK(c0, l_run[0]);
#endif
      const int t = base + tg;
      const bool active = t < S.k;
      const int64_t pi = pbase + (active ? t : S.k - 1);
      const int j = S.idx[pi];
      const bool ok = (S.invalid[pi] == 0) & active;  // uniform within the 8-lane group (both sides evaluated: no branch)
      const float* trow = (const float*)((const char*)kvb + ((int64_t)j * S.ld_kv) * (4 / ES));
      float4 kq[4], v[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        kq[st] = kv_load4<KV16>(trow, S.k_off + st * 32 + s8 * 4);
        v[st] = kv_load4<KV16>(trow, S.v_off + st * 32 + s8 * 4);
      }
      ESlice e;
      load_e(S, pi, s8, fq, e);
      // Online softmax in the base-2 domain (scores pre-multiplied by log2(e) / sqrt(d_head), v_exp_f32 directly) with a
      // LAZY reference: a slot's reference m is its first valid score and moves only when a later score exceeds it by more
      // than 64 (2^64 headroom in fp32; a wave-uniform, practically never taken branch). The steady state has no
      // rescaling of the 80 accumulator registers: ~18 % fewer VALU instructions in a loop that is VALU-issue bound.
      float sc[NH];
      bool jump = false;
#ifdef TBX_ATTN_CLOCK
      TBX_ACLK(c1, ((e.xc.x + e.xs.y) + (e.yc.x + e.ys.y)) + ((e.wc.x + e.wc.w) + (e.ws.y + e.ws.z)));
#endif
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        sc[h] = (tbx::group8_sum(pair_score(kq[h], qv[h], e, qt[h])) + qb[h]) * a.scale2;  // scaled after masking as the reference does
        jump = jump || (ok && m_run[h] > -INFINITY && sc[h] - m_run[h] > 64.f);
      }
#ifdef TBX_ATTN_CLOCK
      TBX_ACLK(c2, (sc[0] + sc[1]) + (sc[2] + sc[3]));
#endif
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
#ifdef TBX_ATTN_CLOCK
      float prs = 0.f;
#pragma unroll
      for (int h = 0; h < NH; ++h) prs += ok ? __builtin_amdgcn_exp2f(sc[h] - ((ok && m_run[h] == -INFINITY) ? sc[h] : m_run[h])) : 0.f;
      TBX_ACLK(c3, prs);
#endif
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        m_run[h] = (ok && m_run[h] == -INFINITY) ? sc[h] : m_run[h];  // the slot's first valid target sets the reference
        const float pr = ok ? __builtin_amdgcn_exp2f(sc[h] - m_run[h]) : 0.f;
        l_run[h] += pr;  // the normaliser is that of the un-dropped softmax
        float pd = pr;
        if constexpr (DROP) pd = dk.keep((uint32_t)(t_off + t), (uint32_t)h, a.drop_thresh) ? pr * a.drop_scale : 0.f;
        fma4(oacc[h], pd, v[h]);  // K/V channel block st == h belongs to head h
        eacc[h].fma(pd, e);
      }
#ifdef TBX_ATTN_CLOCK
      TBX_ACLK(c4, ((oacc[0].x + oacc[1].y) + (oacc[2].z + oacc[3].w)) + ((eacc[0].xc.x + eacc[1].ws.w) + (eacc[2].yc.y + eacc[3].wc.z)));
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_attn_clk[0] += c1 - c0, g_attn_clk[1] += c2 - c1, g_attn_clk[2] += c3 - c2, g_attn_clk[3] += c4 - c3, g_attn_clk[4] += 1;
      }
#endif
    }
  }

}

// Merge of the wavefront's 8 target slots (lanes with equal s8: xor 8, 16, 32): on return every lane holds, for its channel
// slices, the wave's un-normalised sums relative to the wave's maximum M[h], and the wave's normaliser L[h].
__device__ __forceinline__ void merge_slots(RowAcc& st, float (&M)[NH], float (&L)[NH]) {
  float(&m_run)[NH] = st.m_run;
  float(&l_run)[NH] = st.l_run;
  float4(&oacc)[NH] = st.oacc;
  ESlice(&eacc)[NH] = st.eacc;
  // ---- merge the 8 target slots of this wave (lanes with equal s8: xor 8, 16, 32)
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float mm = tbx::slot_max(m_run[h]);
    M[h] = mm;
    const float f = (m_run[h] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[h] - mm);
    float ll = l_run[h] * f;
    scale4(oacc[h], f);
    eacc[h].scale(f);
    ll = tbx::slot_sum(ll);
    L[h] = ll;
    oacc[h].x = tbx::slot_sum(oacc[h].x); oacc[h].y = tbx::slot_sum(oacc[h].y);
    oacc[h].z = tbx::slot_sum(oacc[h].z); oacc[h].w = tbx::slot_sum(oacc[h].w);
    eacc[h].reduce_slots();
  }
}

// merge_slots with the result BY HEAD: on return lane l holds, for head l >> 4 and its channel slice s8 = l & 7, the wave's
// un-normalised sums `o` (4 value channels) / `e` (16 embedding channels) relative to the wave's maximum, that head's maximum `Mh` and
// normaliser `Lh` (lanes l and l ^ 8 hold the same numbers); M[h] = every head's maximum, wave-uniform. Bit-identical to merge_slots
// (tbx::slot_sum4 is the same butterfly) at a quarter of its instructions: the merge was ~1,300 of the ~2,700 vector instructions a
// wave spends on a row at the WOSAC shape, and 2 x ~2 us of the one-launch decoder layer's 27.
__device__ __forceinline__ void merge_slots_by_head(RowAcc& st, float (&M)[NH], float& Mh, float& Lh, float4& o, ESlice& e) {
  static_assert(NH == 4, "four heads: one per quarter of the wavefront");
  float(&m)[NH] = st.m_run;
  Mh = tbx::slot_max4(m[0], m[1], m[2], m[3]);
  float f[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    M[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Mh), 16 * h));
    f[h] = (m[h] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m[h] - M[h]);
    st.l_run[h] *= f[h];
    scale4(st.oacc[h], f[h]);
    st.eacc[h].scale(f[h]);
  }
  const float4(&a)[NH] = st.oacc;
  const ESlice(&b)[NH] = st.eacc;
  Lh = tbx::slot_sum4(st.l_run[0], st.l_run[1], st.l_run[2], st.l_run[3]);
#define TBX_S4(F) tbx::slot_sum4(a[0].F, a[1].F, a[2].F, a[3].F)
  o.x = TBX_S4(x), o.y = TBX_S4(y), o.z = TBX_S4(z), o.w = TBX_S4(w);
#undef TBX_S4
#define TBX_S4(F) tbx::slot_sum4(b[0].F, b[1].F, b[2].F, b[3].F)
  e.xc.x = TBX_S4(xc.x), e.xc.y = TBX_S4(xc.y), e.xs.x = TBX_S4(xs.x), e.xs.y = TBX_S4(xs.y);
  e.yc.x = TBX_S4(yc.x), e.yc.y = TBX_S4(yc.y), e.ys.x = TBX_S4(ys.x), e.ys.y = TBX_S4(ys.y);
  e.wc.x = TBX_S4(wc.x), e.wc.y = TBX_S4(wc.y), e.wc.z = TBX_S4(wc.z), e.wc.w = TBX_S4(wc.w);
  e.ws.x = TBX_S4(ws.x), e.ws.y = TBX_S4(ws.y), e.ws.z = TBX_S4(ws.z), e.ws.w = TBX_S4(ws.w);
#undef TBX_S4
}

// 1 KiB per wave instruction from global memory straight into LDS (global_load_lds_dwordx4), as inline asm: hipcc orders every
// later ds_read behind a DMA it knows about; this form is invisible to its counters (which can then only over-wait: VMEM returns
// in order) and the LDS hazard is the caller's: every consumer sits behind "s_waitcnt vmcnt(0)" + a workgroup barrier
// (cdna_hip_programming.md 5.7: M0 is written in the statement that reads it and restored).
__device__ __forceinline__ void glds_1k(const float* gsrc_lane, uint32_t lds_byte_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_byte_addr)
               : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p);
}

// One output of a LINEAR stage as the k-ordered fma chain of the packed MFMA path: column `c` of the 128-column block of a
// tbx_pack_weight_gemv image at `blk` in LDS (row 0 = bias, row 1 + kb*4 + t = weights of k = kb*16 + {0,4,8,12} + t), inputs
// x[0 .. kblocks*16) in LDS, 16-byte aligned: the k-block's 16 activations come as 4 broadcast ds_read_b128 (element [g*4 + t]
// multiplies W[c][kb*16 + g*4 + t]) - one LDS instruction per 2 fmas instead of 5 per 4.
__device__ __forceinline__ float gemv_chain(const float* blk, int c, const float* x, int kblocks, float acc) {
  const float4* wq = (const float4*)blk + c;
  const float4* x4 = (const float4*)__builtin_assume_aligned(x, 16);
  // software pipeline: the 8 operand reads of k-block kb + 1 go out before the 16 dependent fmas of k-block kb
#define TBX_RD(X, W, KB)                                                                                       \
  X[0] = x4[(KB) * 4], X[1] = x4[(KB) * 4 + 1], X[2] = x4[(KB) * 4 + 2], X[3] = x4[(KB) * 4 + 3];                   \
  W[0] = wq[(1 + (KB) * 4) * D], W[1] = wq[(2 + (KB) * 4) * D], W[2] = wq[(3 + (KB) * 4) * D], W[3] = wq[(4 + (KB) * 4) * D]
#define TBX_FM(X, W)                                                                                                                      \
  acc = __builtin_fmaf(X[0].x, W[0].x, acc); acc = __builtin_fmaf(X[1].x, W[0].y, acc); acc = __builtin_fmaf(X[2].x, W[0].z, acc); acc = __builtin_fmaf(X[3].x, W[0].w, acc); \
  acc = __builtin_fmaf(X[0].y, W[1].x, acc); acc = __builtin_fmaf(X[1].y, W[1].y, acc); acc = __builtin_fmaf(X[2].y, W[1].z, acc); acc = __builtin_fmaf(X[3].y, W[1].w, acc); \
  acc = __builtin_fmaf(X[0].z, W[2].x, acc); acc = __builtin_fmaf(X[1].z, W[2].y, acc); acc = __builtin_fmaf(X[2].z, W[2].z, acc); acc = __builtin_fmaf(X[3].z, W[2].w, acc); \
  acc = __builtin_fmaf(X[0].w, W[3].x, acc); acc = __builtin_fmaf(X[1].w, W[3].y, acc); acc = __builtin_fmaf(X[2].w, W[3].z, acc); acc = __builtin_fmaf(X[3].w, W[3].w, acc)
  float4 xa[4], wa[4], xb[4], wb[4];
  TBX_RD(xa, wa, 0);
  for (int kb = 0; kb < kblocks; kb += 2) {  // kblocks is even (2 or 8) at every call site
    TBX_RD(xb, wb, kb + 1);
    TBX_FM(xa, wa);
    if (kb + 2 < kblocks) { TBX_RD(xa, wa, kb + 2); }
    TBX_FM(xb, wb);
  }
#undef TBX_RD
#undef TBX_FM
  return acc;
}

#pragma clang fp contract(fast)

}  // namespace tbx_attn
