// tbx_layer_tile: the row-local half of a transformer layer for LARGE launches (>= ~1000 rows), as ONE straight-line kernel per
// chain of the three-launch schedule [attention -> chain -> attention -> chain] instead of a tbx_rowchain program:
//   ATTN  x += rows without a valid target ? 0 : out_proj(sum a v + W_rpe_v (sum a e) + b)          attention_rpe.py:152,182-190
//   FFN   x += linear2(relu(linear1(norm2 x))); x[invalid] = 0                                     transformer_rpe.py:234-237
//   PROJ  q [| k | v] = in_proj(norm x), qt_h = W_rpe_k,h^T q_h of the NEXT attention call           attention_rpe.py:92-98,147
// Why a kernel of its own: at 4096 rows a chain launch is 256 tiles of 16 rows = one workgroup per CU, every stage
// a dependent step; the interpreter spends 2.9 us per 128x128 stage (1.3 us of it decode / epilogue / barrier, 1.6 us in 32
// dependent exact-fp32 MFMAs) where the stage's weights (64 KiB per workgroup through a 64 B/clk L2 port: ~0.43 us) are the only
// thing that cannot be avoided. Here:
//   * LINEAR = split-bf16 on v_mfma_f32_16x16x32_bf16: x = x_hi + x_lo, w = w_hi + w_lo (bf16, RNE), three products hi*hi + hi*lo
//     + lo*hi with fp32 accumulation (error < 3e-5 of sum |x||w|: lo*lo and the second-order residue are dropped) - 12 matrix
//     instructions of ~17 cycles per 16x16x128 tile instead of 32 of 32-40 cycles;
//   * activations are split ONCE, by whoever produces them (load, LayerNorm, a LINEAR epilogue), into bf16 hi / lo planes in LDS,
//     laid out so that an MFMA operand is one conflict-free ds_read_b128;
//   * the products are formed TRANSPOSED (D = W x^T: A operand = 16 output channels x 32 k of the weight image, B operand =
//     32 k x 16 rows of activations): a lane ends up with 4 consecutive output channels of ONE row - one 8-byte LDS write per plane,
//     one 16-byte global store - instead of 4 scattered elements;
//   * weights come as a per-wave stream of 8 KiB units (tbx_pack_weight_mfma32: the fragments of [16 channels x 128 k] in register
//     order), double-buffered in registers one unit ahead across stage boundaries;
//   * no program to decode: the stage list is the template instantiation.
#include <atomic>

#include "tile_core.h"

using namespace tbx_tile;

namespace {

constexpr int ROWS = 16;
// Two plane pairs of different widths: the wide one takes whatever is wider than a token row - the attention output (K = 640: sum a v
// | sum a e of 4 heads), the FFN's hidden row (K = 512) and q as the A side of the qt fold; the narrow one the 128-wide rows (the
// folded value rows, the LayerNorm outputs). 69 KiB of LDS per workgroup (43 in the one-product build) instead of the 97 of two wide
// pairs: TWO workgroups per CU, so a CU whose workgroup waits (weights, a barrier, its rows) has another one to run.
typedef Planes<ROWS, 20> PA;
typedef Planes<ROWS, 4> PB;
constexpr int PLANE_A = PA::PLANE, PLANE_B = PB::PLANE;
constexpr int NPL = TBX_TILE_SINGLE ? 1 : 2;  // planes per pair: hi | lo, or the one bf16 plane
constexpr int XLD = 132;      // floats per row of the fp32 buffers X (token rows) and Y (sum a v)
constexpr size_t LDS_BYTES = 2 * ROWS * XLD * sizeof(float) + NPL * (PLANE_A + PLANE_B);

struct TileArgs {
  tbx_layer_tile_t t;
  Entry ent[16];
};

// LayerNorm_128 of row X[r] -> planes (k = 0..127), a wavefront per row, in rowchain.hip's ln_row order (bit-identical values
// before the split)
template <class PL>
__device__ __forceinline__ void ln_to_planes(const float* X, char* P, int r, int lane, const float* gamma, const float* beta, float eps) {
  float v[2], gm[2], bt[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    v[q] = X[r * XLD + lane + 64 * q];
    gm[q] = *(const TBX_GLOBAL float*)(gamma + lane + 64 * q);
    bt[q] = *(const TBX_GLOBAL float*)(beta + lane + 64 * q);
  }
  const float mean = tbx::wave_sum(v[0] + v[1]) / 128.f;
  const float d0 = v[0] - mean, d1 = v[1] - mean;
  const float var = tbx::wave_sum(d0 * d0 + d1 * d1) / 128.f;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float y = (v[q] - mean) * rstd * gm[q] + bt[q];
    const __bf16 h = (__bf16)y;
    const int o = PL::off(r, lane + 64 * q);
    *(__bf16*)(P + o) = h;
#if !TBX_TILE_SINGLE
    *(__bf16*)(P + PL::PLANE + o) = (__bf16)(y - (float)h);
#endif
  }
}

// Depth of the per-wave weight ring (register slots of one 8 KiB unit; RING - 1 units are in flight ahead of the stage that
// multiplies them). Three slots are what a lone workgroup per CU wants (launches of up to one tile per CU: the prefetch is the only
// thing that hides a unit's latency). With two slots every instantiation of the three-product build fits 128 VGPRs = 4 waves per SIMD
// = TWO workgroups per CU, which is what launches of more tiles than CUs want (the other workgroup hides the latency; measured in
// round 5, profiles/MEASUREMENT_LOG.md: +8.6 % at 64 scenes, +4.5 % at the submission shape, -6 % at 16 scenes if used everywhere).
// The one-product build fits 128 VGPRs with three slots and always uses them.
constexpr int RING_DEEP = 3, RING_PAIR = TBX_TILE_SINGLE ? 3 : 2;

// profiling build (make clk): the LAST workgroup's wave 0 stamps the shader clock at the phase boundaries of every launch
#ifdef TBX_STAGE_CLOCK
__device__ unsigned long long g_tl_clk[256 * 16];
__device__ unsigned int g_tl_launch;
#define TL_CLK(i)                                                                                      \
  do {                                                                                                 \
    if (tl_slot < 256u) g_tl_clk[tl_slot * 16 + (i)] = clock64();                                      \
  } while (0)
#else
#define TL_CLK(i)
#endif

template <bool ATTN, bool FFN, int PROJ, int RING>
__global__ __launch_bounds__(NT) void tile_layer_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;
  float* Y = X + ROWS * XLD;
  char* Pa = (char*)(Y + ROWS * XLD);
  char* Pb = Pa + NPL * PLANE_A;  // (Pa: the wide pair, Pb: the narrow one)
  const tbx_layer_tile_t& t = a.t;
  if constexpr (!ATTN && !FFN && PROJ == 2) {
    const int main_tiles = (int)((t.n_rows + ROWS - 1) / ROWS);
    if ((int)blockIdx.x >= main_tiles) {  // (only launched with rider_rows > 0)
      rider_tile<PB, XLD>(t, (int)blockIdx.x - main_tiles, X, Pb, Pa);  // (its 128-wide stages: two narrow pairs)
      return;
    }
  }
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
#ifdef TBX_STAGE_CLOCK
  unsigned tl_slot = 0xffffffffu;
  if (blockIdx.x == 0 && tid == 0) {
    tl_slot = atomicAdd(&g_tl_launch, 1u);
    if (tl_slot < 256u) g_tl_clk[tl_slot * 16 + 15] = (unsigned long long)((ATTN ? 100 : 0) + (FFN ? 10 : 0) + PROJ);  // which instantiation
  }
#endif
  TL_CLK(0);
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nv = (t.n_rows - row0) < ROWS ? (int)(t.n_rows - row0) : ROWS;
  const bool row_ok = j < nv;
  const int64_t grow = row0 + (row_ok ? j : 0);
  const int aoffA = PA::lane_off(lane, 0), aoffB = PB::lane_off(lane, 0);  // the lane's (octet, row) offset inside a plane
  const int c_out = 16 * wave + 4 * g;  // the lane's 4 output channels of a 128-wide stage

  constexpr int E_FOLD = 0, E_OUT = 1, E_L1 = ATTN ? 2 : 0, E_L2 = E_L1 + 4, E_Q = (ATTN ? 2 : 0) + (FFN ? 8 : 0);
  constexpr int E_KV = E_Q + 1, E_QF = E_Q + (PROJ == 2 ? 3 : 1), E_END = E_Q + (PROJ == 2 ? 4 : (PROJ == 1 ? 2 : 0));
  // weight units through a ring of RING register slots, RING - 1 units ahead of the stage that multiplies them. The ring is primed
  // BEHIND the requests for the tile's own rows (memory answers a wave in order: 64 KiB of weights per unit in front of them delayed
  // every stage of the launch)
  W wb[RING];
#define TBX_NEXT(E)                                                          \
  do {                                                                       \
    if constexpr ((E) + RING - 1 < E_END) load_unit(wb[((E) + RING - 1) % RING], a.ent[(E) + RING - 1], wave, lane); \
  } while (0)
#define TBX_PRIME()                                                          \
  do {                                                                       \
    _Pragma("unroll") for (int u = 0; u < RING - 1; ++u)                     \
      if (u < E_END) load_unit(wb[u], a.ent[u], wave, lane);                 \
  } while (0)

  uint8_t f_nov = 0, f_inv = 0;
  if (ATTN) f_nov = *(const TBX_GLOBAL uint8_t*)(t.row_no_valid + grow);
  if (FFN && t.src_invalid != nullptr) f_inv = *(const TBX_GLOBAL uint8_t*)(t.src_invalid + grow);

  // ---- the tile's token rows (and the attention output, split into planes; its first 128 columns also as fp32: the fold's addend)
  const int xr = tid >> 5, xc4 = tid & 31;
  f32x4 xv0 = {0.f, 0.f, 0.f, 0.f};
  if (xr < nv) xv0 = gld4(t.x + (row0 + xr) * D + xc4 * 4);
  if constexpr (!ATTN) {
    TBX_PRIME();
    *(f32x4*)(X + xr * XLD + xc4 * 4) = xv0;
  }
  if constexpr (ATTN) {
    f32x4 v[5];
    int rr[5], cc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int f = tid + NT * i;  // 16 rows x 160 float4
      rr[i] = f / 160;
      cc[i] = f - rr[i] * 160;
      v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rr[i] < nv) v[i] = gld4(t.attn_out + (row0 + rr[i]) * (int64_t)t.ld_attn + cc[i] * 4);
    }
    TBX_PRIME();
    *(f32x4*)(X + xr * XLD + xc4 * 4) = xv0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      planes_write4<PA>(Pa, rr[i], cc[i] * 4, v[i]);
      if (cc[i] < 32) *(f32x4*)(Y + rr[i] * XLD + cc[i] * 4) = v[i];
    }
  }
  __syncthreads();
  TL_CLK(1);

  if constexpr (ATTN) {
    {  // value half of linear_rpe: y_h = (sum a v)_h + W_rpe_v,h (sum a e)_h + b_h; wave w = head w / 2, 16 of its 32 channels
      TBX_NEXT(E_FOLD);
      const W& w = wb[E_FOLD % RING];
      Acc acc;
      acc.zero();
      const int step0 = 4 + 4 * (wave >> 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE_A>(acc, w.hi[s], w.lo[s], Pa + aoffA, step0 + s);
      const f32x4 y = acc.sum() + w.bias + *(const f32x4*)(Y + j * XLD + c_out);
      planes_write4<PB>(Pb, j, c_out, y);
    }
    __syncthreads();
    TL_CLK(2);
    {  // x += row without a valid target ? 0 : out_proj(y)
      TBX_NEXT(E_OUT);
      const W& w = wb[E_OUT % RING];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE_B>(acc, w.hi[s], w.lo[s], Pb + aoffB, s);
      f32x4 xv = *(const f32x4*)(X + j * XLD + c_out);
      f32x4 upd = acc.sum() + w.bias;
      if (t.drop_thresh != 0u && t.drop_site[0] >= 0) {
        DropKey4 dk;
        dk.init(t.drop_seed, (uint32_t)t.drop_site[0], (uint32_t)t.drop_step, t.drop_thresh, t.drop_scale);
        upd = dk.apply(upd, row0 + j, c_out, D);
      }
      if (!f_nov) xv += upd;
      *(f32x4*)(X + j * XLD + c_out) = xv;
      if (!FFN && t.store_x && row_ok) gst4(t.x + grow * D + c_out, xv);
    }
    __syncthreads();
    TL_CLK(3);
  }

  if constexpr (FFN) {
#pragma unroll
    for (int q = 0; q < 2; ++q) ln_to_planes<PB>(X, Pb, wave * 2 + q, lane, t.norm2_weight, t.norm2_bias, t.norm2_eps);
    __syncthreads();
    TL_CLK(4);
    const bool drop_h = t.drop_thresh != 0u && t.drop_site[1] >= 0;
    DropKey4 dkh;
    if (drop_h) dkh.init(t.drop_seed, (uint32_t)t.drop_site[1], (uint32_t)t.drop_step, t.drop_thresh, t.drop_scale);
    // h = relu(linear1(.)): 4 rounds of 128 channels
#define TBX_L1(R)                                                                               \
  do {                                                                                          \
    TBX_NEXT(E_L1 + (R));                                                                       \
    const W& w = wb[(E_L1 + (R)) % RING];                                                          \
    Acc acc;                                                                                    \
    acc.zero();                                                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE_B>(acc, w.hi[s], w.lo[s], Pb + aoffB, s); \
    f32x4 h = relu4(acc.sum() + w.bias);                                                        \
    if (drop_h) h = dkh.apply(h, row0 + j, (R) * D + c_out, 4 * D);                             \
    planes_write4<PA>(Pa, j, (R) * D + c_out, h);                                                   \
  } while (0)
#ifdef TBX_STAGE_CLOCK  // (stamps 10..14: linear1's first unit taken apart - its weights' arrival, the next unit's load latency, MFMAs, epilogue)
    {
      TL_CLK(10);
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) (lgkmcnt / expcnt free): this unit's weights are here
      TL_CLK(11);
      TBX_NEXT(E_L1);
      __builtin_amdgcn_s_waitcnt(0x0070);  // ... and the next unit's: one unit's load latency, nothing else in flight
      TL_CLK(12);
      const W& w = wb[E_L1 % RING];
      Acc acc;
      acc.zero();
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE_B>(acc, w.hi[s], w.lo[s], Pb + aoffB, s);
      f32x4 h = relu4(acc.sum() + w.bias);
      if (tl_slot < 256u) g_tl_clk[tl_slot * 16 + 13] = clock64() + (unsigned long long)(h[0] != h[0]);  // (after the MFMAs' results)
      if (drop_h) h = dkh.apply(h, row0 + j, c_out, 4 * D);
      planes_write4<PA>(Pa, j, c_out, h);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the planes are written
      TL_CLK(14);
    }
#else
    TBX_L1(0);
#endif
    TBX_L1(1);
    TBX_L1(2);
    TBX_L1(3);
#undef TBX_L1
    __syncthreads();
    TL_CLK(5);
    {  // x += linear2(h): K = 512 in 4 units into one accumulator triple; x[invalid] = 0
      Acc acc;
      acc.zero();
      f32x4 bias;
#define TBX_L2(R)                                                                                         \
  do {                                                                                                    \
    TBX_NEXT(E_L2 + (R));                                                                                 \
    const W& w = wb[(E_L2 + (R)) % RING];                                                                    \
    if ((R) == 0) bias = w.bias;                                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE_A>(acc, w.hi[s], w.lo[s], Pa + aoffA, 4 * (R) + s); \
  } while (0)
      TBX_L2(0);
      TBX_L2(1);
      TBX_L2(2);
      TBX_L2(3);
#undef TBX_L2
      f32x4 upd = acc.sum() + bias;
      if (t.drop_thresh != 0u && t.drop_site[2] >= 0) {
        DropKey4 dk;
        dk.init(t.drop_seed, (uint32_t)t.drop_site[2], (uint32_t)t.drop_step, t.drop_thresh, t.drop_scale);
        upd = dk.apply(upd, row0 + j, c_out, D);
      }
      f32x4 xv = *(const f32x4*)(X + j * XLD + c_out) + upd;
      if (f_inv) xv = (f32x4){0.f, 0.f, 0.f, 0.f};
      *(f32x4*)(X + j * XLD + c_out) = xv;
      if (t.store_x && row_ok) gst4(t.x + grow * D + c_out, xv);
    }
    __syncthreads();
    TL_CLK(6);
  }

  if constexpr (PROJ != 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) ln_to_planes<PB>(X, Pb, wave * 2 + q, lane, t.proj_norm_weight, t.proj_norm_bias, t.proj_norm_eps);
    __syncthreads();
    TL_CLK(7);
    {  // q
      TBX_NEXT(E_Q);
      const W& w = wb[E_Q % RING];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE_B>(acc, w.hi[s], w.lo[s], Pb + aoffB, s);
      const f32x4 q = acc.sum() + w.bias;
      planes_write4<PA>(Pa, j, c_out, q);
      if (row_ok) gst4(t.proj_out + grow * (int64_t)t.ld_proj + c_out, q);
    }
    if constexpr (PROJ == 2) {  // k | v: straight to the table (fp32 columns [128, 384) of proj_out, or the bfloat16 table)
#define TBX_KV(R)                                                                               \
  do {                                                                                          \
    TBX_NEXT(E_KV + (R));                                                                       \
    const W& w = wb[(E_KV + (R)) % RING];                                                          \
    Acc acc;                                                                                    \
    acc.zero();                                                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE_B>(acc, w.hi[s], w.lo[s], Pb + aoffB, s); \
    const f32x4 kv = acc.sum() + w.bias;                                                        \
    if (row_ok) {                                                                               \
      if (t.kv16_out != nullptr) {                                                              \
        const bf16x4 h = __builtin_convertvector(kv, bf16x4);                                   \
        *(TBX_GLOBAL u32x2*)((TBX_GLOBAL uint16_t*)t.kv16_out + grow * 256 + (R) * D + c_out) = __builtin_bit_cast(u32x2, h); \
      } else {                                                                                  \
        gst4(t.proj_out + grow * (int64_t)t.ld_proj + D + (R) * D + c_out, kv);                 \
      }                                                                                         \
    }                                                                                           \
  } while (0)
      TBX_KV(0);
      TBX_KV(1);
#undef TBX_KV
    }
    __syncthreads();
    TL_CLK(8);
    {  // qt_h = W_rpe_k,h^T q_h: wave w = head w / 2, 4 of its 8 tiles of 16 channels, K = 32 (one step, the head's own)
      TBX_NEXT(E_QF);
      const W& w = wb[E_QF % RING];
      const int h = wave >> 1;
      const int qt_off = PROJ == 2 ? 3 * D : D;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Acc acc;
        acc.zero();
        mfma_step<PLANE_A>(acc, w.hi[s], w.lo[s], Pa + aoffA, h);
        const f32x4 v = acc.sum();
        if (row_ok) gst4(t.proj_out + grow * (int64_t)t.ld_proj + qt_off + h * D + ((wave & 1) * 4 + s) * 16 + 4 * g, v);
      }
    }
  }
#ifdef TBX_STAGE_CLOCK
  __builtin_amdgcn_s_waitcnt(0);  // the launch's stores have left
#endif
  TL_CLK(9);
#undef TBX_NEXT
#undef TBX_PRIME
}

// tbx_pack_weight_mfma32 image of W_g [n x k] (g < groups; [k x n] with wt): T = groups * n / 16 tiles of 16 output channels, a
// unit = 4 groups of one (tile, 32-k step) each + the 4 groups' tile biases:
//   k % 128 == 0: unit u = (tile u % T, k-chunk u / T of 128): group s = k-step 4 * chunk + s of that tile;
//   k == 64:      unit u = tiles 2u, 2u + 1: group s = (tile 2u + (s >> 1), step s & 1);
//   k == 32:      unit u = tiles 4u .. 4u + 3: group s = tile 4u + s (its one k-step).
// A group = [64 lanes x 8 bf16 hi][64 lanes x 8 bf16 lo]: lane l, element e = W[tile * 16 + (l & 15)][step * 32 + (l >> 4) * 8 + e];
// tiles past T (a last partial unit) are zero.
__device__ __forceinline__ void unit_group(int k, int T, int64_t u, int s, int& tile, int& step) {
  if (k == 32) {
    tile = (int)u * 4 + s, step = 0;
  } else if (k == 64) {
    tile = (int)u * 2 + (s >> 1), step = s & 1;
  } else {
    tile = (int)(u % T), step = 4 * (int)(u / T) + s;
  }
}

__device__ __forceinline__ void pack_mfma32_body(const float* __restrict__ w, const float* __restrict__ bias, int n, int k, int ld, int groups, int wt,
                                                 float* __restrict__ out, int64_t total) {
  const int T = groups * n / 16;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = e / UNIT;
    const int f = (int)(e - u * UNIT);
    int tile, step;
    if (f >= 2048) {  // bias floats: 16 per group
      const int s = (f - 2048) >> 4, i = (f - 2048) & 15;
      unit_group(k, T, u, s, tile, step);
      out[e] = (bias != nullptr && tile < T) ? bias[tile * 16 + i] : 0.f;
      continue;
    }
    // float slot f of the unit's 8 KiB: group s, half (hi / lo), lane l, dword q of the lane's 16 bytes (elements 2q, 2q + 1)
    const int s = f >> 9, half = (f >> 8) & 1, l = (f >> 2) & 63, q = f & 3;
    unit_group(k, T, u, s, tile, step);
    uint32_t bits = 0u;
    if (tile < T) {
      const int oc = tile * 16 + (l & 15);
      const int grp = oc / n, col = oc - grp * n;
#pragma unroll
      for (int z = 0; z < 2; ++z) {
        const int kk = step * 32 + (l >> 4) * 8 + 2 * q + z;
        const float v = wt ? w[((int64_t)grp * k + kk) * ld + col] : w[((int64_t)grp * n + col) * ld + kk];
        const __bf16 hi = (__bf16)v;
        const __bf16 x = half ? (__bf16)(v - (float)hi) : hi;
        bits |= (uint32_t)__builtin_bit_cast(unsigned short, x) << (16 * z);
      }
    }
    out[e] = __uint_as_float(bits);
  }
}

__global__ void pack_mfma32_kernel(const float* __restrict__ w, const float* __restrict__ bias, int n, int k, int ld, int groups, int wt,
                                   float* __restrict__ out, int64_t total) {
  pack_mfma32_body(w, bias, n, k, ld, groups, wt, out, total);
}

// tbx_pack_weight_mfma32_multi: blockIdx.y = the job (a training step packs ~300 images of at most 640 x 128 each: one launch per
// group of them instead of one each)
constexpr int PACK_MULTI_MAX = 48;
struct PackMultiArgs {
  tbx_pack_job_t job[PACK_MULTI_MAX];
  int64_t total[PACK_MULTI_MAX];
};
__global__ void pack_mfma32_multi_kernel(const PackMultiArgs a) {
  const tbx_pack_job_t& j = a.job[blockIdx.y];
  pack_mfma32_body(j.w, j.bias, j.n, j.k, j.ld, j.groups, j.wt, j.out, a.total[blockIdx.y]);
}

int64_t mfma32_units(int n, int k, int groups) {
  if (n <= 0 || k <= 0 || groups <= 0 || n % 16 != 0) return TBX_ERR_UNSUPPORTED;
  const int64_t T = (int64_t)groups * n / 16;
  if (k == 32) return (T + 3) / 4;
  if (k == 64) return (T + 1) / 2;
  if (k % 128 != 0) return TBX_ERR_UNSUPPORTED;
  return T * (k / 128);
}

template <bool ATTN, bool FFN, int PROJ, int RING>
int launch_ring(const TileArgs& a, unsigned grid, hipStream_t s) {
  static tbx::PerDeviceOnce lds_attr;  // (per device, thread-safe: tbx_common.h)
  if (!lds_attr([&] { return !(hipFuncSetAttribute((const void*)tile_layer_kernel<ATTN, FFN, PROJ, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess); })) return TBX_ERR_LAUNCH;
  hipLaunchKernelGGL((tile_layer_kernel<ATTN, FFN, PROJ, RING>), dim3(grid), dim3(NT), LDS_BYTES, s, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

// the device's CU count (cached per device ordinal): a launch of at least that many tiles takes the two-workgroups-per-CU form
int cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

template <bool ATTN, bool FFN, int PROJ>
int launch(const TileArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.t.n_rows + ROWS - 1) / ROWS) + (unsigned)((a.t.rider_rows + ROWS - 1) / ROWS);
  if constexpr (RING_PAIR != RING_DEEP) {
    if ((int)grid >= cu_count()) return launch_ring<ATTN, FFN, PROJ, RING_PAIR>(a, grid, s);
  }
  return launch_ring<ATTN, FFN, PROJ, RING_DEEP>(a, grid, s);
}

}  // namespace

#if !TBX_TILE_SINGLE
extern "C" int64_t tbx_pack_weight_mfma32_size(int n, int k, int groups) {
  const int64_t u = mfma32_units(n, k, groups);
  return u < 0 ? u : u * UNIT;
}

extern "C" int tbx_pack_weight_mfma32(const float* w, const float* bias, int n, int k, int ld, int groups, int wt, float* out,
                                      void* stream) {
  if (w == nullptr || out == nullptr || ld <= 0) return TBX_ERR_ARG;
  const int64_t total = tbx_pack_weight_mfma32_size(n, k, groups);
  if (total < 0) return (int)total;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_mfma32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, bias, n, k, ld, groups, wt, out, total);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_pack_weight_mfma32_multi(const tbx_pack_job_t* jobs, int n_jobs, void* stream) {
  if (n_jobs < 0 || (n_jobs > 0 && jobs == nullptr)) return TBX_ERR_ARG;
  if (n_jobs == 0) return TBX_OK;
  for (int j = 0; j < n_jobs; ++j) {  // every job is checked before the first launch
    if (jobs[j].w == nullptr || jobs[j].out == nullptr || jobs[j].ld <= 0) return TBX_ERR_ARG;
    const int64_t total = tbx_pack_weight_mfma32_size(jobs[j].n, jobs[j].k, jobs[j].groups);
    if (total < 0) return (int)total;
  }
  for (int j0 = 0; j0 < n_jobs; j0 += PACK_MULTI_MAX) {
    PackMultiArgs a;
    const int m = n_jobs - j0 < PACK_MULTI_MAX ? n_jobs - j0 : PACK_MULTI_MAX;
    int64_t most = 0;
    for (int j = 0; j < m; ++j) {
      const tbx_pack_job_t& q = jobs[j0 + j];
      if (q.w == nullptr || q.out == nullptr || q.ld <= 0) return TBX_ERR_ARG;
      const int64_t total = tbx_pack_weight_mfma32_size(q.n, q.k, q.groups);
      if (total < 0) return (int)total;
      a.job[j] = q, a.total[j] = total;
      most = total > most ? total : most;
    }
    const int blocks = (int)((most + 255) / 256 < 256 ? (most + 255) / 256 : 256);
    hipLaunchKernelGGL(pack_mfma32_multi_kernel, dim3(blocks, m), dim3(256), 0, (hipStream_t)stream, a);
  }
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

#endif  // !TBX_TILE_SINGLE

extern "C" int TBX_TILE_ENTRY(tbx_layer_tile)(const tbx_layer_tile_t* args, void* stream) {
  if (args == nullptr || args->x == nullptr || args->n_rows <= 0) return TBX_ERR_ARG;
  const tbx_layer_tile_t& t = *args;
  const bool attn = t.attn_out != nullptr, ffn = t.linear1_image != nullptr;
  const int proj = t.proj_image == nullptr ? 0 : (t.proj_n == 3 * D ? 2 : (t.proj_n == D ? 1 : -1));
  if (proj < 0 || (!attn && !ffn && proj == 0)) return TBX_ERR_ARG;
  if (attn && (t.row_no_valid == nullptr || t.fold_image == nullptr || t.out_proj_image == nullptr || t.ld_attn < 5 * D || t.ld_attn % 4 != 0))
    return TBX_ERR_ARG;
  if (ffn && (t.linear2_image == nullptr || t.norm2_weight == nullptr || t.norm2_bias == nullptr)) return TBX_ERR_ARG;
  if (proj && (t.qfold_image == nullptr || t.proj_norm_weight == nullptr || t.proj_norm_bias == nullptr || t.proj_out == nullptr ||
               t.ld_proj < (proj == 2 ? 7 * D : 5 * D) || t.ld_proj % 4 != 0))
    return TBX_ERR_ARG;
  if (t.kv16_out != nullptr && proj != 2) return TBX_ERR_ARG;
  if (t.drop_thresh != 0u && t.drop_seed == nullptr) return TBX_ERR_ARG;
  if ((((uintptr_t)t.x) | ((uintptr_t)t.attn_out) | ((uintptr_t)t.proj_out)) & 15) return TBX_ERR_ALIGN;
  if (t.rider_rows < 0) return TBX_ERR_ARG;
  if (t.rider_rows > 0) {
    if (attn || ffn || proj != 2) return TBX_ERR_UNSUPPORTED;
    if ((t.rider_in == nullptr && t.rider_pose3 == nullptr) || t.rider_add == nullptr || t.rider_valid == nullptr || t.rider_out == nullptr)
      return TBX_ERR_ARG;
    if (t.rider_pose3 != nullptr && (t.rider_freqs_xy == nullptr || t.rider_freqs_yaw == nullptr)) return TBX_ERR_ARG;
    for (int i = 0; i < 4; ++i)
      if (t.rider_images[i] == nullptr) return TBX_ERR_ARG;
    if ((((uintptr_t)t.rider_in) | ((uintptr_t)t.rider_add) | ((uintptr_t)t.rider_out)) & 15) return TBX_ERR_ALIGN;
  }
  TileArgs a;
  a.t = t;
  int e = 0;
  auto put = [&](const float* img, int unit0) {
    a.ent[e].img = img, a.ent[e].unit0 = unit0, a.ent[e].pad = 0;
    ++e;
  };
  if (attn) put(t.fold_image, 0), put(t.out_proj_image, 0);
  if (ffn) {
    for (int r = 0; r < 4; ++r) put(t.linear1_image, 8 * r);
    for (int r = 0; r < 4; ++r) put(t.linear2_image, 8 * r);
  }
  if (proj) {
    put(t.proj_image, 0);
    if (proj == 2) put(t.proj_image, 8), put(t.proj_image, 16);
    put(t.qfold_image, 0);
  }
  for (; e < 16; ++e) a.ent[e].img = nullptr, a.ent[e].unit0 = 0, a.ent[e].pad = 0;
  hipStream_t s = (hipStream_t)stream;
#define TBX_TL(A, F, P) \
  if (attn == A && ffn == F && proj == P) return launch<A, F, P>(a, s)
  TBX_TL(true, false, 1);   // after the self attention of a decoder layer: out_proj -> q | qt of the cross attention
  TBX_TL(true, true, 2);    // after the (cross) attention: out_proj -> FFN -> the next layer's q | k | v | qt
  TBX_TL(true, true, 0);    // ... of the last layer
  TBX_TL(false, false, 2);  // the first projection
  TBX_TL(false, false, 1);
  TBX_TL(true, false, 2);
  TBX_TL(true, false, 0);
#undef TBX_TL
  return TBX_ERR_UNSUPPORTED;
}

#if defined(TBX_STAGE_CLOCK) && !TBX_TILE_SINGLE
extern "C" int tbx_debug_tl_dump(unsigned long long* host_out, int max_launches) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tl_launch), sizeof(n)) != hipSuccess) return -1;
  const int m = (int)n < max_launches ? (int)n : max_launches;
  if (m > 0 && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tl_clk), (size_t)m * 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_launch), &z, sizeof(z));
  return m;
}
#endif
