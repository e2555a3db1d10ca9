// tbx_tl_tail_tile: the traffic lights' tail behind their LAST transformer layer for LARGE launches (several scenes' lights: 2,048
// rows at 16 scenes, 8,192 at 64), as one straight-line tile kernel - what tbx_knarpe_dec_layer's `lights` tail does inside the
// one-launch layer for <= 384 rows (traffic_bots.py:188-199):
//   * for each of the 4 layers l of the agents' block the K/V rows its light cross-attention gathers,
//       k | v = in_proj_kv,l(norm_tgt,l(x))  ->  kv_out[row, l * 256 ..]   (fp32, or bfloat16 with kv_bf16)      transformer_rpe.py:220-223
//   * the next-state logits  clamp(mlp(x) masked, -3, 3)  ->  logits_out [rows, n_state]                           traffic_light.py:249-286
// Until round 5 this was a 16-row tbx_rowchain program on the exact-fp32 MFMA: 48 us per step at 2,048 rows, 97 us at 8,192 (8 % / 7 %
// of the step, profiles/r05_s16_kernel_stats.md / r05_s64_kernel_stats.md): 330 kFLOP per row in 11 dependent interpreter stages.
// Here: tile_core.h's blocks (16-row tiles, 8 waves, LayerNorm straight into bf16 planes, LINEAR = split-bf16 products - or ONE bf16
// product in the TBX_TILE_SINGLE build, tbx_tl_tail_tile_bf16 - on v_mfma_f32_16x16x32_bf16 with the weights as per-wave register
// units one unit ahead), 11 units per wave: per layer the k half and the v half (wave w = output tile w of each), then the
// predictor's three layers.
#include "tile_core.h"

using namespace tbx_tile;

namespace {

constexpr int ROWS = 16;
typedef Planes<ROWS, 4> PL;  // K = 128 everywhere
constexpr int PLANE = PL::PLANE;
constexpr int XLD = 132;
constexpr int NPL = TBX_TILE_SINGLE ? 1 : 2;
constexpr size_t LDS_BYTES = ROWS * XLD * sizeof(float) + 2 * NPL * PLANE;

struct TailArgs {
  tbx_tl_tail_t t;
  const float* x;
  int64_t n_rows;
};

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
    *(__bf16*)(P + PLANE + o) = (__bf16)(y - (float)h);
#endif
  }
}

__device__ __forceinline__ f32x4 stage(const W& w, const char* P, int aoff) {
  Acc acc;
  acc.zero();
#pragma unroll
  for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], P + aoff, s);
  return acc.sum() + w.bias;
}

__global__ __launch_bounds__(NT) void tile_tail_kernel(const TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;
  char* Pa = (char*)(X + ROWS * XLD);
  char* Pb = Pa + NPL * PLANE;
  const tbx_tl_tail_t& t = a.t;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nv = (a.n_rows - row0) < ROWS ? (int)(a.n_rows - row0) : ROWS;
  const bool row_ok = j < nv;
  const int64_t grow = row0 + (row_ok ? j : 0);
  const int aoff = PL::lane_off(lane, 0);
  const int c_out = 16 * wave + 4 * g;
  // the tile's rows
  const int xr = tid >> 5, xc4 = tid & 31;
  f32x4 xv = {0.f, 0.f, 0.f, 0.f};
  if (xr < nv) xv = gld4(a.x + (row0 + xr) * D + xc4 * 4);
  // units: kv image of layer l = 16 tiles (k: 0..7, v: 8..15), the wave takes tile `wave` of each half; predictor: 8 / 8 / 1 tiles
  W wa, wb;
  load_unit(wa, t.kv_images[0], wave, lane);
  *(f32x4*)(X + xr * XLD + xc4 * 4) = xv;
  __syncthreads();
#pragma unroll
  for (int l = 0; l < 4; ++l) {
#pragma unroll
    for (int q = 0; q < 2; ++q) ln_to_planes(X, Pa, wave * 2 + q, lane, t.norm_weight[l], t.norm_bias[l], t.norm_eps[l]);
    load_unit(wb, t.kv_images[l], 8 + wave, lane);
    __syncthreads();
    const f32x4 kq = stage(wa, Pa, aoff);
    if (l + 1 < 4) load_unit(wa, t.kv_images[l + 1], wave, lane);
    else load_unit(wa, t.mlp_images[0], wave, lane);
    const f32x4 vq = stage(wb, Pa, aoff);
    if (row_ok) {
      if (t.kv_bf16) {
        TBX_GLOBAL uint16_t* o = (TBX_GLOBAL uint16_t*)t.kv_out + grow * (int64_t)t.ld_kv + l * 2 * D + c_out;
        *(TBX_GLOBAL u32x2*)o = __builtin_bit_cast(u32x2, __builtin_convertvector(kq, bf16x4));
        *(TBX_GLOBAL u32x2*)(o + D) = __builtin_bit_cast(u32x2, __builtin_convertvector(vq, bf16x4));
      } else {
        float* o = (float*)t.kv_out + grow * (int64_t)t.ld_kv + l * 2 * D + c_out;
        gst4(o, kq);
        gst4(o + D, vq);
      }
    }
    __syncthreads();  // (Pa is rewritten by the next layer's LayerNorm)
  }
  // ---- the state predictor: x -> relu -> relu -> n_state logits (traffic_light.py:279-286), rows of invalid lights -> 0, clamp
  {
    const int o = tid;  // x itself (no LayerNorm) as planes: thread = (row tid >> 5, 4 columns)
    planes_write4<PL>(Pa, o >> 5, (o & 31) * 4, *(const f32x4*)(X + (o >> 5) * XLD + (o & 31) * 4));
  }
  load_unit(wb, t.mlp_images[1], wave, lane);
  __syncthreads();
  planes_write4<PL>(Pb, j, c_out, relu4(stage(wa, Pa, aoff)));
  if (wave == 0) load_unit(wa, t.mlp_images[2], 0, lane);
  __syncthreads();
  planes_write4<PL>(Pa, j, c_out, relu4(stage(wb, Pb, aoff)));
  __syncthreads();
  if (wave == 0) {  // the last layer's one tile of 16 (zero-padded) outputs: lane = (row j, outputs 4 g .. 4 g + 3)
    f32x4 v = stage(wa, Pa, aoff);
    if (row_ok) {
      const bool inv = *(const TBX_GLOBAL uint8_t*)(t.tl_invalid + grow) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * g + i;
        if (c < t.n_state) {
          float y = inv ? 0.f : v[i];
          y = fminf(fmaxf(y, t.clamp_lo), t.clamp_hi);
          *(TBX_GLOBAL float*)(t.logits_out + grow * t.n_state + c) = y;
        }
      }
    }
  }
}

}  // namespace

extern "C" int TBX_TILE_ENTRY(tbx_tl_tail_tile)(const float* x, int64_t n_rows, const tbx_tl_tail_t* tail, void* stream) {
  if (x == nullptr || tail == nullptr || n_rows <= 0) return TBX_ERR_ARG;
  const tbx_tl_tail_t& t = *tail;
  for (int l = 0; l < 4; ++l)
    if (t.kv_images[l] == nullptr || t.norm_weight[l] == nullptr || t.norm_bias[l] == nullptr) return TBX_ERR_ARG;
  for (int i = 0; i < 3; ++i)
    if (t.mlp_images[i] == nullptr) return TBX_ERR_ARG;
  if (t.kv_out == nullptr || t.tl_invalid == nullptr || t.logits_out == nullptr || t.n_state <= 0 || t.n_state > 16) return TBX_ERR_ARG;
  if (t.ld_kv < 8 * D || (t.ld_kv % 4) || (((uintptr_t)x) & 15) || (((uintptr_t)t.kv_out) & 15)) return TBX_ERR_ALIGN;
  TailArgs a{t, x, n_rows};
  static tbx::PerDeviceOnce lds_attr;
  if (!lds_attr([&] { return hipFuncSetAttribute((const void*)tile_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) == hipSuccess; }))
    return TBX_ERR_LAUNCH;
  hipLaunchKernelGGL(tile_tail_kernel, dim3((unsigned)((n_rows + ROWS - 1) / ROWS)), dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
