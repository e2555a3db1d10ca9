// The temporal-PointNet window tile (see csrc/tile_window.hip for the design) as a device function, shared by tbx_window_tile's kernels
// and the fused front launch (csrc/front.hip).
#pragma once
#include "tile_core.h"

namespace tbx_window {
using namespace tbx_tile;


constexpr int RT = 2;           // windows (16-row tiles) per workgroup
constexpr int ROWS = 16 * RT;
typedef Planes<ROWS, 4> PL;     // K <= 128
constexpr int PLANE = PL::PLANE;
constexpr size_t LDS_BYTES = 4 * PLANE;

__device__ __forceinline__ float row16_max(float v) {  // maximum over the 16 lanes of a DPP row, in every lane
  v = fmaxf(v, tbx::dpp<tbx::DPP_XOR1>(v));
  v = fmaxf(v, tbx::dpp<tbx::DPP_XOR2>(v));
  v = fmaxf(v, tbx::dpp<tbx::DPP_HALF_MIRROR>(v));
  return fmaxf(v, tbx::dpp<tbx::DPP_MIRROR>(v));
}

// DM = width of the input MLP (64: "cat" mode, the pose embedding fills channels [64, 128); 128: "add" mode, a per-window
// feature row is added to the MLP's output)
// `block` = the workgroup's index among the window workgroups, lds_c = LDS_BYTES of LDS. pooled (LDS [RT][128] floats or NULL): the
// two pooled rows also stay on chip for a caller that goes on with them (csrc/front.hip: the first projection in the same launch)
template <int DM, bool ADD>
__device__ __forceinline__ void window_body(const tbx_window_tile_t& t, const int block, char* lds_c, float* pooled) {
  char* P0 = lds_c;             // ping: hi, lo
  char* P1 = P0 + 2 * PLANE;    // pong
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int nt = wave & 3, rt = wave >> 2;  // PointNet layers: the wave's output tile (16 of 64 channels) and row tile (window)
  const int64_t grp0 = (int64_t)block * RT;
  const int Wn = t.window;
  const int aoff = PL::lane_off(lane, rt * 16);
  const int c_out = 16 * nt + 4 * g;
  // the lane's row in the PointNet layers: step j of window grp0 + rt
  const int64_t grp = grp0 + rt;
  const bool grp_ok = grp < t.n_groups;
  bool inv = true;
  if (grp_ok && j < Wn) inv = *(const TBX_GLOBAL uint8_t*)(t.row_invalid + grp * Wn + j) != 0;

  W wb[2];
  // ---- inputs: attribute rows (first 32 columns; columns past attr_cols read as 0) -> P0[k 0..31]; cat mode: pose embedding -> P1[k 64..127]
  if (tid < ROWS * 8) {
    const int r = tid >> 3, c4 = tid & 7;
    const int64_t gq = grp0 + (r >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gq < t.n_groups && (r & 15) < Wn && c4 * 4 < t.attr_cols) v = gld4(t.attr + (gq * Wn + (r & 15)) * (int64_t)t.ld_attr + c4 * 4);
    planes_write4<PL>(P0, r, c4 * 4, v);
  }
  if constexpr (!ADD) {
    const int r = tid >> 4, c4 = tid & 15;
    const int64_t gq = grp0 + (r >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gq < t.n_groups && (r & 15) < Wn) v = gld4(t.pe + (gq * Wn + (r & 15)) * (int64_t)64 + c4 * 4);
    planes_write4<PL>(P1, r, 64 + c4 * 4, v);
  }
  if constexpr (DM == 64) {
    // input MLP 32 -> 64 (one unit of 4 tiles), 64 -> 64 twice (2 units of 2 tiles x 2 steps); wave = (tile nt, row tile rt)
    load_unit(wb[0], t.in_images[0], 0, lane);
    const f32x4 b_in0 = unit_bias(t.in_images[0], 0, nt, lane);
    __syncthreads();
    {
      load_unit(wb[1], t.in_images[1], nt >> 1, lane);
      Acc acc;
      acc.zero();
      const W& w = wb[0];
      // (groups are indexed with a wave-uniform runtime value: select by hand so the fragments stay in registers)
      const bf16x8 wh = nt == 0 ? w.hi[0] : (nt == 1 ? w.hi[1] : (nt == 2 ? w.hi[2] : w.hi[3]));
      const bf16x8 wl = nt == 0 ? w.lo[0] : (nt == 1 ? w.lo[1] : (nt == 2 ? w.lo[2] : w.lo[3]));
      mfma_step<PLANE>(acc, wh, wl, P0 + aoff, 0);
      planes_write4<PL>(P1, rt * 16 + j, c_out, relu4(acc.sum() + b_in0));
    }
    __syncthreads();
#define TBX_IN64(CUR, SRC, DST, RELU, NEXT_IMG, NEXT_UNIT, BIAS_IMG)                                      \
  do {                                                                                                    \
    const f32x4 bias = unit_bias(BIAS_IMG, nt >> 1, 2 * (nt & 1), lane);                                  \
    load_unit(wb[1 - (CUR)], NEXT_IMG, NEXT_UNIT, lane);                                                  \
    Acc acc;                                                                                              \
    acc.zero();                                                                                           \
    const W& w = wb[CUR];                                                                                 \
    const bool odd = (nt & 1) != 0;                                                                       \
    mfma_step<PLANE>(acc, odd ? w.hi[2] : w.hi[0], odd ? w.lo[2] : w.lo[0], (SRC) + aoff, 0);             \
    mfma_step<PLANE>(acc, odd ? w.hi[3] : w.hi[1], odd ? w.lo[3] : w.lo[1], (SRC) + aoff, 1);             \
    f32x4 v = acc.sum() + bias;                                                                           \
    if (RELU) v = relu4(v);                                                                               \
    planes_write4<PL>(DST, rt * 16 + j, c_out, v);                                                        \
  } while (0)
    TBX_IN64(1, P1, P0, true, t.in_images[2], nt >> 1, t.in_images[1]);
    __syncthreads();
    TBX_IN64(0, P0, P1, false, t.pn_images[0], nt, t.in_images[2]);
#undef TBX_IN64
    __syncthreads();
  } else {
    // input MLP 32 -> 128 (2 units of 4 tiles), 128 -> 128 twice (8 units of 4 steps); wave w = output tile w (16 of 128 channels),
    // BOTH row tiles (the weight fragments serve the two windows)
    const int c8 = 16 * wave + 4 * g;
    load_unit(wb[0], t.in_images[0], wave >> 2, lane);
    const f32x4 b_in0 = unit_bias(t.in_images[0], wave >> 2, wave & 3, lane);
    __syncthreads();
    {
      load_unit(wb[1], t.in_images[1], wave, lane);
      const W& w = wb[0];
      const int q = wave & 3;
      const bf16x8 wh = q == 0 ? w.hi[0] : (q == 1 ? w.hi[1] : (q == 2 ? w.hi[2] : w.hi[3]));
      const bf16x8 wl = q == 0 ? w.lo[0] : (q == 1 ? w.lo[1] : (q == 2 ? w.lo[2] : w.lo[3]));
#pragma unroll
      for (int r2 = 0; r2 < RT; ++r2) {
        Acc acc;
        acc.zero();
        mfma_step<PLANE>(acc, wh, wl, P0 + PL::lane_off(lane, r2 * 16), 0);
        planes_write4<PL>(P1, r2 * 16 + j, c8, relu4(acc.sum() + b_in0));
      }
    }
    __syncthreads();
    {  // layer 2: P1 -> P0, relu
      load_unit(wb[0], t.in_images[2], wave, lane);
      const W& w = wb[1];
#pragma unroll
      for (int r2 = 0; r2 < RT; ++r2) {
        Acc acc;
        acc.zero();
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) mfma_step<PLANE>(acc, w.hi[s2], w.lo[s2], P1 + PL::lane_off(lane, r2 * 16), s2);
        planes_write4<PL>(P0, r2 * 16 + j, c8, relu4(acc.sum() + w.bias));
      }
    }
    __syncthreads();
    {  // layer 3: P0 -> P1, no activation, + the window's feature row (input encoder "add" mode)
      load_unit(wb[1], t.pn_images[0], nt, lane);
      const W& w = wb[0];
#pragma unroll
      for (int r2 = 0; r2 < RT; ++r2) {
        Acc acc;
        acc.zero();
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) mfma_step<PLANE>(acc, w.hi[s2], w.lo[s2], P0 + PL::lane_off(lane, r2 * 16), s2);
        f32x4 v = acc.sum() + w.bias;
        const int64_t gq = grp0 + r2;
        if (gq < t.n_groups) v += gld4(t.pe + gq * (int64_t)D + c8);
        planes_write4<PL>(P1, r2 * 16 + j, c8, v);
      }
    }
    __syncthreads();
  }
  // ---- PointNet layers: P1 -> P0 -> P1 -> out (wb[1] holds layer 1's unit)
#define TBX_PN(CUR, SRC, DST, LAST, NEXT_IMG, LAYER)                                                      \
  do {                                                                                                    \
    if (!(LAST)) load_unit(wb[1 - (CUR)], NEXT_IMG, nt, lane);                                            \
    Acc acc;                                                                                              \
    acc.zero();                                                                                           \
    const W& w = wb[CUR];                                                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], (SRC) + aoff, s); \
    f32x4 h = relu4(acc.sum() + w.bias);                                                                  \
    if (t.drop_thresh != 0u && j < Wn) {                                                                  \
      DropKey4 dk;                                                                                        \
      dk.init(t.drop_seed, (uint32_t)t.drop_site[LAYER], (uint32_t)t.drop_step, t.drop_thresh, t.drop_scale); \
      h = dk.apply(h, grp * Wn + j, c_out, 64);                                                           \
    }                                                                                                     \
    f32x4 m;                                                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) m[r] = row16_max(inv ? -INFINITY : h[r]);               \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) m[r] = m[r] == -INFINITY ? 0.f : m[r];                  \
    if (!(LAST)) {                                                                                        \
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                               \
      planes_write4<PL>(DST, rt * 16 + j, c_out, inv ? z : h);                                            \
      planes_write4<PL>(DST, rt * 16 + j, 64 + c_out, inv ? z : m);                                       \
    } else if (j == 0) {                                                                                  \
      if (grp_ok) {                                                                                       \
        gst4(t.out + grp * D + c_out, m);                                                                 \
        gst4(t.out + grp * D + 64 + c_out, m);                                                            \
      }                                                                                                   \
      if (pooled != nullptr) {                                                                            \
        *(f32x4*)(pooled + rt * D + c_out) = m;                                                           \
        *(f32x4*)(pooled + rt * D + 64 + c_out) = m;                                                      \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
  TBX_PN(1, P1, P0, false, t.pn_images[1], 0);
  __syncthreads();
  TBX_PN(0, P0, P1, false, t.pn_images[2], 1);
  __syncthreads();
  TBX_PN(1, P1, P0, true, t.pn_images[2], 2);
#undef TBX_PN
}


}  // namespace tbx_window
