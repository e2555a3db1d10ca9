// tbx_heads_tile: the agents' heads for LARGE launches (traffic_bots.py:206-221) as one straight-line kernel on the building blocks
// of tile_core.h (16-row tiles, split-bf16 v_mfma_f32_16x16x32_bf16 stages, bf16 hi / lo planes in LDS, transposed products):
//   x += navi_valid ? add_navi.mlp([x | navi_emb]) : 0          add_navi_latent.py:52-65 (mlp = 256 -> 128 -> 128 -> 128, relu each)
//   x += latent valid ? add_latent.mlp([x | latent_emb]) : 0
//   action = sum over the agent's type of branch_g(x)            action_head.py:74-100: the three per-type branches as one 128 -> 384
//            stage, a block-diagonal 3 x (128 -> 128) stage, a block-diagonal 3 x (128 -> 16, the 2 outputs zero-padded) stage and
//            the masked sum in branch order from 0.
// Both embeddings arrive as mlp_in(.) with their invalid rows already zeroed (the navigation embedding from the auxiliary stream,
// the latent embedding once per rollout). 15 unit rounds of weights (~1 MB per tile) instead of a 44-stage tbx_rowchain program.
// RAW (training's stepping pass, tbx_heads_tile_t.raw): the embeddings are made here, with the adders' keyed dropouts -
//   navi_emb = mask(mlp_in(dest_feature + mlp_pe(navi_pe))), latent_emb = mask(mlp_in(z))   (navigation.py:65-79, add_navi_latent.py:43-50)
// 7 more unit rounds in front; every relu of the two adders' six-layer MLPs is followed by its DROPOUT (mlp.py:60-61) with the mask of
// tbx_keyed_dropout for (seed, site, step, row, column) - the masks of the row-chain stages this replaces (two 53 us launches).
#include "tile_core.h"

using namespace tbx_tile;

namespace {

constexpr int ROWS = 16;
typedef Planes<ROWS, 12> PL;  // K <= 384: [x | navi_emb | latent_emb], the action head's 3 x 128 hidden rows
constexpr int PLANE = PL::PLANE;
constexpr int XLD = 132;
constexpr size_t LDS_BYTES = ROWS * XLD * sizeof(float) + 4 * PLANE + 3 * ROWS * 2 * sizeof(float);

struct HeadsArgs {
  tbx_heads_tile_t t;
  Entry ent[24];
};

template <bool RAW>
__global__ __launch_bounds__(NT) void tile_heads_kernel(const HeadsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* X = lds;
  char* Pa = (char*)(X + ROWS * XLD);
  char* Pb = Pa + 2 * PLANE;
  float* O = (float*)(Pb + 2 * PLANE);  // [3 branches][16 rows][2]
  const tbx_heads_tile_t& t = a.t;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nv = (t.n_rows - row0) < ROWS ? (int)(t.n_rows - row0) : ROWS;
  const bool row_ok = j < nv;
  const int64_t grow = row0 + (row_ok ? j : 0);
  const int aoff = PL::lane_off(lane, 0);
  const int c_out = 16 * wave + 4 * g;
  constexpr int OFF = RAW ? 7 : 0;  // entries in front of the 15 of the heads proper
  constexpr int E_END = 15;
  W wb[2];
  load_unit(wb[0], a.ent[0], wave, lane);
  // (entry OFF + E lives in slot (OFF + E) & 1; the last entry's unit is the branch's, not the wave's)
#define TBX_NEXT(E)                                                                                                                 \
  do {                                                                                                                              \
    if constexpr ((E) + 1 < E_END)                                                                                                  \
      load_unit(wb[(OFF + (E) + 1) & 1], a.ent[OFF + (E) + 1].img,                                                                  \
                a.ent[OFF + (E) + 1].unit0 + ((E) + 1 == 14 ? (wave < 3 ? wave : 2) : wave), lane);                                 \
  } while (0)
#define TBX_SLOT(E) wb[(OFF + (E)) & 1]
  const bool ok_navi = *(const TBX_GLOBAL uint8_t*)(t.navi_valid + grow) != 0;
  const bool ok_lat = *(const TBX_GLOBAL uint8_t*)(t.latent_invalid + grow) == 0;
  const bool dropping = t.drop_thresh != 0u;
  // the keyed dropout of site `SITE` (index into t.drop_site) on the lane's 4 channels of its row
  auto drop = [&](f32x4 v, int site, int col) -> f32x4 {
    if (!dropping || t.drop_site[site] < 0) return v;
    DropKey4 dk;
    dk.init(t.drop_seed, (uint32_t)t.drop_site[site], (uint32_t)t.drop_step, t.drop_thresh, t.drop_scale);
    return dk.apply(v, grow, col, D);
  };
  {
    const int r = tid >> 5, c4 = tid & 31;
    f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vn = vx, vl = vx;
    if (r < nv) {
      vx = gld4(t.x + (row0 + r) * D + c4 * 4);
      if constexpr (RAW) {
        vn = gld4(t.navi_pe + (row0 + r) * D + c4 * 4);  // (-> Pb[:, 0:128): the navigation MLP's input)
        if (c4 < 4) vl = gld4(t.latent_z + (row0 + r) * (int64_t)t.ld_z + c4 * 4);
      } else {
        vn = gld4(t.navi_emb + (row0 + r) * D + c4 * 4);
        vl = gld4(t.latent_emb + (row0 + r) * D + c4 * 4);
      }
    }
    *(f32x4*)(X + r * XLD + c4 * 4) = vx;
    planes_write4<PL>(Pa, r, c4 * 4, vx);
    if constexpr (RAW) {
      planes_write4<PL>(Pb, r, c4 * 4, vn);
      if (c4 < 8) planes_write4<PL>(Pb, r, D + c4 * 4, c4 < 4 ? vl : (f32x4){0.f, 0.f, 0.f, 0.f});  // z: 16 values, zero-padded to one 32-k step
    } else {
      planes_write4<PL>(Pa, r, D + c4 * 4, vn);
      planes_write4<PL>(Pa, r, 2 * D + c4 * 4, vl);
    }
  }
  __syncthreads();
  if constexpr (RAW) {
    // ---- prologue entries 0..6: mlp_pe | navi mlp_in 0, 1, 2 | latent mlp_in 0 (k = 32), 1, 2
#define TBX_PRE(P) load_unit(wb[((P) + 1) & 1], a.ent[(P) + 1].img, a.ent[(P) + 1].unit0 + ((P) + 1 == 4 ? (wave >> 2) : wave), lane)
    {  // navigation feature = dest_feature + mlp_pe(navi_pe)   (Pb[0:128) -> Pb[256:384))
      TBX_PRE(0);
      const W& w = wb[0];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, s);
      f32x4 v = acc.sum() + w.bias;
      if (row_ok) v += gld4(t.dest_feature + grow * D + c_out);
      planes_write4<PL>(Pb, j, 2 * D + c_out, v);
    }
    __syncthreads();
    {  // navi mlp_in 0: Pb[256:384) -> Pb[0:128)
      TBX_PRE(1);
      const W& w = wb[1];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, 8 + s);
      planes_write4<PL>(Pb, j, c_out, drop(relu4(acc.sum() + w.bias), 0, c_out));
    }
    __syncthreads();
    {  // navi mlp_in 1: Pb[0:128) -> Pb[256:384)
      TBX_PRE(2);
      const W& w = wb[0];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, s);
      planes_write4<PL>(Pb, j, 2 * D + c_out, drop(relu4(acc.sum() + w.bias), 1, c_out));
    }
    __syncthreads();
    {  // navi mlp_in 2: Pb[256:384) -> navi_emb = Pa[128:256), rows without a valid destination 0
      TBX_PRE(3);
      const W& w = wb[1];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, 8 + s);
      f32x4 v = drop(relu4(acc.sum() + w.bias), 2, c_out);
      if (!ok_navi) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      planes_write4<PL>(Pa, j, D + c_out, v);
    }
    {  // latent mlp_in 0 (k = 32: unit = 4 tiles, the wave's is group wave & 3 of unit wave >> 2): z = Pb[128:160) -> Pb[0:128)
      TBX_PRE(4);
      const W& w = wb[0];
      const int q = wave & 3;
      const bf16x8 wh = q == 0 ? w.hi[0] : (q == 1 ? w.hi[1] : (q == 2 ? w.hi[2] : w.hi[3]));
      const bf16x8 wl = q == 0 ? w.lo[0] : (q == 1 ? w.lo[1] : (q == 2 ? w.lo[2] : w.lo[3]));
      const f32x4 b0 = unit_bias(a.ent[4].img, a.ent[4].unit0 + (wave >> 2), q, lane);
      Acc acc;
      acc.zero();
      mfma_step<PLANE>(acc, wh, wl, Pb + aoff, 4);
      planes_write4<PL>(Pb, j, c_out, drop(relu4(acc.sum() + b0), 6, c_out));  // (Pb[0:128) was last read two barriers ago)
    }
    __syncthreads();
    {  // latent mlp_in 1: Pb[0:128) -> Pb[256:384)
      TBX_PRE(5);
      const W& w = wb[1];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, s);
      planes_write4<PL>(Pb, j, 2 * D + c_out, drop(relu4(acc.sum() + w.bias), 7, c_out));
    }
    __syncthreads();
    {  // latent mlp_in 2: Pb[256:384) -> latent_emb = Pa[256:384), invalid latents 0
      load_unit(wb[1], a.ent[7], wave, lane);  // (the heads' first entry)
      const W& w = wb[0];
      Acc acc;
      acc.zero();
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, 8 + s);
      f32x4 v = drop(relu4(acc.sum() + w.bias), 8, c_out);
      if (!ok_lat) v = (f32x4){0.f, 0.f, 0.f, 0.f};
      planes_write4<PL>(Pa, j, 2 * D + c_out, v);
    }
    __syncthreads();
#undef TBX_PRE
  }

  // ---- the two adders: entries 4 * A .. 4 * A + 3
  // (SITE0: the adder's first dropout site in t.drop_site - its MLP's three relu outputs are dropped as the chain's DROPOUT stages are)
#define TBX_ADDER(A, ZSTEP, OK, SITE0)                                                              \
  do {                                                                                              \
    {                                                                                               \
      Acc acc;                                                                                      \
      acc.zero();                                                                                   \
      TBX_NEXT(4 * (A));                                                                            \
      const W& w0 = TBX_SLOT(4 * (A));                                                              \
      const f32x4 bias = w0.bias;                                                                   \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w0.hi[s], w0.lo[s], Pa + aoff, s); \
      TBX_NEXT(4 * (A) + 1);                                                                        \
      const W& w1 = TBX_SLOT(4 * (A) + 1);                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w1.hi[s], w1.lo[s], Pa + aoff, (ZSTEP) + s); \
      planes_write4<PL>(Pb, j, c_out, drop(relu4(acc.sum() + bias), (SITE0), c_out));               \
    }                                                                                               \
    __syncthreads();                                                                                \
    {                                                                                               \
      Acc acc;                                                                                      \
      acc.zero();                                                                                   \
      TBX_NEXT(4 * (A) + 2);                                                                        \
      const W& w = TBX_SLOT(4 * (A) + 2);                                                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, s); \
      planes_write4<PL>(Pb, j, D + c_out, drop(relu4(acc.sum() + w.bias), (SITE0) + 1, c_out));     \
    }                                                                                               \
    __syncthreads();                                                                                \
    {                                                                                               \
      Acc acc;                                                                                      \
      acc.zero();                                                                                   \
      TBX_NEXT(4 * (A) + 3);                                                                        \
      const W& w = TBX_SLOT(4 * (A) + 3);                                                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, 4 + s); \
      f32x4 xv = *(const f32x4*)(X + j * XLD + c_out);                                              \
      if (OK) xv += drop(relu4(acc.sum() + w.bias), (SITE0) + 2, c_out);                            \
      *(f32x4*)(X + j * XLD + c_out) = xv;                                                          \
      planes_write4<PL>(Pa, j, c_out, xv);                                                          \
    }                                                                                               \
    __syncthreads();                                                                                \
  } while (0)
  TBX_ADDER(0, 4, ok_navi, 3);
  TBX_ADDER(1, 8, ok_lat, 9);
#undef TBX_ADDER

  // ---- action head, layer 1: the three branches' first layers on the same x (entries 8..10) -> Pb[g * 128 ..]
#define TBX_L1(R)                                                                                   \
  do {                                                                                              \
    Acc acc;                                                                                        \
    acc.zero();                                                                                     \
    TBX_NEXT(8 + (R));                                                                              \
    const W& w = TBX_SLOT(8 + (R));                                                                 \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pa + aoff, s); \
    planes_write4<PL>(Pb, j, (R) * D + c_out, relu4(acc.sum() + w.bias));                           \
  } while (0)
  TBX_L1(0);
  TBX_L1(1);
  TBX_L1(2);
#undef TBX_L1
  __syncthreads();
  // ---- layer 2: block-diagonal (entries 11..13): branch g reads Pb[g * 128 ..] -> Pa[g * 128 ..]
#define TBX_L2(R)                                                                                   \
  do {                                                                                              \
    Acc acc;                                                                                        \
    acc.zero();                                                                                     \
    TBX_NEXT(11 + (R));                                                                             \
    const W& w = TBX_SLOT(11 + (R));                                                                \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, 4 * (R) + s); \
    planes_write4<PL>(Pa, j, (R) * D + c_out, relu4(acc.sum() + w.bias));                           \
  } while (0)
  TBX_L2(0);
  TBX_L2(1);
  TBX_L2(2);
#undef TBX_L2
  __syncthreads();
  // ---- layer 3 (entry 14): wave g < 3 = branch g, 16 zero-padded outputs of which the first 2 are the action
  if (wave < 3) {
    Acc acc;
    acc.zero();
    const W& w = TBX_SLOT(14);
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pa + aoff, 4 * wave + s);
    const f32x4 o = acc.sum() + w.bias;
    if (g == 0) O[(wave * ROWS + j) * 2] = o[0], O[(wave * ROWS + j) * 2 + 1] = o[1];
  }
  __syncthreads();
  if (tid < 2 * ROWS) {  // the masked sum over the branches, in branch order from 0
    const int r = tid >> 1, c = tid & 1;
    if (r < nv) {
      float v = 0.f;
#pragma unroll
      for (int b = 0; b < 3; ++b)
        if (*(const TBX_GLOBAL uint8_t*)(t.type_mask + (int64_t)b * t.mask_stride + row0 + r) == 0) v += O[(b * ROWS + r) * 2 + c];
      *(TBX_GLOBAL float*)(t.action_out + (row0 + r) * 2 + c) = v;
    }
  }
#undef TBX_NEXT
#undef TBX_SLOT
}

}  // namespace

extern "C" int TBX_TILE_ENTRY(tbx_heads_tile)(const tbx_heads_tile_t* args, void* stream) {
  if (args == nullptr || args->n_rows <= 0) return TBX_ERR_ARG;
  const tbx_heads_tile_t& t = *args;
  const bool raw = t.raw != 0;
  if (t.x == nullptr || t.navi_valid == nullptr || t.latent_invalid == nullptr || t.type_mask == nullptr || t.action_out == nullptr ||
      t.mask_stride < t.n_rows)
    return TBX_ERR_ARG;
  if (!raw && (t.navi_emb == nullptr || t.latent_emb == nullptr)) return TBX_ERR_ARG;
  for (int i = 0; i < 9; ++i)
    if (t.images[i] == nullptr) return TBX_ERR_ARG;
  if ((((uintptr_t)t.x) | ((uintptr_t)t.navi_emb) | ((uintptr_t)t.latent_emb)) & 15) return TBX_ERR_ALIGN;
  if (raw) {
    if (t.navi_pe == nullptr || t.dest_feature == nullptr || t.latent_z == nullptr || t.ld_z < 16 || (t.ld_z % 4)) return TBX_ERR_ARG;
    for (int i = 0; i < 7; ++i)
      if (t.raw_images[i] == nullptr) return TBX_ERR_ARG;
    if ((((uintptr_t)t.navi_pe) | ((uintptr_t)t.dest_feature) | ((uintptr_t)t.latent_z)) & 15) return TBX_ERR_ALIGN;
  }
  if (t.drop_thresh != 0u && t.drop_seed == nullptr) return TBX_ERR_ARG;
  HeadsArgs a;
  a.t = t;
  int e = 0;
  auto put = [&](const float* img, int unit0) {
    a.ent[e].img = img, a.ent[e].unit0 = unit0, a.ent[e].pad = 0;
    ++e;
  };
  if (raw)
    for (int i = 0; i < 7; ++i) put(t.raw_images[i], 0);
  for (int ad = 0; ad < 2; ++ad) {
    put(t.images[3 * ad], 0), put(t.images[3 * ad], 8);  // 256 -> 128: the k-chunks [x] and [embedding]
    put(t.images[3 * ad + 1], 0), put(t.images[3 * ad + 2], 0);
  }
  for (int r = 0; r < 3; ++r) put(t.images[6], 8 * r);
  for (int r = 0; r < 3; ++r) put(t.images[7], 8 * r);
  put(t.images[8], 0);
  for (; e < 24; ++e) a.ent[e].img = nullptr, a.ent[e].unit0 = 0, a.ent[e].pad = 0;
  static tbx::PerDeviceOnce lds_attr;  // (per device, thread-safe: tbx_common.h)
  if (!lds_attr([&] { return !(hipFuncSetAttribute((const void*)tile_heads_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)tile_heads_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess); })) return TBX_ERR_LAUNCH;
  const dim3 grid((unsigned)((t.n_rows + ROWS - 1) / ROWS));
  if (raw)
    hipLaunchKernelGGL(tile_heads_kernel<true>, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(tile_heads_kernel<false>, grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
