// tbx_heads_tile: the agents' heads for LARGE launches (traffic_bots.py:206-221) as one straight-line kernel on the building blocks
// of tile_core.h (16-row tiles, split-bf16 v_mfma_f32_16x16x32_bf16 stages, bf16 hi / lo planes in LDS, transposed products):
//   x += navi_valid ? add_navi.mlp([x | navi_emb]) : 0          add_navi_latent.py:52-65 (mlp = 256 -> 128 -> 128 -> 128, relu each)
//   x += latent valid ? add_latent.mlp([x | latent_emb]) : 0
//   action = sum over the agent's type of branch_g(x)            action_head.py:74-100: the three per-type branches as one 128 -> 384
//            stage, a block-diagonal 3 x (128 -> 128) stage, a block-diagonal 3 x (128 -> 16, the 2 outputs zero-padded) stage and
//            the masked sum in branch order from 0.
// Both embeddings arrive as mlp_in(.) with their invalid rows already zeroed (the navigation embedding from the auxiliary stream,
// the latent embedding once per rollout). 15 unit rounds of weights (~1 MB per tile) instead of a 44-stage tbx_rowchain program.
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
  Entry ent[16];
};

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
  constexpr int E_END = 15;
  W wb[2];
  load_unit(wb[0], a.ent[0], wave, lane);
#define TBX_NEXT(E)                                                                                                                 \
  do {                                                                                                                              \
    if constexpr ((E) + 1 < E_END)                                                                                                  \
      load_unit(wb[((E) + 1) & 1], a.ent[(E) + 1].img, a.ent[(E) + 1].unit0 + ((E) + 1 == 14 ? (wave < 3 ? wave : 2) : wave), lane); \
  } while (0)
  const bool ok_navi = *(const TBX_GLOBAL uint8_t*)(t.navi_valid + grow) != 0;
  const bool ok_lat = *(const TBX_GLOBAL uint8_t*)(t.latent_invalid + grow) == 0;
  {
    const int r = tid >> 5, c4 = tid & 31;
    f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vn = vx, vl = vx;
    if (r < nv) {
      vx = gld4(t.x + (row0 + r) * D + c4 * 4);
      vn = gld4(t.navi_emb + (row0 + r) * D + c4 * 4);
      vl = gld4(t.latent_emb + (row0 + r) * D + c4 * 4);
    }
    *(f32x4*)(X + r * XLD + c4 * 4) = vx;
    planes_write4<PL>(Pa, r, c4 * 4, vx);
    planes_write4<PL>(Pa, r, D + c4 * 4, vn);
    planes_write4<PL>(Pa, r, 2 * D + c4 * 4, vl);
  }
  __syncthreads();

  // ---- the two adders: entries 4 * A .. 4 * A + 3
#define TBX_ADDER(A, ZSTEP, OK)                                                                     \
  do {                                                                                              \
    {                                                                                               \
      Acc acc;                                                                                      \
      acc.zero();                                                                                   \
      TBX_NEXT(4 * (A));                                                                            \
      const W& w0 = wb[(4 * (A)) & 1];                                                              \
      const f32x4 bias = w0.bias;                                                                   \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w0.hi[s], w0.lo[s], Pa + aoff, s); \
      TBX_NEXT(4 * (A) + 1);                                                                        \
      const W& w1 = wb[(4 * (A) + 1) & 1];                                                          \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w1.hi[s], w1.lo[s], Pa + aoff, (ZSTEP) + s); \
      planes_write4<PL>(Pb, j, c_out, relu4(acc.sum() + bias));                                     \
    }                                                                                               \
    __syncthreads();                                                                                \
    {                                                                                               \
      Acc acc;                                                                                      \
      acc.zero();                                                                                   \
      TBX_NEXT(4 * (A) + 2);                                                                        \
      const W& w = wb[(4 * (A) + 2) & 1];                                                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, s); \
      planes_write4<PL>(Pb, j, D + c_out, relu4(acc.sum() + w.bias));                               \
    }                                                                                               \
    __syncthreads();                                                                                \
    {                                                                                               \
      Acc acc;                                                                                      \
      acc.zero();                                                                                   \
      TBX_NEXT(4 * (A) + 3);                                                                        \
      const W& w = wb[(4 * (A) + 3) & 1];                                                           \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, w.hi[s], w.lo[s], Pb + aoff, 4 + s); \
      f32x4 xv = *(const f32x4*)(X + j * XLD + c_out);                                              \
      if (OK) xv += relu4(acc.sum() + w.bias);                                                      \
      *(f32x4*)(X + j * XLD + c_out) = xv;                                                          \
      planes_write4<PL>(Pa, j, c_out, xv);                                                          \
    }                                                                                               \
    __syncthreads();                                                                                \
  } while (0)
  TBX_ADDER(0, 4, ok_navi);
  TBX_ADDER(1, 8, ok_lat);
#undef TBX_ADDER

  // ---- action head, layer 1: the three branches' first layers on the same x (entries 8..10) -> Pb[g * 128 ..]
#define TBX_L1(R)                                                                                   \
  do {                                                                                              \
    Acc acc;                                                                                        \
    acc.zero();                                                                                     \
    TBX_NEXT(8 + (R));                                                                              \
    const W& w = wb[(8 + (R)) & 1];                                                                 \
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
    const W& w = wb[(11 + (R)) & 1];                                                                \
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
    const W& w = wb[14 & 1];
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
}

}  // namespace

extern "C" int tbx_heads_tile(const tbx_heads_tile_t* args, void* stream) {
  if (args == nullptr || args->n_rows <= 0) return TBX_ERR_ARG;
  const tbx_heads_tile_t& t = *args;
  if (t.x == nullptr || t.navi_emb == nullptr || t.latent_emb == nullptr || t.navi_valid == nullptr || t.latent_invalid == nullptr ||
      t.type_mask == nullptr || t.action_out == nullptr || t.mask_stride < t.n_rows)
    return TBX_ERR_ARG;
  for (int i = 0; i < 9; ++i)
    if (t.images[i] == nullptr) return TBX_ERR_ARG;
  if ((((uintptr_t)t.x) | ((uintptr_t)t.navi_emb) | ((uintptr_t)t.latent_emb)) & 15) return TBX_ERR_ALIGN;
  HeadsArgs a;
  a.t = t;
  int e = 0;
  auto put = [&](const float* img, int unit0) {
    a.ent[e].img = img, a.ent[e].unit0 = unit0, a.ent[e].pad = 0;
    ++e;
  };
  for (int ad = 0; ad < 2; ++ad) {
    put(t.images[3 * ad], 0), put(t.images[3 * ad], 8);  // 256 -> 128: the k-chunks [x] and [embedding]
    put(t.images[3 * ad + 1], 0), put(t.images[3 * ad + 2], 0);
  }
  for (int r = 0; r < 3; ++r) put(t.images[6], 8 * r);
  for (int r = 0; r < 3; ++r) put(t.images[7], 8 * r);
  put(t.images[8], 0);
  for (; e < 16; ++e) a.ent[e].img = nullptr, a.ent[e].unit0 = 0, a.ent[e].pad = 0;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)tile_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess)
      return TBX_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(tile_heads_kernel, dim3((unsigned)((t.n_rows + ROWS - 1) / ROWS)), dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
