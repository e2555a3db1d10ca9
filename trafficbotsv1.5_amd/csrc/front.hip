// tbx_front: everything between tbx_agent_prep / tbx_tl_prep and a block's first decoder layer as ONE launch, for the closed loop at a
// few scenes (launches of <= 1024 windows). It replaced three dependent launches on the critical path of every step - the window
// PointNet (tbx_window_tile: 8 us), the K-nearest searches (tbx_knn_embed_multi_pe: 13 us) and the block's first projection
// (tbx_layer_tile + rider: 7 us) - whose work is independent or per-window:
//   window workgroups (2 windows each, csrc/window_core.h): the temporal PointNet, then - the two pooled rows never leave the CU -
//     LayerNorm + q | k | v = in_proj(norm x) + qt = W_rpe_k^T q of the first layer (transformer_rpe.py:207-211, attention_rpe.py:92-98,147)
//     on single-row B operands (tile_core.h row1), weights as per-wave register units;
//   rider workgroups (16 rows each): tbx_layer_tile_t's rider (the heads' navigation embedding);
//   search workgroups (one source row each, 256 of the 512 threads, csrc/knn_core.h): the K-nearest searches + pose-embedding job.
// The launch takes as long as its longest part (the searches) instead of the sum.
#include <string.h>

#include "knn_core.h"
#include "tile_core.h"
#include "window_core.h"

using namespace tbx_tile;

namespace {

typedef Planes<16, 4> RPL;  // the rider's planes (K = 128)
constexpr int XLD = 132;
constexpr size_t WIN_BYTES = tbx_window::LDS_BYTES;                   // the window body's ping / pong planes
constexpr size_t POOL_OFF = WIN_BYTES;                                // float pooled[2][128]
constexpr size_t ROWP_OFF = POOL_OFF + 2 * D * sizeof(float);         // row planes: Ph[2], Pq[2] of 512 B
constexpr size_t WIN_TOTAL = ROWP_OFF + 4 * 512;
constexpr size_t RIDER_TOTAL = 16 * XLD * sizeof(float) + 4 * RPL::PLANE;
constexpr size_t LDS_BYTES = WIN_TOTAL > RIDER_TOTAL ? WIN_TOTAL : RIDER_TOTAL;

struct FrontArgs {
  tbx_window_tile_t win;
  tbx_layer_tile_t lt;
  tbx_knn::KnnMulti knn;
  int n_win_blocks, n_rider_blocks, n_knn_blocks;
};

// the first projection of the workgroup's two pooled rows (global rows 2 * block, 2 * block + 1)
__device__ __forceinline__ void proj_rows(const tbx_layer_tile_t& t, const int block, const float* pooled, char* rowp) {
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g4 = lane >> 4;
  const bool col0 = (lane & 15) == 0;
  const int c_out = 16 * wave + 4 * g4;
  constexpr int LO = 256;
  char* Ph = rowp;          // [2][512]: LayerNorm(x) planes
  char* Pq = rowp + 1024;   // [2][512]: q planes
  W wb[2];
  load_unit(wb[0], t.proj_image, wave, lane);
  load_unit(wb[1], t.proj_image, 8 + wave, lane);
  if (wave < 2) row1::ln_planes(pooled + wave * D, Ph + wave * 512, lane, t.proj_norm_eps, t.proj_norm_weight, t.proj_norm_bias);
  __syncthreads();
  const int64_t row0 = (int64_t)block * 2;
  const int nv = (t.n_rows - row0) < 2 ? (int)(t.n_rows - row0) : 2;
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // q
    const f32x4 q = row1::gemv4(wb[0], Ph + r * 512, LO, 0, g4) + wb[0].bias;
    if (col0) {
      row1::put4(Pq + r * 512, LO, c_out, q);
      if (r < nv) gst4(t.proj_out + (row0 + r) * (int64_t)t.ld_proj + c_out, q);
    }
  }
  load_unit(wb[0], t.proj_image, 16 + wave, lane);
#pragma unroll
  for (int kv = 0; kv < 2; ++kv) {  // k (wb[1]), v (wb[0])
    const W& w = wb[1 - kv];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const f32x4 y = row1::gemv4(w, Ph + r * 512, LO, 0, g4) + w.bias;
      if (col0 && r < nv) {
        gst4(t.proj_out + (row0 + r) * (int64_t)t.ld_proj + (1 + kv) * D + c_out, y);
        if (t.kv16_out != nullptr) {
          const bf16x4 h16 = __builtin_convertvector(y, bf16x4);
          *(TBX_GLOBAL u32x2*)((uint16_t*)t.kv16_out + (row0 + r) * (2 * D) + kv * D + c_out) = __builtin_bit_cast(u32x2, h16);
        }
      }
    }
    if (kv == 0) load_unit(wb[1], t.qfold_image, wave, lane);
  }
  __syncthreads();  // q's planes complete
  {  // qt_h = W_rpe_k,h^T q_h: wave w = head w / 2, 4 of its 8 tiles of 16 channels, K = 32 (the head's own step)
    const W& w = wb[1];
    const int h = wave >> 1;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        Acc acc;
        acc.zero();
        row1::step(acc, w.hi[st], w.lo[st], Pq + r * 512, LO, h, g4);
        if (col0 && r < nv) gst4(t.proj_out + (row0 + r) * (int64_t)t.ld_proj + 3 * D + h * D + ((wave & 1) * 4 + st) * 16 + 4 * g4, acc.sum());
      }
  }
}

template <int DM, bool ADD>
__device__ __forceinline__ void front_body(const FrontArgs& a, int b, char* lds_c) {
  if (b < a.n_win_blocks) {
    float* pooled = (float*)(lds_c + POOL_OFF);
    tbx_window::window_body<DM, ADD>(a.win, b, lds_c, pooled);
    __syncthreads();
    proj_rows(a.lt, b, pooled, lds_c + ROWP_OFF);
    return;
  }
  b -= a.n_win_blocks;
  if (b < a.n_rider_blocks) {
    float* X = (float*)lds_c;
    char* Pa = (char*)(X + 16 * XLD);
    rider_tile<RPL, XLD>(a.lt, b, X, Pa, Pa + 2 * RPL::PLANE);
    return;
  }
  b -= a.n_rider_blocks;
  if (threadIdx.x < 256) tbx_knn::knn_multi_body(a.knn, b);
}

template <int DM, bool ADD>
__global__ __launch_bounds__(NT) void front_kernel(const FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_c[];
  front_body<DM, ADD>(a, (int)blockIdx.x, lds_c);
}

// The agents' front ("cat" input encoder, d_mlp 64) and the lights' ("add", d_mlp 128) of one closed-loop step as ONE launch (round 6,
// with tbx_knarpe_dec_layer_pair: the step on one queue): blocks [0, n_a) run the first, the rest the second - the two read nothing of
// each other's (the lights run one step ahead).
// Block order: the lights' blocks (all window blocks, the launch's longest) first, then the agents' (windows, rider, and last their
// K-nearest searches - the short ones): at configs[1] the grid is 64 + 228 = 292 workgroups on 256 CUs with one workgroup per CU (LDS),
// so the last 36 to start must be short ones (agents first, the lights' last 36 window blocks waited for a CU: the launch 13 -> 20 us).
struct FrontPair {
  FrontArgs a, b;
  int n_b;
};
__global__ __launch_bounds__(NT) void front_pair_kernel(const FrontPair p) {
  extern __shared__ __attribute__((aligned(16))) char lds_c[];
  if ((int)blockIdx.x < p.n_b) front_body<128, true>(p.b, (int)blockIdx.x, lds_c);
  else front_body<64, false>(p.a, (int)blockIdx.x - p.n_b, lds_c);
}

// tbx_front's checks + the launch descriptor; blocks = its workgroups
static int front_fill(const tbx_front_t* args, FrontArgs& a, int& blocks) {
  if (args == nullptr) return TBX_ERR_ARG;
  const tbx_window_tile_t& w = args->win;
  const tbx_layer_tile_t& t = args->layer;
  // the window part: tbx_window_tile's checks
  if (w.n_groups <= 0 || w.attr == nullptr || w.pe == nullptr || w.row_invalid == nullptr || w.out == nullptr) return TBX_ERR_ARG;
  for (int i = 0; i < 3; ++i)
    if (w.in_images[i] == nullptr || w.pn_images[i] == nullptr) return TBX_ERR_ARG;
  if (w.window <= 0 || w.window > 16 || w.ld_attr % 4 != 0 || w.attr_cols <= 0 || w.attr_cols > 32 || w.attr_cols % 4 != 0 || w.attr_cols > w.ld_attr)
    return TBX_ERR_UNSUPPORTED;
  if (!((w.d_mlp == 64 && w.add_mode == 0) || (w.d_mlp == 128 && w.add_mode == 1))) return TBX_ERR_UNSUPPORTED;
  if (w.drop_thresh != 0u) return TBX_ERR_UNSUPPORTED;  // inference only
  if ((((uintptr_t)w.attr) | ((uintptr_t)w.pe) | ((uintptr_t)w.out)) & 15) return TBX_ERR_ALIGN;
  // the projection part: a first-projection tbx_layer_tile_t on the windows' pooled rows
  if (t.x != w.out || t.n_rows != w.n_groups || t.attn_out != nullptr || t.linear1_image != nullptr || t.proj_n != 3 * D) return TBX_ERR_ARG;
  if (t.proj_image == nullptr || t.qfold_image == nullptr || t.proj_norm_weight == nullptr || t.proj_norm_bias == nullptr || t.proj_out == nullptr ||
      t.ld_proj < 7 * D || t.ld_proj % 4 != 0 || t.drop_thresh != 0u)
    return TBX_ERR_ARG;
  if (((uintptr_t)t.proj_out) & 15) return TBX_ERR_ALIGN;
  if (t.rider_rows < 0) return TBX_ERR_ARG;
  if (t.rider_rows > 0) {
    if ((t.rider_in == nullptr && t.rider_pose3 == nullptr) || t.rider_add == nullptr || t.rider_valid == nullptr || t.rider_out == nullptr)
      return TBX_ERR_ARG;
    if (t.rider_pose3 != nullptr && (t.rider_freqs_xy == nullptr || t.rider_freqs_yaw == nullptr)) return TBX_ERR_ARG;
    for (int i = 0; i < 4; ++i)
      if (t.rider_images[i] == nullptr) return TBX_ERR_ARG;
    if ((((uintptr_t)t.rider_in) | ((uintptr_t)t.rider_add) | ((uintptr_t)t.rider_out)) & 15) return TBX_ERR_ALIGN;
  }
  a.win = w, a.lt = t;
  a.n_win_blocks = (int)((w.n_groups + tbx_window::RT - 1) / tbx_window::RT);
  a.n_rider_blocks = (int)((t.rider_rows + 15) / 16);
  a.n_knn_blocks = 0;
  memset(&a.knn, 0, sizeof(a.knn));
  if (args->n_jobs > 0) {
    const int rc = tbx_knn::knn_multi_fill(args->jobs, args->n_jobs, args->freqs_xy, args->freqs_yaw, args->pe_dim, args->pe, a.knn, a.n_knn_blocks);
    if (rc != TBX_OK) return rc;
    for (int j = 0; j < args->n_jobs; ++j)
      if (args->jobs[j].emb != nullptr) return TBX_ERR_UNSUPPORTED;  // (the relative-pose form only: no embedding phase behind a barrier)
  } else if (args->pe != nullptr) {
    return TBX_ERR_ARG;
  }
  blocks = a.n_win_blocks + a.n_rider_blocks + a.n_knn_blocks;
  return TBX_OK;
}

static bool front_lds_attr() {
  static tbx::PerDeviceOnce lds_attr;  // (per device, thread-safe: tbx_common.h)
  return lds_attr([&] { return !(hipFuncSetAttribute((const void*)front_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)front_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)front_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess); });
}

}  // namespace

extern "C" int tbx_front(const tbx_front_t* args, void* stream) {
  FrontArgs a;
  int blocks = 0;
  const int rc = front_fill(args, a, blocks);
  if (rc != TBX_OK) return rc;
  if (!front_lds_attr()) return TBX_ERR_LAUNCH;
  const dim3 grid((unsigned)blocks);
  if (args->win.d_mlp == 64)
    hipLaunchKernelGGL((front_kernel<64, false>), grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((front_kernel<128, true>), grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_front_pair(const tbx_front_t* agents, const tbx_front_t* lights, void* stream) {
  FrontPair p;
  int na = 0, nb = 0;
  int rc = front_fill(agents, p.a, na);
  if (rc != TBX_OK) return rc;
  rc = front_fill(lights, p.b, nb);
  if (rc != TBX_OK) return rc;
  if (agents->win.d_mlp != 64 || lights->win.d_mlp != 128) return TBX_ERR_UNSUPPORTED;  // ("cat" agents first, "add" lights second)
  if (!front_lds_attr()) return TBX_ERR_LAUNCH;
  p.n_b = nb;
  hipLaunchKernelGGL(front_pair_kernel, dim3((unsigned)(na + nb)), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
