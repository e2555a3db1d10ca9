// tbx_window_tile: the temporal PointNet over the agents' W-step windows (agent_encoder.py:130-159 input encoder in "cat" mode +
// polyline_encoder.py:49-61 + pooling.py:18-19,38) for LARGE launches, on the building blocks of tile_core.h:
//   f = [mlp(attr) (3 layers: d_in -> 64 -> 64 -> 64, relu after the first two) | pose embedding (64)]          per row
//   3 x { h = relu(W f + b) (128 -> 64);  f = [h | max over the window's valid rows of h], invalid rows 0 }   per window
//   out[window] = max over the valid rows of f (= [m | m] of the last layer's maximum m; 0 for a window without a valid row)
// A workgroup (8 waves) owns 2 windows = two 16-row tiles (rows >= W are padding); wave w = output tile w & 3 of the 64 channels
// of row tile w >> 2. In the transposed product a lane holds 4 channels of ONE row and the 16 rows of a window are the 16 lanes
// of a DPP row: the masked maximum over a window is 4 DPP steps in registers - no LDS pass, no GROUPMAX stage.
// Replaces a ~20-stage grouped tbx_rowchain program (48-row tiles, 158-167 us at 4096 windows).
#include "window_core.h"

using namespace tbx_tile;

namespace {

using namespace tbx_window;

struct WindowArgs {
  tbx_window_tile_t t;
};

template <int DM, bool ADD>
__global__ __launch_bounds__(NT) void tile_window_kernel(const WindowArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds_c[];
  window_body<DM, ADD>(a.t, (int)blockIdx.x, lds_c, nullptr);
}

}  // namespace

extern "C" int TBX_TILE_ENTRY(tbx_window_tile)(const tbx_window_tile_t* args, void* stream) {
  if (args == nullptr || args->n_groups <= 0) return TBX_ERR_ARG;
  const tbx_window_tile_t& t = *args;
  if (t.attr == nullptr || t.pe == nullptr || t.row_invalid == nullptr || t.out == nullptr) return TBX_ERR_ARG;
  for (int i = 0; i < 3; ++i)
    if (t.in_images[i] == nullptr || t.pn_images[i] == nullptr) return TBX_ERR_ARG;
  if (t.window <= 0 || t.window > 16 || t.ld_attr % 4 != 0 || t.attr_cols <= 0 || t.attr_cols > 32 || t.attr_cols % 4 != 0 ||
      t.attr_cols > t.ld_attr)
    return TBX_ERR_UNSUPPORTED;
  if (!((t.d_mlp == 64 && t.add_mode == 0) || (t.d_mlp == 128 && t.add_mode == 1))) return TBX_ERR_UNSUPPORTED;
  if ((((uintptr_t)t.attr) | ((uintptr_t)t.pe) | ((uintptr_t)t.out)) & 15) return TBX_ERR_ALIGN;
  if (t.drop_thresh != 0u && t.drop_seed == nullptr) return TBX_ERR_ARG;
  WindowArgs a;
  a.t = t;
  static tbx::PerDeviceOnce lds_attr;  // (per device, thread-safe: tbx_common.h)
  if (!lds_attr([&] { return !(hipFuncSetAttribute((const void*)tile_window_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)tile_window_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess); })) return TBX_ERR_LAUNCH;
  const dim3 grid((unsigned)((t.n_groups + RT - 1) / RT));
  if (t.d_mlp == 64)
    hipLaunchKernelGGL((tile_window_kernel<64, false>), grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((tile_window_kernel<128, true>), grid, dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
