// Per-step feature preparation kernels: tbx_agent_prep, tbx_tl_prep, tbx_map_prep (see include/tbx_hip.h).
// Small, latency-bound elementwise work; one wavefront per agent / one thread per light step or polyline node,
// written so that a step needs no host-side branching (graph-capturable).
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"
#include "step_core.h"

namespace {

using tbx_step::AgentPrepArgs;
using tbx_step::to_local;

// One workgroup (4 wavefronts) per agent: the window's steps are dealt to the wavefronts, the last valid step comes from one
// ballot over the validity bytes (the kernel opens every agent step: 14 -> ~6 us at 64 agents).
__global__ __launch_bounds__(256) void agent_prep_kernel(const AgentPrepArgs a) { tbx_step::agent_prep(a, (int)blockIdx.x, (int)threadIdx.x, 256); }
// ... and a WAVEFRONT per agent (its two 32-lane slots walk the window) for launches of many agents: at 4096 agents the form above is
// 4096 workgroups of a few hundred instructions each - bound by the rate workgroups are dispatched at (14.3 us; this form: see
// DESIGN.md 5). Same per-(agent, step) arithmetic: bit-identical rows.
__global__ __launch_bounds__(256) void agent_prep_wave_kernel(const AgentPrepArgs a) {
  tbx_step::agent_prep(a, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), 64);
}

__global__ void tl_prep_kernel(const uint8_t* __restrict__ hist_tl, const uint8_t* __restrict__ tl_invalid, int n_tok,
                               int window, int ld_attr, float* __restrict__ attr, uint8_t* __restrict__ row_invalid) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (token, step) row
  if (r >= (int64_t)n_tok * window) return;
  const int l = (int)(r / window), w = (int)(r - (int64_t)l * window);
  const uint8_t st = hist_tl[r];
  const bool missing = st == 0xFF;
  for (int c = 0; c < ld_attr; ++c) {
    float v = 0.f;
    if (c < 5)
      v = (!missing && ((st >> c) & 1)) ? 1.f : 0.f;
    else if (c - 5 == w)
      v = 1.f;
    attr[r * ld_attr + c] = v;
  }
  row_invalid[r] = (missing || tl_invalid[l] != 0) ? 1 : 0;
}

__global__ void map_prep_kernel(const uint8_t* __restrict__ mp_valid, const float* __restrict__ mp_type11,
                                const float* __restrict__ mp_pose, int n_pl, int n_node, float* __restrict__ attr,
                                float* __restrict__ pe, uint8_t* __restrict__ row_invalid, float* __restrict__ tok_pose,
                                uint8_t* __restrict__ tok_invalid) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (polyline, node) row
  if (r >= (int64_t)n_pl * n_node) return;
  const int m = (int)(r / n_node), i = (int)(r - (int64_t)m * n_node);
  const float* p0 = mp_pose + (int64_t)m * n_node * 3;
  const float x0 = p0[0], y0 = p0[1], yaw0 = p0[2];
  const float c = cosf(yaw0), s = sinf(yaw0);
  float px, py;
  to_local(x0, y0, c, s, p0[i * 3], p0[i * 3 + 1], px, py);
  const float yaw = __fsub_rn(p0[i * 3 + 2], yaw0);
  // 7-d MultiPath++ polyline feature (pose_emb.py:58-89) with the unit heading as the segment vector
  const float dx = cosf(yaw), dy = sinf(yaw);
  const float eps = FLT_EPSILON;
  const float proj = (-px * dx + -py * dy) / (dx * dx + dy * dy + eps);
  const float cl = fminf(fmaxf(proj, 0.f), 1.f);
  const float cx = px + cl * dx, cy = py + cl * dy;
  const float rn = sqrtf(cx * cx + cy * cy);
  const float dn = sqrtf(dx * dx + dy * dy);
  const float ex = px + dx - cx, ey = py + dy - cy;
  float* f = pe + r * 8;
  f[0] = rn;
  f[1] = cx / (rn + eps);
  f[2] = cy / (rn + eps);
  f[3] = dx / (dn + eps);
  f[4] = dy / (dn + eps);
  f[5] = dn;
  f[6] = sqrtf(ex * ex + ey * ey);
  f[7] = 0.f;
  float* at = attr + r * 32;
  for (int k = 0; k < 32; ++k) {
    float v = 0.f;
    if (k < 11)
      v = mp_type11[(int64_t)m * 11 + k];
    else if (k - 11 == i)
      v = 1.f;
    at[k] = v;
  }
  row_invalid[r] = mp_valid[r] ? 0 : 1;
  if (i == 0) {
    tok_pose[m * 3] = x0;
    tok_pose[m * 3 + 1] = y0;
    tok_pose[m * 3 + 2] = yaw0;
    tok_invalid[m] = mp_valid[r] ? 0 : 1;
  }
}

}  // namespace

extern "C" int tbx_agent_prep(const uint8_t* hist_valid, const float* hist_pose, const float* hist_motion,
                              const float* ag_attr6, const uint8_t* ag_type_idx, int n_batch, int n_ag, int window,
                              const float* freqs_xy, const float* freqs_yaw, int pe_dim, float* tok_pose,
                              uint8_t* tok_invalid, float* attr, float* pe, uint8_t* row_invalid, uint8_t* type_mask,
                              const int64_t* dest, const float* mp_tok_pose, int n_mp, int mp_batch_div, float* navi_pose3,
                              int32_t* navi_row, void* stream) {
  if (!hist_valid || !hist_pose || !hist_motion || !ag_attr6 || !freqs_xy || !freqs_yaw || !tok_pose || !tok_invalid ||
      !attr || !pe || !row_invalid)
    return TBX_ERR_ARG;
  if (n_batch <= 0 || n_ag <= 0 || window <= 0 || window > 23 || (pe_dim != 64 && pe_dim != 128)) return TBX_ERR_UNSUPPORTED;
  if (type_mask != nullptr && !ag_type_idx) return TBX_ERR_ARG;
  if (dest != nullptr && (!mp_tok_pose || !navi_pose3 || !navi_row || n_mp <= 0 || mp_batch_div <= 0)) return TBX_ERR_ARG;
  AgentPrepArgs a{hist_valid, hist_pose, hist_motion, ag_attr6, ag_type_idx, freqs_xy, freqs_yaw, tok_pose, tok_invalid,
                  attr, pe, row_invalid, type_mask, dest, mp_tok_pose, navi_pose3, navi_row, n_batch * n_ag, n_ag, window,
                  pe_dim, n_mp, mp_batch_div};
  static const int wave_rows = [] {
    const char* e = getenv("TBX_PREP_WAVE_ROWS");
    return e && atoi(e) > 0 ? atoi(e) : 512;
  }();
  if (a.n_tok >= wave_rows)
    hipLaunchKernelGGL(agent_prep_wave_kernel, dim3((a.n_tok + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(agent_prep_kernel, dim3(a.n_tok), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_tl_prep(const uint8_t* hist_tl, const uint8_t* tl_invalid, int n_batch, int n_tl, int window,
                           int ld_attr, float* attr, uint8_t* row_invalid, void* stream) {
  if (!hist_tl || !tl_invalid || !attr || !row_invalid || n_batch <= 0 || n_tl <= 0 || window <= 0) return TBX_ERR_ARG;
  if (ld_attr < 5 + window || ld_attr % 4) return TBX_ERR_UNSUPPORTED;
  const int64_t rows = (int64_t)n_batch * n_tl * window;
  hipLaunchKernelGGL(tl_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hist_tl,
                     tl_invalid, n_batch * n_tl, window, ld_attr, attr, row_invalid);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_map_prep(const uint8_t* mp_valid, const float* mp_type11, const float* mp_pose, int n_batch, int n_mp,
                            int n_node, float* attr, float* pe, uint8_t* row_invalid, float* tok_pose,
                            uint8_t* tok_invalid, void* stream) {
  if (!mp_valid || !mp_type11 || !mp_pose || !attr || !pe || !row_invalid || !tok_pose || !tok_invalid) return TBX_ERR_ARG;
  if (n_batch <= 0 || n_mp <= 0 || n_node <= 0 || n_node > 21) return TBX_ERR_UNSUPPORTED;
  const int64_t rows = (int64_t)n_batch * n_mp * n_node;
  hipLaunchKernelGGL(map_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mp_valid,
                     mp_type11, mp_pose, n_batch * n_mp, n_node, attr, pe, row_invalid, tok_pose, tok_invalid);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
