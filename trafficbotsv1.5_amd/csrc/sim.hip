// tbx_sim_step: one closed-loop simulation step for every agent and traffic light of every scene
// (see include/tbx_hip.h; restates dynamics.py / teacher_forcing.py / the feeding-back subset of
// traffic_rule_checker.py as the oracle's Sim.rollout does). Elementwise, one thread per agent / light; the 1-based
// step index lives in device memory and is bumped by a second single-thread kernel so that one captured graph replays
// every step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"
#include "step_core.h"

namespace {

using tbx_step::LPA;

// Agents: 32 lanes (half a wavefront) per agent - every lane repeats the agent's scalar dynamics (broadcast loads), the
// lanes split the destination polyline's nodes and the window shift, so the kernel is 2-3 dependent loads deep instead
// of ~30 (it closes the critical path of every step: 23 -> ~7 us). Lights: one thread per light.

using tbx_step::LPT;
using tbx_step::TlPrepArgs;

__global__ void sim_step_kernel(const tbx_sim_state_t s, const int parts, const TlPrepArgs tp) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i_ag = gid / LPA, sub = gid % LPA;
  const int t = *s.step;  // step being simulated: model saw the state of step t-1
  const int n_ag_tot = s.n_batch * s.n_ag;
  const int n_tl_tot = s.n_batch * s.n_tl;
  if ((parts & TBX_SIM_AGENTS) && i_ag < n_ag_tot) tbx_step::sim_agent(s, parts, t, i_ag, sub, (int)(threadIdx.x & 32));
  // LPT lanes per light (tbx_step::sim_light): every lane repeats the light's few scalar operations, the lanes split the window's W
  // entries - the shift and, with tbx_tl_prep riding, the W attribute rows (one thread per light wrote W x ld_attr floats one by one:
  // 15 us for 128 lights, on the lights' stream of every step)
  if ((parts & TBX_SIM_LIGHTS) && gid / LPT < n_tl_tot) tbx_step::sim_light(s, parts, t, gid / LPT, gid % LPT, tp);
  if (parts & TBX_SIM_ADVANCE) tbx_step::sim_advance(s, t, gridDim.x);
}

__global__ void sim_bump_kernel(int32_t* step) { *step += 1; }

// TBX_SIM_APPEND: TrafficBots._append_hist (traffic_bots.py:123-143) on the current state - a thread per agent / light.
__global__ void sim_append_kernel(const tbx_sim_state_t s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = s.window;
  if (i < s.n_batch * s.n_ag) {
    uint8_t* hv = s.hist_valid + (int64_t)i * W;
    float* hp = s.hist_pose + (int64_t)i * W * 3;
    float* hm = s.hist_motion + (int64_t)i * W * 3;
    for (int w = 0; w < W - 1; ++w) {
      hv[w] = hv[w + 1];
      for (int c = 0; c < 3; ++c) hp[w * 3 + c] = hp[(w + 1) * 3 + c], hm[w * 3 + c] = hm[(w + 1) * 3 + c];
    }
    hv[W - 1] = s.ag_valid[i];
    for (int c = 0; c < 3; ++c) hp[(W - 1) * 3 + c] = s.ag_pose[i * 3 + c], hm[(W - 1) * 3 + c] = s.ag_motion[i * 3 + c];
  }
  if (i < s.n_batch * s.n_tl) {
    uint8_t* ht = s.hist_tl + (int64_t)i * W;
    for (int w = 0; w < W - 1; ++w) ht[w] = ht[w + 1];
    ht[W - 1] = s.tl_state[i];
  }
}

}  // namespace

extern "C" int tbx_sim_step(const tbx_sim_state_t* st, void* stream) {
  return tbx_sim_step_parts(st, TBX_SIM_AGENTS | TBX_SIM_LIGHTS | TBX_SIM_ADVANCE, stream);
}

extern "C" int tbx_sim_step_parts(const tbx_sim_state_t* st, int parts, void* stream) {
  return tbx_sim_step_tl_prep(st, parts, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int tbx_sim_step_tl_prep(const tbx_sim_state_t* st, int parts, const uint8_t* tl_invalid, int ld_attr, float* attr,
                                    uint8_t* row_invalid, void* stream) {
  TlPrepArgs tp{nullptr, nullptr, nullptr, 0};
  if (tl_invalid != nullptr) {
    if (!st || !(parts & TBX_SIM_LIGHTS) || (parts & TBX_SIM_NO_APPEND) || !attr || !row_invalid || ld_attr < 5 + st->window) return TBX_ERR_ARG;
    if ((ld_attr % 4) || (((uintptr_t)attr) & 15)) return TBX_ERR_ALIGN;  // rows are written as float4
    tp = TlPrepArgs{tl_invalid, attr, row_invalid, ld_attr};
  }
  const int known = TBX_SIM_AGENTS | TBX_SIM_LIGHTS | TBX_SIM_ADVANCE | TBX_SIM_NO_DISABLE | TBX_SIM_NO_APPEND | TBX_SIM_APPEND;
  if (!st || (parts & ~known) || parts == 0) return TBX_ERR_ARG;
  if ((parts & TBX_SIM_APPEND) && parts != TBX_SIM_APPEND) return TBX_ERR_ARG;  // a part of its own
  if ((parts & (TBX_SIM_NO_DISABLE | TBX_SIM_NO_APPEND)) && !(parts & (TBX_SIM_AGENTS | TBX_SIM_LIGHTS))) return TBX_ERR_ARG;
  const tbx_sim_state_t& s = *st;
  if (s.n_batch <= 0 || s.n_ag <= 0 || s.n_tl <= 0 || s.window <= 0 || s.n_step_out <= 0 || s.n_node <= 0) return TBX_ERR_ARG;
  const void* need[] = {s.step, s.ag_valid, s.ag_disabled, s.ag_pose, s.ag_motion, s.navi_valid, s.outside_map,
                        s.dest_reached, s.tl_state, s.hist_valid, s.hist_pose, s.hist_motion, s.hist_tl, s.ag_type_idx,
                        s.tf_mask, s.gt_valid, s.gt_pose, s.gt_motion, s.tl_gt, s.boundary, s.dest_pos, s.dest_dir,
                        s.dest_invalid, s.dest_kind, s.dest_thresh, s.action_mean, s.tl_logits, s.out_valid, s.out_pose,
                        s.out_motion, s.out_action, s.out_tl_state, s.out_outside_map, s.out_dest_reached};
  for (const void* p : need)
    if (p == nullptr) return TBX_ERR_ARG;
  if (s.player_valid != nullptr && s.player_action == nullptr) return TBX_ERR_ARG;
  if (s.ov_valid != nullptr && (!s.ov_pose || !s.ov_motion || !s.ov_tl_valid || !s.ov_tl_state)) return TBX_ERR_ARG;
  if (parts == TBX_SIM_APPEND) {
    const int64_t na = (int64_t)s.n_batch * (s.n_ag > s.n_tl ? s.n_ag : s.n_tl);
    hipLaunchKernelGGL(sim_append_kernel, dim3((unsigned)((na + 127) / 128)), dim3(128), 0, (hipStream_t)stream, s);
    return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
  }
  const int64_t th_ag = (parts & TBX_SIM_AGENTS) ? (int64_t)s.n_batch * s.n_ag * LPA : 0;
  const int64_t th_tl = (parts & TBX_SIM_LIGHTS) ? (int64_t)s.n_batch * s.n_tl * LPT : 0;
  const int64_t n = th_ag > th_tl ? th_ag : th_tl;
  hipStream_t hs = (hipStream_t)stream;
  if (parts & (TBX_SIM_AGENTS | TBX_SIM_LIGHTS)) {
    // TBX_SIM_ADVANCE inside the launch = one arrival per workgroup on ONE counter: same-address atomics serialise in L2 at ~10 ns
    // each (measured at 32 x 128 agents: 1024 arrivals = ~8 us of a 16 us launch). Large grids advance by a one-thread launch
    // behind the step instead (stream order: every workgroup has read *step by then).
    // ... unless workgroups of up to 1024 threads bring the grid down to 256 arrivals (4096 agents: 256 workgroups of 512): the extra
    // launch is ~5 us + its gaps on the critical path of every step at the WOSAC shape.
    unsigned bs = 128;
    if (parts & TBX_SIM_ADVANCE)
      while (bs < 1024 && (n + bs - 1) / bs > 256) bs *= 2;
    const unsigned blocks = (unsigned)((n + bs - 1) / bs);
    const bool bump_after = (parts & TBX_SIM_ADVANCE) && blocks > 256;
    hipLaunchKernelGGL(sim_step_kernel, dim3(blocks), dim3(bs), 0, hs, s, bump_after ? (parts & ~TBX_SIM_ADVANCE) : parts, tp);
    if (bump_after) hipLaunchKernelGGL(sim_bump_kernel, dim3(1), dim3(1), 0, hs, s.step);
  } else {
    hipLaunchKernelGGL(sim_bump_kernel, dim3(1), dim3(1), 0, hs, s.step);
  }
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

// utils/rewards.py:35-85 (DifferentiableReward.get, default configuration: imitation terms, w_collision = 0) for ONE step as a
// call of its own: the expressions of tbx_sim_step's log (step_core.h) on caller-supplied prediction / ground truth.
namespace {
using tbx_step::sim_sl1;
__global__ __launch_bounds__(256) void diffbar_reward_kernel(const uint8_t* __restrict__ pred_valid, const float* __restrict__ pred_pose,
                                                             const float* __restrict__ pred_motion, const uint8_t* __restrict__ gt_valid,
                                                             const float* __restrict__ gt_pose, const float* __restrict__ gt_motion,
                                                             int64_t n, float w_pos, float w_rot, float w_spd, float* __restrict__ out4,
                                                             uint8_t* __restrict__ out_valid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool r_valid = pred_valid[i] != 0;
  float r_pos = 0.f, r_rot = 0.f, r_spd = 0.f;
  if (gt_valid != nullptr) {
    r_valid = r_valid && gt_valid[i] != 0;
    if (r_valid) {
      r_pos = -w_pos * (sim_sl1(gt_pose[i * 3] - pred_pose[i * 3]) + sim_sl1(gt_pose[i * 3 + 1] - pred_pose[i * 3 + 1]));
      r_rot = -w_rot * (0.5f * (1.f - cosf(gt_pose[i * 3 + 2] - pred_pose[i * 3 + 2])));
      r_spd = -w_spd * sim_sl1(gt_motion[i * 3] - pred_motion[i * 3]);
    }
  }
  out4[i * 4] = r_pos, out4[i * 4 + 1] = r_rot, out4[i * 4 + 2] = r_spd;
  out4[i * 4 + 3] = (r_pos + r_rot) + r_spd;
  out_valid[i] = r_valid ? 1 : 0;
}
}  // namespace

extern "C" int tbx_diffbar_reward(const uint8_t* pred_valid, const float* pred_pose, const float* pred_motion, const uint8_t* gt_valid,
                                  const float* gt_pose, const float* gt_motion, int64_t n, float w_pos, float w_rot, float w_spd,
                                  float* out4, uint8_t* out_valid, void* stream) {
  if (!pred_valid || !pred_pose || !pred_motion || !out4 || !out_valid || n <= 0) return TBX_ERR_ARG;
  if (gt_valid != nullptr && (!gt_pose || !gt_motion)) return TBX_ERR_ARG;
  hipLaunchKernelGGL(diffbar_reward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred_valid, pred_pose,
                     pred_motion, gt_valid, gt_pose, gt_motion, n, w_pos, w_rot, w_spd, out4, out_valid);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
