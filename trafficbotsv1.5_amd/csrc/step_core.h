// Per-agent bodies of the step's two small kernels - tbx_sim_step's agents' part (csrc/sim.hip) and tbx_agent_prep (csrc/prep.hip) - as
// device functions, so that the launch that produces an agent's action (the last decoder layer with the heads, csrc/dec_mid.hip) can
// run the agent's simulation step and the next step's feature preparation in its own tail instead of two more launches on the
// critical path of every step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace tbx_step {

constexpr int LPA = 32;  // lanes per agent in sim_agent

__device__ __forceinline__ float sim_sl1(float d) {  // F.smooth_l1_loss, beta = 1
  const float a = fabsf(d);
  return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

// One closed-loop step of agent i by 32 lanes (`sub` = 0..31; half_shift = 0 / 32: which half of the wavefront's ballot is theirs),
// t = *s.step read by the caller: every lane repeats the agent's scalar dynamics (broadcast loads), the lanes split the destination
// polyline's nodes and the window shift.
__device__ __forceinline__ void sim_agent(const tbx_sim_state_t& s, const int parts, const int t, const int i, const int sub, const int half_shift) {
  const int T = s.n_step_out;
  const int W = s.window;
  const int b = i / s.n_ag;
  const bool valid0 = s.ag_valid[i] != 0;
  const int ty = s.ag_type_idx[i];
  float px = s.ag_pose[i * 3], py = s.ag_pose[i * 3 + 1], pyaw = s.ag_pose[i * 3 + 2];
  float spd = s.ag_motion[i * 3];
  // Dynamics.update_ag + MultiPathPP (dynamics.py:84-120,237-274)
  float acc = 0.f, yr = 0.f;
  if (valid0) {
    acc = tanhf(s.action_mean[i * 2]) * s.max_acc[ty];
    yr = tanhf(s.action_mean[i * 2 + 1]) * s.max_yaw_rate[ty];
    if (s.player_valid != nullptr && s.player_valid[i] != 0) {  // player-controlled agent (dynamics.py:104-107)
      acc = s.player_action[i * 2];
      yr = s.player_action[i * 2 + 1];
    }
  }
  const float half_dt = 0.5f * s.dt;
  const float v_t = spd + half_dt * acc;
  const float th_t = pyaw + half_dt * yr;
  float nx = px + s.dt * (v_t * cosf(th_t));
  float ny = py + s.dt * (v_t * sinf(th_t));
  float nyaw = pyaw + s.dt * yr;
  float nspd = spd + s.dt * acc, nacc = acc, nyr = yr;
  if (!valid0) nx = ny = nyaw = nspd = nacc = nyr = 0.f;
  const float qx = nx, qy = ny, qyaw = nyaw, qspd = nspd;  // the prediction (before the override), for the reward
  if (t - 1 < T && sub == 0) {
    const int64_t o = (int64_t)i * T + (t - 1);
    s.out_valid[o] = valid0 ? 1 : 0;
    s.out_pose[o * 3] = nx;
    s.out_pose[o * 3 + 1] = ny;
    s.out_pose[o * 3 + 2] = nyaw;
    s.out_motion[o * 3] = nspd;
    s.out_motion[o * 3 + 1] = nacc;
    s.out_motion[o * 3 + 2] = nyr;
    s.out_action[o * 2] = acc;
    s.out_action[o * 2 + 1] = yr;
  }
  // outside-map / destination-reached on the predicted (pre-override) state (traffic_rule_checker.py:109-120,300-330)
  const float* bd = s.boundary + b * 4;
  const bool out_now = valid0 && (nx > bd[1] || nx < bd[0] || ny > bd[3] || ny < bd[2]);
  const bool outside = (s.outside_map[i] != 0) || out_now;
  bool pos_ok = false, rot_ok = false;
  const float hx = cosf(nyaw), hy = sinf(nyaw);
  for (int k = sub; k < s.n_node; k += LPA) {
    const int64_t d = (int64_t)i * s.n_node + k;
    const bool ok = s.dest_invalid[d] == 0;
    const float ex = nx - s.dest_pos[d * 2], ey = ny - s.dest_pos[d * 2 + 1];
    pos_ok = pos_ok || (ok && sqrtf(ex * ex + ey * ey) < s.dest_thresh[i]);
    rot_ok = rot_ok || (ok && hx * s.dest_dir[d * 2] + hy * s.dest_dir[d * 2 + 1] > 0.8660254037844387f);
  }
  {  // any() over the agent's 32 lanes (its half of the wavefront's ballot)
    const int sh = half_shift;
    pos_ok = ((__ballot(pos_ok) >> sh) & 0xffffffffull) != 0ull;
    rot_ok = ((__ballot(rot_ok) >> sh) & 0xffffffffull) != 0ull;
  }
  const uint8_t kind = s.dest_kind[i];
  const bool reached0 = s.dest_reached[i] != 0;
  const bool reach_now = !reached0 && valid0 && (((kind & 1) && pos_ok && rot_ok) || ((kind & 2) && pos_ok));
  const bool reached = reached0 || reach_now;
  if (t - 1 < T && sub == 0) {
    s.out_outside_map[(int64_t)i * T + (t - 1)] = outside ? 1 : 0;
    s.out_dest_reached[(int64_t)i * T + (t - 1)] = reached ? 1 : 0;
  }
  // TeacherForcing.get + Dynamics.override_ag (teacher_forcing.py:128-147, dynamics.py:122-141)
  bool valid = valid0;
  bool disabled = s.ag_disabled[i] != 0;
  bool has_gt = t < s.n_step_gt;
  bool gt_v = false, tf_now = false;
  const int64_t g = (int64_t)i * s.n_step_gt + t;
  if (has_gt) gt_v = s.gt_valid[g] != 0;
  if (s.ov_valid != nullptr) {  // the step's ag_override handed over explicitly (WaymoMotion.forward)
    tf_now = s.ov_valid[i] != 0;
    if (tf_now && !disabled) {
      valid = true;
      nx = s.ov_pose[i * 3], ny = s.ov_pose[i * 3 + 1], nyaw = s.ov_pose[i * 3 + 2];
      nspd = s.ov_motion[i * 3], nacc = s.ov_motion[i * 3 + 1], nyr = s.ov_motion[i * 3 + 2];
    }
  } else if (has_gt) {
    tf_now = s.tf_mask[g] != 0;
    if (tf_now && !disabled) {
      valid = true;
      nx = s.gt_pose[g * 3];
      ny = s.gt_pose[g * 3 + 1];
      nyaw = s.gt_pose[g * 3 + 2];
      nspd = s.gt_motion[g * 3];
      nacc = s.gt_motion[g * 3 + 1];
      nyr = s.gt_motion[g * 3 + 2];
    }
  }
  if (t - 1 < T && sub == 0) {
    const int64_t o = (int64_t)i * T + (t - 1);
    if (s.out_tf != nullptr) s.out_tf[o] = tf_now ? 1 : 0;
    // DifferentiableReward.get on the prediction (rewards.py:58-74; the same expressions as tbx_train_chain_fwd)
    if (s.out_reward != nullptr) {
      float r_pos = 0.f, r_rot = 0.f, r_spd = 0.f;
      bool r_valid = valid0;
      if (has_gt) {
        r_valid = valid0 && gt_v;
        if (r_valid) {
          r_pos = -s.w_pos * (sim_sl1(s.gt_pose[g * 3] - qx) + sim_sl1(s.gt_pose[g * 3 + 1] - qy));
          r_rot = -s.w_rot * (0.5f * (1.f - cosf(s.gt_pose[g * 3 + 2] - qyaw)));
          r_spd = -s.w_spd * sim_sl1(s.gt_motion[g * 3] - qspd);
        }
      }
      s.out_reward[o * 4] = r_pos, s.out_reward[o * 4 + 1] = r_rot, s.out_reward[o * 4 + 2] = r_spd;
      s.out_reward[o * 4 + 3] = (r_pos + r_rot) + r_spd;
      if (s.out_reward_valid != nullptr) s.out_reward_valid[o] = r_valid ? 1 : 0;
    }
  }
  // Dynamics.disable_ag / disable_navi (dynamics.py:165-204); a step-wise caller does both itself from now_*
  const bool no_disable = (parts & TBX_SIM_NO_DISABLE) != 0;
  const bool dis = !no_disable && out_now && !(has_gt && gt_v);
  disabled = disabled || dis;
  valid = valid && !dis;
  if (sub == 0 && s.now_outside != nullptr) s.now_outside[i] = out_now ? 1 : 0;
  if (sub == 0 && s.now_reached != nullptr) s.now_reached[i] = reach_now ? 1 : 0;
  // TrafficBots._append_hist (traffic_bots.py:123-143): slide the window, append the state the next step will see. Lane w
  // moves entry w + 1 to w: the wavefront runs in lockstep, so every lane has loaded before any lane stores (chunks of 32
  // go upwards, each reads only entries no earlier chunk wrote).
  uint8_t* hv = s.hist_valid + (int64_t)i * W;
  float* hp = s.hist_pose + (int64_t)i * W * 3;
  float* hm = s.hist_motion + (int64_t)i * W * 3;
  const bool append = (parts & TBX_SIM_NO_APPEND) == 0;
  for (int w0 = 0; append && w0 < W - 1; w0 += LPA) {
    const int w = w0 + sub;
    const bool mv = w < W - 1;
    uint8_t v1 = 0;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (mv) {
      v1 = hv[w + 1];
      p0 = hp[(w + 1) * 3], p1 = hp[(w + 1) * 3 + 1], p2 = hp[(w + 1) * 3 + 2];
      m0 = hm[(w + 1) * 3], m1 = hm[(w + 1) * 3 + 1], m2 = hm[(w + 1) * 3 + 2];
    }
    __builtin_amdgcn_wave_barrier();
    if (mv) {
      hv[w] = v1;
      hp[w * 3] = p0, hp[w * 3 + 1] = p1, hp[w * 3 + 2] = p2;
      hm[w * 3] = m0, hm[w * 3 + 1] = m1, hm[w * 3 + 2] = m2;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (sub == 0) {
    s.ag_valid[i] = valid ? 1 : 0;
    s.ag_disabled[i] = disabled ? 1 : 0;
    s.ag_pose[i * 3] = nx;
    s.ag_pose[i * 3 + 1] = ny;
    s.ag_pose[i * 3 + 2] = nyaw;
    s.ag_motion[i * 3] = nspd;
    s.ag_motion[i * 3 + 1] = nacc;
    s.ag_motion[i * 3 + 2] = nyr;
    s.outside_map[i] = outside ? 1 : 0;
    s.dest_reached[i] = reached ? 1 : 0;
    if (reach_now && !no_disable) s.navi_valid[i] = 0;
    if (append) {
      hv[W - 1] = valid ? 1 : 0;
      hp[(W - 1) * 3] = nx;
      hp[(W - 1) * 3 + 1] = ny;
      hp[(W - 1) * 3 + 2] = nyaw;
      hm[(W - 1) * 3] = nspd;
      hm[(W - 1) * 3 + 1] = nacc;
      hm[(W - 1) * 3 + 2] = nyr;
    }
  }
}

// TBX_SIM_ADVANCE next to a part: every thread of this workgroup has read *step; the last of the n_wg workgroups to arrive advances it
// (No __threadfence: the barrier below already waits for every thread's load of *step to have RETURNED (s_waitcnt vmcnt(0) in front of
// s_barrier), the arrival counter is a device-scope atomic, and the last arriver's two stores are read by later launches only. An
// agent-scope fence here is an L2 write-back + invalidate per workgroup: ~3.5 us each, measured - most of this kernel's time.)
__device__ __forceinline__ void sim_advance(const tbx_sim_state_t& s, const int t, const unsigned n_wg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd((unsigned*)(s.step + 1), 1u);
    if (prev == n_wg - 1) {
      s.step[1] = 0;
      s.step[0] = t + 1;
    }
  }
}

typedef tbx_agent_prep_args_t AgentPrepArgs;  // (field order = tbx_agent_prep's parameter groups)

__device__ __forceinline__ void to_local(float x0, float y0, float c, float s, float x, float y, float& rx, float& ry) {
  const float dx = __fsub_rn(x, x0), dy = __fsub_rn(y, y0);
  rx = __fadd_rn(__fmul_rn(dx, c), __fmul_rn(dy, s));
  ry = __fadd_rn(__fmul_rn(dx, -s), __fmul_rn(dy, c));
}

// tbx_agent_prep for agent i by 256 threads (tid = 0..255; 4 wavefronts): the window's steps are dealt to the wavefronts, the last valid
// step comes from one ballot over the validity bytes.
__device__ __forceinline__ void agent_prep(const AgentPrepArgs& a, const int i, const int tid) {
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (i >= a.n_tok) return;
  const int W = a.window;  // <= 23 (attribute row: 9 + W <= 32)
  const uint8_t* hv = a.hist_valid + (int64_t)i * W;
  const float* hp = a.hist_pose + (int64_t)i * W * 3;
  const float* hm = a.hist_motion + (int64_t)i * W * 3;
  const unsigned long long vmask = __ballot(lane < W && hv[lane < W ? lane : 0] != 0);
  const int last = vmask ? 63 - __builtin_clzll(vmask) : -1;
  float x0 = 0.f, y0 = 0.f, yaw0 = 0.f;
  if (last >= 0) {
    x0 = hp[last * 3];
    y0 = hp[last * 3 + 1];
    yaw0 = hp[last * 3 + 2];
  }
  if (tid == 0) {
    a.tok_pose[i * 3] = x0;
    a.tok_pose[i * 3 + 1] = y0;
    a.tok_pose[i * 3 + 2] = yaw0;
    a.tok_invalid[i] = last < 0 ? 1 : 0;
  }
  const float c = cosf(yaw0), s = sinf(yaw0);
  for (int w = wave; w < W; w += 4) {
    const int64_t r = (int64_t)i * W + w;
    float rx, ry;
    to_local(x0, y0, c, s, hp[w * 3], hp[w * 3 + 1], rx, ry);
    const float ryaw = __fsub_rn(hp[w * 3 + 2], yaw0);
    tbx::pose_emb_write(a.pe + r * a.pe_dim, a.pe_dim, rx, ry, ryaw, a.freqs_xy, a.freqs_yaw, lane, 64);
    if (lane < 32) {
      float v = 0.f;
      if (lane < 6)
        v = a.ag_attr6[(int64_t)i * 6 + lane];
      else if (lane < 9)
        v = hm[w * 3 + lane - 6];
      else if (lane - 9 == w)
        v = 1.f;
      a.attr[r * 32 + lane] = v;
    }
    if (lane == 32) a.row_invalid[r] = ((vmask >> w) & 1ull) ? 0 : 1;
  }
  if (wave != 0) return;
  if (a.type_mask != nullptr && lane < 3) {
    const bool now = hv[W - 1] != 0;
    a.type_mask[(int64_t)lane * a.n_tok + i] = (now && a.ag_type_idx[i] == lane) ? 0 : 1;
  }
  if (a.dest != nullptr && lane == 0) {
    const int b = i / a.n_ag;
    const int64_t mrow = (int64_t)(b / a.mp_batch_div) * a.n_mp + a.dest[i];
    const float ax = hp[(W - 1) * 3], ay = hp[(W - 1) * 3 + 1], ayaw = hp[(W - 1) * 3 + 2];
    const float cc = cosf(ayaw), ss = sinf(ayaw);
    float rx, ry;
    to_local(ax, ay, cc, ss, a.mp_tok_pose[mrow * 3], a.mp_tok_pose[mrow * 3 + 1], rx, ry);
    a.navi_pose3[i * 3] = rx;
    a.navi_pose3[i * 3 + 1] = ry;
    a.navi_pose3[i * 3 + 2] = __fsub_rn(a.mp_tok_pose[mrow * 3 + 2], ayaw);
    a.navi_row[i] = (int32_t)mrow;
  }
}

}  // namespace tbx_step
