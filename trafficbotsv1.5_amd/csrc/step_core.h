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
// Shape of the body: EVERY load first (one round trip to memory - two for the type's limits - instead of one per section: with the
// log's stores between the sections the compiler may not move a later section's loads above them, and in the tail of the last
// decoder layer's launch each round trip is ~0.8 us of the step's critical path), then the arithmetic, then every store.
// Everything sim_agent reads EXCEPT the step's action, as a value: a caller that produces the action itself (the last decoder layer's
// launch, csrc/dec_layer_mf.inc) requests these a few stages before its heads finish - the loads (two dependent round trips: the
// agent's type, then that type's limits) then fly under the heads instead of behind them.
struct SimLoads {
  bool valid0, player, outside0, reached0, disabled0, d_ok, gt_v, tf_gt, tf_ov;
  int ty;
  uint8_t kind, h_v;
  float px, py, pyaw, spd, bd0, bd1, bd2, bd3, thresh, d_px, d_py, d_dx, d_dy, g_px, g_py, g_yaw, g_spd, g_acc, g_yr;
  float o_px, o_py, o_yaw, o_spd, o_acc, o_yr, h_p0, h_p1, h_p2, h_m0, h_m1, h_m2, lim_acc, lim_yr, pl0, pl1;
};

__device__ __forceinline__ SimLoads sim_loads(const tbx_sim_state_t& s, const int parts, const int t, const int i, const int sub) {
  SimLoads L;
  const int W = s.window;
  const int b = i / s.n_ag;
  L.valid0 = s.ag_valid[i] != 0;
  L.ty = s.ag_type_idx[i];
  L.px = s.ag_pose[i * 3], L.py = s.ag_pose[i * 3 + 1], L.pyaw = s.ag_pose[i * 3 + 2];
  L.spd = s.ag_motion[i * 3];
  L.player = s.player_valid != nullptr && s.player_valid[i] != 0;
  const float* bd = s.boundary + b * 4;
  L.bd0 = bd[0], L.bd1 = bd[1], L.bd2 = bd[2], L.bd3 = bd[3];
  L.outside0 = s.outside_map[i] != 0;
  L.thresh = s.dest_thresh[i];
  L.kind = s.dest_kind[i];
  L.reached0 = s.dest_reached[i] != 0;
  L.disabled0 = s.ag_disabled[i] != 0;
  // the destination polyline's first 32 nodes (n_node = 20 by default: all of them)
  L.d_ok = false;
  L.d_px = L.d_py = L.d_dx = L.d_dy = 0.f;
  if (sub < s.n_node) {
    const int64_t d = (int64_t)i * s.n_node + sub;
    L.d_ok = s.dest_invalid[d] == 0;
    L.d_px = s.dest_pos[d * 2], L.d_py = s.dest_pos[d * 2 + 1];
    L.d_dx = s.dest_dir[d * 2], L.d_dy = s.dest_dir[d * 2 + 1];
  }
  const bool has_gt = t < s.n_step_gt;
  const int64_t g = (int64_t)i * s.n_step_gt + t;
  L.gt_v = L.tf_gt = false;
  L.g_px = L.g_py = L.g_yaw = L.g_spd = L.g_acc = L.g_yr = 0.f;
  if (has_gt) {
    L.gt_v = s.gt_valid[g] != 0;
    if (s.ov_valid == nullptr) L.tf_gt = s.tf_mask[g] != 0;
    L.g_px = s.gt_pose[g * 3], L.g_py = s.gt_pose[g * 3 + 1], L.g_yaw = s.gt_pose[g * 3 + 2];
    L.g_spd = s.gt_motion[g * 3], L.g_acc = s.gt_motion[g * 3 + 1], L.g_yr = s.gt_motion[g * 3 + 2];
  }
  L.tf_ov = false;
  L.o_px = L.o_py = L.o_yaw = L.o_spd = L.o_acc = L.o_yr = 0.f;
  if (s.ov_valid != nullptr) {  // the step's ag_override handed over explicitly (WaymoMotion.forward)
    L.tf_ov = s.ov_valid[i] != 0;
    L.o_px = s.ov_pose[i * 3], L.o_py = s.ov_pose[i * 3 + 1], L.o_yaw = s.ov_pose[i * 3 + 2];
    L.o_spd = s.ov_motion[i * 3], L.o_acc = s.ov_motion[i * 3 + 1], L.o_yr = s.ov_motion[i * 3 + 2];
  }
  // TrafficBots._append_hist (traffic_bots.py:123-143): slide the window, append the state the next step will see. Lane w
  // moves entry w + 1 to w (first chunk of 32 loaded here; W - 1 <= 32 by default: all of it)
  const uint8_t* hv = s.hist_valid + (int64_t)i * W;
  const float* hp = s.hist_pose + (int64_t)i * W * 3;
  const float* hm = s.hist_motion + (int64_t)i * W * 3;
  const bool mv0 = (parts & TBX_SIM_NO_APPEND) == 0 && sub < W - 1;
  L.h_v = 0;
  L.h_p0 = L.h_p1 = L.h_p2 = L.h_m0 = L.h_m1 = L.h_m2 = 0.f;
  if (mv0) {
    L.h_v = hv[sub + 1];
    L.h_p0 = hp[(sub + 1) * 3], L.h_p1 = hp[(sub + 1) * 3 + 1], L.h_p2 = hp[(sub + 1) * 3 + 2];
    L.h_m0 = hm[(sub + 1) * 3], L.h_m1 = hm[(sub + 1) * 3 + 1], L.h_m2 = hm[(sub + 1) * 3 + 2];
  }
  L.pl0 = L.pl1 = 0.f;
  if (L.player) L.pl0 = s.player_action[i * 2], L.pl1 = s.player_action[i * 2 + 1];  // player-controlled agent (dynamics.py:104-107)
  L.lim_acc = s.max_acc[L.ty], L.lim_yr = s.max_yaw_rate[L.ty];
  return L;
}

// One closed-loop step of agent i by 32 lanes (`sub` = 0..31; half_shift = 0 / 32: which half of the wavefront's ballot is theirs),
// t = *s.step read by the caller, on loads `Ld` (sim_loads) and the action (am0, am1).
__device__ __forceinline__ void sim_agent_on(const tbx_sim_state_t& s, const int parts, const int t, const int i, const int sub, const int half_shift,
                                             const SimLoads& Ld, const float am0, const float am1) {
  const int T = s.n_step_out;
  const int W = s.window;
  const bool valid0 = Ld.valid0, player = Ld.player, outside0 = Ld.outside0, reached0 = Ld.reached0, disabled0 = Ld.disabled0, d_ok = Ld.d_ok;
  const bool gt_v = Ld.gt_v, tf_gt = Ld.tf_gt, tf_ov = Ld.tf_ov;
  const uint8_t kind = Ld.kind, h_v = Ld.h_v;
  const float px = Ld.px, py = Ld.py, pyaw = Ld.pyaw, spd = Ld.spd, bd0 = Ld.bd0, bd1 = Ld.bd1, bd2 = Ld.bd2, bd3 = Ld.bd3, thresh = Ld.thresh;
  const float d_px = Ld.d_px, d_py = Ld.d_py, d_dx = Ld.d_dx, d_dy = Ld.d_dy;
  const float g_px = Ld.g_px, g_py = Ld.g_py, g_yaw = Ld.g_yaw, g_spd = Ld.g_spd, g_acc = Ld.g_acc, g_yr = Ld.g_yr;
  const float o_px = Ld.o_px, o_py = Ld.o_py, o_yaw = Ld.o_yaw, o_spd = Ld.o_spd, o_acc = Ld.o_acc, o_yr = Ld.o_yr;
  const float h_p0 = Ld.h_p0, h_p1 = Ld.h_p1, h_p2 = Ld.h_p2, h_m0 = Ld.h_m0, h_m1 = Ld.h_m1, h_m2 = Ld.h_m2;
  const float lim_acc = Ld.lim_acc, lim_yr = Ld.lim_yr;
  const bool has_gt = t < s.n_step_gt;
  uint8_t* hv = s.hist_valid + (int64_t)i * W;
  float* hp = s.hist_pose + (int64_t)i * W * 3;
  float* hm = s.hist_motion + (int64_t)i * W * 3;
  const bool append = (parts & TBX_SIM_NO_APPEND) == 0;
  const bool mv0 = append && sub < W - 1;
  // ---------------------------------------------------------------- Dynamics.update_ag + MultiPathPP (dynamics.py:84-120,237-274)
  float acc = 0.f, yr = 0.f;
  if (valid0) {
    acc = tanhf(am0) * lim_acc;
    yr = tanhf(am1) * lim_yr;
    if (player) acc = Ld.pl0, yr = Ld.pl1;  // player-controlled agent (dynamics.py:104-107)
  }
  const float half_dt = 0.5f * s.dt;
  const float v_t = spd + half_dt * acc;
  const float th_t = pyaw + half_dt * yr;
  float nx = px + s.dt * (v_t * cosf(th_t));
  float ny = py + s.dt * (v_t * sinf(th_t));
  float nyaw = pyaw + s.dt * yr;
  float nspd = spd + s.dt * acc, nacc = acc, nyr = yr;
  if (!valid0) nx = ny = nyaw = nspd = nacc = nyr = 0.f;
  const float qx = nx, qy = ny, qyaw = nyaw, qspd = nspd, qacc = nacc, qyr = nyr;  // the prediction (before the override): the log's, the reward's
  // outside-map / destination-reached on the predicted (pre-override) state (traffic_rule_checker.py:109-120,300-330)
  const bool out_now = valid0 && (nx > bd1 || nx < bd0 || ny > bd3 || ny < bd2);
  const bool outside = outside0 || out_now;
  bool pos_ok = false, rot_ok = false;
  const float hx = cosf(nyaw), hy = sinf(nyaw);
  if (sub < s.n_node) {
    const float ex = nx - d_px, ey = ny - d_py;
    pos_ok = d_ok && sqrtf(ex * ex + ey * ey) < thresh;
    rot_ok = d_ok && hx * d_dx + hy * d_dy > 0.8660254037844387f;
  }
  for (int k = sub + LPA; k < s.n_node; k += LPA) {  // (polylines of more than 32 nodes)
    const int64_t d = (int64_t)i * s.n_node + k;
    const bool ok = s.dest_invalid[d] == 0;
    const float ex = nx - s.dest_pos[d * 2], ey = ny - s.dest_pos[d * 2 + 1];
    pos_ok = pos_ok || (ok && sqrtf(ex * ex + ey * ey) < thresh);
    rot_ok = rot_ok || (ok && hx * s.dest_dir[d * 2] + hy * s.dest_dir[d * 2 + 1] > 0.8660254037844387f);
  }
  {  // any() over the agent's 32 lanes (its half of the wavefront's ballot)
    const int sh = half_shift;
    pos_ok = ((__ballot(pos_ok) >> sh) & 0xffffffffull) != 0ull;
    rot_ok = ((__ballot(rot_ok) >> sh) & 0xffffffffull) != 0ull;
  }
  const bool reach_now = !reached0 && valid0 && (((kind & 1) && pos_ok && rot_ok) || ((kind & 2) && pos_ok));
  const bool reached = reached0 || reach_now;
  // TeacherForcing.get + Dynamics.override_ag (teacher_forcing.py:128-147, dynamics.py:122-141)
  bool valid = valid0;
  bool disabled = disabled0;
  const bool tf_now = s.ov_valid != nullptr ? tf_ov : tf_gt;
  if (tf_now && !disabled) {
    valid = true;
    if (s.ov_valid != nullptr) {
      nx = o_px, ny = o_py, nyaw = o_yaw, nspd = o_spd, nacc = o_acc, nyr = o_yr;
    } else {
      nx = g_px, ny = g_py, nyaw = g_yaw, nspd = g_spd, nacc = g_acc, nyr = g_yr;
    }
  }
  // Dynamics.disable_ag / disable_navi (dynamics.py:165-204); a step-wise caller does both itself from now_*
  const bool no_disable = (parts & TBX_SIM_NO_DISABLE) != 0;
  const bool dis = !no_disable && out_now && !(has_gt && gt_v);
  disabled = disabled || dis;
  valid = valid && !dis;
  // ---------------------------------------------------------------- stores
  // the window: the wavefront runs in lockstep, so every lane has loaded before any lane stores (chunks of 32 go upwards, each reads
  // only entries no earlier chunk wrote)
  if (mv0) {
    hv[sub] = h_v;
    hp[sub * 3] = h_p0, hp[sub * 3 + 1] = h_p1, hp[sub * 3 + 2] = h_p2;
    hm[sub * 3] = h_m0, hm[sub * 3 + 1] = h_m1, hm[sub * 3 + 2] = h_m2;
  }
  __builtin_amdgcn_wave_barrier();
  for (int w0 = LPA; append && w0 < W - 1; w0 += LPA) {  // (windows of more than 33 steps)
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
  if (sub != 0) return;
  if (t - 1 < T) {
    const int64_t o = (int64_t)i * T + (t - 1);
    s.out_valid[o] = valid0 ? 1 : 0;
    s.out_pose[o * 3] = qx;
    s.out_pose[o * 3 + 1] = qy;
    s.out_pose[o * 3 + 2] = qyaw;
    s.out_motion[o * 3] = qspd;
    s.out_motion[o * 3 + 1] = qacc;
    s.out_motion[o * 3 + 2] = qyr;
    s.out_action[o * 2] = acc;
    s.out_action[o * 2 + 1] = yr;
    s.out_outside_map[o] = outside ? 1 : 0;
    s.out_dest_reached[o] = reached ? 1 : 0;
    if (s.out_tf != nullptr) s.out_tf[o] = tf_now ? 1 : 0;
    // DifferentiableReward.get on the prediction (rewards.py:58-74; the same expressions as tbx_train_chain_fwd)
    if (s.out_reward != nullptr) {
      float r_pos = 0.f, r_rot = 0.f, r_spd = 0.f;
      bool r_valid = valid0;
      if (has_gt) {
        r_valid = valid0 && gt_v;
        if (r_valid) {
          r_pos = -s.w_pos * (sim_sl1(g_px - qx) + sim_sl1(g_py - qy));
          r_rot = -s.w_rot * (0.5f * (1.f - cosf(g_yaw - qyaw)));
          r_spd = -s.w_spd * sim_sl1(g_spd - qspd);
        }
      }
      s.out_reward[o * 4] = r_pos, s.out_reward[o * 4 + 1] = r_rot, s.out_reward[o * 4 + 2] = r_spd;
      s.out_reward[o * 4 + 3] = (r_pos + r_rot) + r_spd;
      if (s.out_reward_valid != nullptr) s.out_reward_valid[o] = r_valid ? 1 : 0;
    }
  }
  if (s.now_outside != nullptr) s.now_outside[i] = out_now ? 1 : 0;
  if (s.now_reached != nullptr) s.now_reached[i] = reach_now ? 1 : 0;
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

// ... with its own loads and the action from memory (s.action_mean): the stand-alone kernel's form
__device__ __forceinline__ void sim_agent(const tbx_sim_state_t& s, const int parts, const int t, const int i, const int sub, const int half_shift) {
  const float am0 = s.action_mean[i * 2], am1 = s.action_mean[i * 2 + 1];
  const SimLoads Ld = sim_loads(s, parts, t, i, sub);
  sim_agent_on(s, parts, t, i, sub, half_shift, Ld, am0, am1);
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

// tbx_tl_prep of the lights' new windows riding on their update (tbx_sim_step_tl_prep): a light's lanes write its own W rows
struct TlPrepArgs {
  const uint8_t* tl_invalid;  // NULL: off
  float* attr;
  uint8_t* row_invalid;
  int ld_attr;
};

constexpr int LPT = 8;  // lanes per traffic light

// One closed-loop step of light i by LPT lanes (lsub = 0..7 of one wavefront): Dynamics.override_tl (dynamics.py:143-163: argmax of the
// predictor's logits -> one-hot, ground truth while it lasts), the log (state, NLL: waymo_motion.py:276-283), the window shift
// (traffic_bots.py:123-143) and, with tp.tl_invalid, the tbx_tl_prep rows of the new window. Every lane repeats the light's few scalar
// operations (broadcast loads), the lanes split the window's W entries. Shared by sim_step_kernel (csrc/sim.hip) and the lights' tail
// of the one-launch decoder layer (csrc/dec_layer_mf.inc: the row's own light, right behind its logits).
__device__ __forceinline__ void sim_light(const tbx_sim_state_t& s, const int parts, const int t, const int i, const int lsub, const TlPrepArgs& tp) {
  const int T = s.n_step_out;
  const int W = s.window;
  const float* lg = s.tl_logits + (int64_t)i * 5;
  int am = 0;
  float best = lg[0];
  for (int c = 1; c < 5; ++c)
    if (lg[c] > best) {
      best = lg[c];
      am = c;
    }
  uint8_t st = (uint8_t)(1u << am);
  if (s.ov_valid != nullptr) {
    if (s.ov_tl_valid[i] != 0) st = s.ov_tl_state[i];
  } else if (t < s.n_step_tl_gt) {
    st = s.tl_gt[(int64_t)i * s.n_step_tl_gt + t];
  }
  if (lsub == 0) {
    s.tl_state[i] = st;
    if (t - 1 < T) s.out_tl_state[(int64_t)i * T + (t - 1)] = st;
    if (s.out_tl_nll != nullptr && t - 1 < T) {
      // -Categorical(logits).log_prob(gt) = logsumexp(logits) - logits[gt] (waymo_motion.py:276-283); 0 past the ground truth
      float nll = 0.f;
      if (t < s.n_step_tl_gt) {
        const uint8_t gm = s.tl_gt[(int64_t)i * s.n_step_tl_gt + t];
        const int gi = gm ? (__ffs((int)gm) - 1) : 0;
        float se = 0.f;
        for (int c = 0; c < 5; ++c) se += expf(lg[c] - best);
        nll = (best + logf(se)) - lg[gi < 5 ? gi : 0];
      }
      s.out_tl_nll[(int64_t)i * T + (t - 1)] = nll;
    }
  }
  // the window: entry w of the new window = old entry w + 1, the last one = the new state. Every lane loads its entries before
  // any lane stores (the wavefront runs in lockstep; chunks of LPT go upwards, each reads only entries no earlier chunk wrote)
  uint8_t* ht = s.hist_tl + (int64_t)i * W;
  const bool append = (parts & TBX_SIM_NO_APPEND) == 0;
  const bool tok_bad = tp.tl_invalid != nullptr && tp.tl_invalid[i] != 0;
  for (int w0 = 0; w0 < W; w0 += LPT) {
    const int w = w0 + lsub;
    uint8_t hs = 0;
    if (w < W) hs = append ? (w < W - 1 ? ht[w + 1] : st) : ht[w];
    __builtin_amdgcn_wave_barrier();
    if (w < W && append) ht[w] = hs;
    __builtin_amdgcn_wave_barrier();
    if (w < W && tp.tl_invalid != nullptr) {  // csrc/prep.hip tl_prep_kernel for row (i, w) of the window just written
      const bool missing = hs == 0xFF;
      const int64_t r = (int64_t)i * W + w;
      for (int c0 = 0; c0 < tp.ld_attr; c0 += 4) {
        float4 v;
        float* vv = &v.x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + q;
          vv[q] = c < 5 ? ((!missing && ((hs >> c) & 1)) ? 1.f : 0.f) : (c - 5 == w ? 1.f : 0.f);
        }
        *(float4*)(tp.attr + r * tp.ld_attr + c0) = v;
      }
      tp.row_invalid[r] = (missing || tok_bad) ? 1 : 0;
    }
  }
}

typedef tbx_agent_prep_args_t AgentPrepArgs;  // (field order = tbx_agent_prep's parameter groups)

// (contraction off: HIP's __fmul_rn / __fadd_rn are plain operators, and which of the two products the compiler fuses into the sum
// is otherwise its choice per call site - the standalone kernel and the decoder layer's tail would round differently)
__device__ __forceinline__ void to_local(float x0, float y0, float c, float s, float x, float y, float& rx, float& ry) {
#pragma clang fp contract(off)
  const float dx = __fsub_rn(x, x0), dy = __fsub_rn(y, y0);
  rx = __fadd_rn(__fmul_rn(dx, c), __fmul_rn(dy, s));
  ry = __fadd_rn(__fmul_rn(dx, -s), __fmul_rn(dy, c));
}

// tbx_agent_prep for agent i by `nthreads` threads (a multiple of 64; tid = 0..nthreads-1). A "slot" = 32 lanes (the pose embedding
// of one window step is 32 sincosf arguments at the default pe_dim = 64): the window's steps are dealt to the slots - with the 512
// threads of the decoder layer's workgroup every step of the default window of 11 has its own - and the destination's relative pose
// goes to the last slot. Every wavefront loads the whole window once (validity bytes, 3 W + 3 W floats: one round trip to memory);
// the last valid step comes from one ballot, its pose and each slot's own step from lane shuffles of those registers.
__device__ __forceinline__ void agent_prep(const AgentPrepArgs& a, const int i, const int tid, const int nthreads) {
  const int lane = tid & 63;
  const int l = tid & 31;
  const int slot = tid >> 5, n_slot = nthreads >> 5;
  if (i >= a.n_tok) return;
  const int W = a.window;  // <= 23 (attribute row: 9 + W <= 32)
  const uint8_t* hv = a.hist_valid + (int64_t)i * W;
  const float* hp = a.hist_pose + (int64_t)i * W * 3;
  const float* hm = a.hist_motion + (int64_t)i * W * 3;
  // ---------------------------------------------------------------- loads
  const bool hv_l = lane < W && hv[lane < W ? lane : 0] != 0;
  const float hp0 = lane < 3 * W ? hp[lane] : 0.f, hp1 = lane + 64 < 3 * W ? hp[lane + 64] : 0.f;
  const float hm0 = lane < 3 * W ? hm[lane] : 0.f, hm1 = lane + 64 < 3 * W ? hm[lane + 64] : 0.f;
  const float attr6 = l < 6 ? a.ag_attr6[(int64_t)i * 6 + l] : 0.f;
  const bool dest_slot = slot == n_slot - 1;
  float dpx = 0.f, dpy = 0.f, dpyaw = 0.f;
  int64_t mrow = 0;
  int ty = 0;
  if (dest_slot) {
    if (a.type_mask != nullptr) ty = a.ag_type_idx[i];
    if (a.dest != nullptr) {
      const int b = i / a.n_ag;
      mrow = (int64_t)(b / a.mp_batch_div) * a.n_mp + a.dest[i];
      dpx = a.mp_tok_pose[mrow * 3], dpy = a.mp_tok_pose[mrow * 3 + 1], dpyaw = a.mp_tok_pose[mrow * 3 + 2];
    }
  }
  auto at = [&](const float f0, const float f1, const int idx) { return idx < 64 ? __shfl(f0, idx) : __shfl(f1, idx - 64); };
  // ----------------------------------------------------------------
  const unsigned long long vmask = __ballot(hv_l);
  const int last = vmask ? 63 - __builtin_clzll(vmask) : -1;
  float x0 = 0.f, y0 = 0.f, yaw0 = 0.f;
  {
    const int q = last >= 0 ? last * 3 : 0;
    const float fx = at(hp0, hp1, q), fy = at(hp0, hp1, q + 1), fw = at(hp0, hp1, q + 2);
    if (last >= 0) x0 = fx, y0 = fy, yaw0 = fw;
  }
  if (tid == 0) {
    a.tok_pose[i * 3] = x0;
    a.tok_pose[i * 3 + 1] = y0;
    a.tok_pose[i * 3 + 2] = yaw0;
    a.tok_invalid[i] = last < 0 ? 1 : 0;
  }
  const float c = cosf(yaw0), s = sinf(yaw0);
  const int n_it = (W + n_slot - 1) / n_slot;  // (the shuffles below want every lane of the wavefront: no slot leaves the loop early)
  for (int it = 0; it < n_it; ++it) {
    const int w = it * n_slot + slot;
    const int wq = w < W ? w : 0;
    const float qx = at(hp0, hp1, wq * 3), qy = at(hp0, hp1, wq * 3 + 1), qyaw = at(hp0, hp1, wq * 3 + 2);
    const int mi = wq * 3 + (l >= 6 && l < 9 ? l - 6 : 0);
    const float mot = at(hm0, hm1, mi);
    if (w >= W) continue;
    const int64_t r = (int64_t)i * W + w;
    float rx, ry;
    to_local(x0, y0, c, s, qx, qy, rx, ry);
    const float ryaw = __fsub_rn(qyaw, yaw0);
    tbx::pose_emb_write(a.pe + r * a.pe_dim, a.pe_dim, rx, ry, ryaw, a.freqs_xy, a.freqs_yaw, l, 32);
    float v = 0.f;
    if (l < 6)
      v = attr6;
    else if (l < 9)
      v = mot;
    else if (l - 9 == w)
      v = 1.f;
    a.attr[r * 32 + l] = v;
    if (l == 0) a.row_invalid[r] = ((vmask >> w) & 1ull) ? 0 : 1;
  }
  const float ax = at(hp0, hp1, (W - 1) * 3), ay = at(hp0, hp1, (W - 1) * 3 + 1), ayaw = at(hp0, hp1, (W - 1) * 3 + 2);
  if (!dest_slot) return;
  if (a.type_mask != nullptr && l < 3) {
    const bool now = ((vmask >> (W - 1)) & 1ull) != 0;
    a.type_mask[(int64_t)l * a.n_tok + i] = (now && ty == l) ? 0 : 1;
  }
  if (a.dest != nullptr && l == 0) {
    const float cc = cosf(ayaw), ss = sinf(ayaw);
    float rx, ry;
    to_local(ax, ay, cc, ss, dpx, dpy, rx, ry);
    a.navi_pose3[i * 3] = rx;
    a.navi_pose3[i * 3 + 1] = ry;
    a.navi_pose3[i * 3 + 2] = __fsub_rn(dpyaw, ayaw);
    a.navi_row[i] = (int32_t)mrow;
  }
}

}  // namespace tbx_step
