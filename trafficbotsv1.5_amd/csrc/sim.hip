// tbx_sim_step: one closed-loop simulation step for every agent and traffic light of every scene
// (see include/tbx_hip.h; restates dynamics.py / teacher_forcing.py / the feeding-back subset of
// traffic_rule_checker.py as the oracle's Sim.rollout does). Elementwise, one thread per agent / light; the 1-based
// step index lives in device memory and is bumped by a second single-thread kernel so that one captured graph replays
// every step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

// Agents: 32 lanes (half a wavefront) per agent - every lane repeats the agent's scalar dynamics (broadcast loads), the
// lanes split the destination polyline's nodes and the window shift, so the kernel is 2-3 dependent loads deep instead
// of ~30 (it closes the critical path of every step: 23 -> ~7 us). Lights: one thread per light.
constexpr int LPA = 32;

__device__ __forceinline__ float sim_sl1(float d) {  // F.smooth_l1_loss, beta = 1
  const float a = fabsf(d);
  return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

// tbx_tl_prep of the lights' new windows riding on their update (tbx_sim_step_tl_prep): a light's thread writes its own W rows
struct TlPrepArgs {
  const uint8_t* tl_invalid;  // NULL: off
  float* attr;
  uint8_t* row_invalid;
  int ld_attr;
};

__global__ void sim_step_kernel(const tbx_sim_state_t s, const int parts, const TlPrepArgs tp) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i_ag = gid / LPA, sub = gid % LPA;
  const int t = *s.step;  // step being simulated: model saw the state of step t-1
  const int n_ag_tot = s.n_batch * s.n_ag;
  const int n_tl_tot = s.n_batch * s.n_tl;
  const int T = s.n_step_out;
  const int W = s.window;
  if ((parts & TBX_SIM_AGENTS) && i_ag < n_ag_tot) {
    const int i = i_ag;
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
      const int sh = (threadIdx.x & 32);
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
  if ((parts & TBX_SIM_LIGHTS) && gid < n_tl_tot) {
    const int i = gid;
    // Dynamics.override_tl (dynamics.py:143-163): argmax -> one-hot, ground truth while it lasts
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
    if ((parts & TBX_SIM_NO_APPEND) == 0) {
      uint8_t* ht = s.hist_tl + (int64_t)i * W;
      for (int w = 0; w < W - 1; ++w) ht[w] = ht[w + 1];
      ht[W - 1] = st;
    }
    if (tp.tl_invalid != nullptr) {  // csrc/prep.hip tl_prep_kernel for rows (i, 0 .. W-1) of the window just written
      const uint8_t* ht = s.hist_tl + (int64_t)i * W;
      const bool tok_bad = tp.tl_invalid[i] != 0;
      for (int w = 0; w < W; ++w) {
        const uint8_t hs = ht[w];
        const bool missing = hs == 0xFF;
        const int64_t r = (int64_t)i * W + w;
        for (int c = 0; c < tp.ld_attr; ++c) {
          float v = 0.f;
          if (c < 5)
            v = (!missing && ((hs >> c) & 1)) ? 1.f : 0.f;
          else if (c - 5 == w)
            v = 1.f;
          tp.attr[r * tp.ld_attr + c] = v;
        }
        tp.row_invalid[r] = (missing || tok_bad) ? 1 : 0;
      }
    }
  }
  if (parts & TBX_SIM_ADVANCE) {
    // every thread of this workgroup has read *step above; the last workgroup to arrive advances it
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned prev = atomicAdd((unsigned*)(s.step + 1), 1u);
      if (prev == gridDim.x - 1) {
        s.step[1] = 0;
        __threadfence();
        s.step[0] = t + 1;
      }
    }
  }
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
  const int64_t th_tl = (parts & TBX_SIM_LIGHTS) ? (int64_t)s.n_batch * s.n_tl : 0;
  const int64_t n = th_ag > th_tl ? th_ag : th_tl;
  hipStream_t hs = (hipStream_t)stream;
  if (parts & (TBX_SIM_AGENTS | TBX_SIM_LIGHTS))
    hipLaunchKernelGGL(sim_step_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, hs, s, parts, tp);
  else
    hipLaunchKernelGGL(sim_bump_kernel, dim3(1), dim3(1), 0, hs, s.step);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
