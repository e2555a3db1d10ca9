// Training rollout's per-step state machine as one kernel pair (tbx_train_chain_fwd / _bwd, include/tbx_hip.h): action
// scaling, the kinematic step, teacher-forcing override, the feeding-back rule flags (outside the map, destination reached), the
// differentiable reward and the validity bookkeeping of waymo_motion.py:206-311 (training=True) - utils/dynamics.py:66-204,
// 237-274, utils/teacher_forcing.py:108-167, utils/traffic_rule_checker.py:109-120,300-330, utils/rewards.py:35-85 - for steps
// [t0, t1) of every agent. Agents never interact in it, so a thread owns an agent and walks the steps; the state lives in
// device memory between calls: the no-grad stepping pass calls it one step at a time (a policy evaluation sits between two
// steps), the differentiated pass once over all steps. The only gradient path across steps is pose / speed through the
// kinematic step; the backward walks the steps in reverse with a 4-float adjoint per agent (analytic, no tape).
// Replaces ~75 elementwise launches per step forward and ~230 backward (28 k launches of a 90-step training step).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

namespace {

__device__ __forceinline__ float sl1(float d) {  // F.smooth_l1_loss, beta = 1
  const float a = fabsf(d);
  return a < 1.f ? 0.5f * d * d : a - 0.5f;
}
__device__ __forceinline__ float sl1_grad(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }

// The policy inputs of the NEXT step in the layout the policy reads them in (tbx_train_chain_fwd_windows): the W-step windows [n, A, W
// (, 3)], oldest first, and the current validity / navigation flags - written by the thread that owns the agent, behind its step (the
// stepping pass spent six strided-copy launches per closed-loop step on these).
struct WinOut {
  uint8_t* hv;   // [n, A, W]
  float* hp;     // [n, A, W, 3]
  float* hm;     // [n, A, W, 3]
  uint8_t* valid;       // [n, A]
  uint8_t* navi_valid;  // [n, A]
};

__global__ __launch_bounds__(64) void train_chain_fwd_kernel(const tbx_train_chain_t c, const float* __restrict__ mean, int64_t sn,
                                                             int64_t st, int t0, int t1, const WinOut win) {
#pragma clang fp contract(off)
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (a >= c.n_ag) return;
  const int A = c.n_ag, T = c.n_step, Tg = c.n_step_gt, N = c.n_node;
  const int64_t ia = (int64_t)b * A + a;
  bool valid = c.valid[ia] != 0, disabled = c.disabled[ia] != 0, navi_valid = c.navi_valid[ia] != 0;
  bool outside = c.outside[ia] != 0, reached = c.reached[ia] != 0;
  float px = c.pose[ia * 3], py = c.pose[ia * 3 + 1], pw = c.pose[ia * 3 + 2];
  float mv = c.motion[ia * 3], ma = c.motion[ia * 3 + 1], mw = c.motion[ia * 3 + 2];
  const float lim0 = c.lim[ia * 2], lim1 = c.lim[ia * 2 + 1];
  const float thresh = c.dest_thresh[ia];
  const int kind = c.dest_kind[ia];
  const float bx0 = c.boundary[b * 4], bx1 = c.boundary[b * 4 + 1], by0 = c.boundary[b * 4 + 2], by1 = c.boundary[b * 4 + 3];
  const float dt = c.dt;
  const int rec_T = T + c.window;  // record slots per scene: window - 1 leading (empty) slots, then the state before step s in slot window - 2 + s
  for (int t = t0; t < t1; ++t) {
    const int s = t + 1;  // step number (gt index)
    const int64_t it = ((int64_t)b * T + t) * A + a;
    const float m0 = mean[b * sn + (t - t0) * st + a * 2], m1 = mean[b * sn + (t - t0) * st + a * 2 + 1];
    float acc = tanhf(m0) * lim0, yr = tanhf(m1) * lim1;
    if (!valid) acc = 0.f, yr = 0.f;
    const float v_t = mv + 0.5f * dt * acc, th_t = pw + 0.5f * dt * yr;
    float nx = px + dt * (v_t * cosf(th_t)), ny = py + dt * (v_t * sinf(th_t)), nw = pw + dt * yr;
    float nv = mv + dt * acc, na = acc, nyr = yr;
    if (!valid) nx = ny = nw = nv = na = nyr = 0.f;
    const bool pred_valid = valid;
    c.pred_valid[it] = pred_valid;
    c.pred_pose[it * 3] = nx, c.pred_pose[it * 3 + 1] = ny, c.pred_pose[it * 3 + 2] = nw;
    c.pred_motion[it * 3] = nv, c.pred_motion[it * 3 + 1] = na, c.pred_motion[it * 3 + 2] = nyr;
    // teacher forcing (teacher_forcing.py:108-167 through the precomputed mask)
    bool ov = false, tf_log = false, g_valid = false;
    float gx = 0.f, gy = 0.f, gw = 0.f, gv = 0.f;
    const int64_t ig = ((int64_t)b * A + a) * Tg + s;
    if (s < Tg) {
      tf_log = c.tf_mask[ig] != 0;
      ov = tf_log && !disabled;
      g_valid = c.gt_valid[ig] != 0;
      gx = c.gt_pose[ig * 3], gy = c.gt_pose[ig * 3 + 1], gw = c.gt_pose[ig * 3 + 2];
      gv = c.gt_motion[ig * 3];
    }
    c.tf[it] = tf_log;
    c.ov[it] = ov;
    valid = valid || ov;
    if (ov) {
      px = gx, py = gy, pw = gw;
      mv = gv, ma = c.gt_motion[ig * 3 + 1], mw = c.gt_motion[ig * 3 + 2];
    } else {
      px = nx, py = ny, pw = nw;
      mv = nv, ma = na, mw = nyr;
    }
    // feeding-back rule flags on the prediction (traffic_rule_checker.py:109-120,300-330)
    const bool out_now = ((nx > bx1) || (nx < bx0) || (ny > by1) || (ny < by0)) && pred_valid;
    outside = outside || out_now;
    bool pos_ok = false, rot_ok = false;
    const float hx = cosf(nw), hy = sinf(nw);
    for (int j = 0; j < N; ++j) {
      const int64_t id = ia * N + j;
      if (c.dest_invalid[id]) continue;
      const float dx = nx - c.dest_pos[id * 2], dy = ny - c.dest_pos[id * 2 + 1];
      pos_ok = pos_ok || (sqrtf(dx * dx + dy * dy) < thresh);
      rot_ok = rot_ok || (hx * c.dest_dir[id * 2] + hy * c.dest_dir[id * 2 + 1] > 0.8660254037844387f);
    }
    const bool reach_now = !reached && pred_valid && (((kind & 1) && pos_ok && rot_ok) || ((kind & 2) && pos_ok));
    reached = reached || reach_now;
    // differentiable reward (rewards.py:58-74), disabling (waymo_motion.py:262-275)
    bool r_valid, dis;
    float rew = 0.f;
    if (s < Tg) {
      r_valid = pred_valid && g_valid;
      if (r_valid) {
        const float e_pos = sl1(gx - nx) + sl1(gy - ny);
        const float e_rot = 0.5f * (1.f - cosf(gw - nw));
        const float e_spd = sl1(gv - nv);
        rew = (-c.w_pos * e_pos) + (-c.w_rot * e_rot) + (-c.w_spd * e_spd);
      }
      dis = out_now && !g_valid;
    } else {
      r_valid = pred_valid;
      dis = out_now;
    }
    c.reward[it] = rew;
    c.reward_valid[it] = r_valid;
    disabled = disabled || dis;
    valid = valid && !dis;
    navi_valid = navi_valid && !reach_now;
    // the state before step s + 1
    const int64_t ir = ((int64_t)b * rec_T + (c.window - 1 + s)) * A + a;
    c.rec_valid[ir] = valid;
    c.rec_navi_valid[ir] = navi_valid;
    c.rec_pose[ir * 3] = px, c.rec_pose[ir * 3 + 1] = py, c.rec_pose[ir * 3 + 2] = pw;
    c.rec_motion[ir * 3] = mv, c.rec_motion[ir * 3 + 1] = ma, c.rec_motion[ir * 3 + 2] = mw;
  }
  c.valid[ia] = valid, c.disabled[ia] = disabled, c.navi_valid[ia] = navi_valid, c.outside[ia] = outside, c.reached[ia] = reached;
  c.pose[ia * 3] = px, c.pose[ia * 3 + 1] = py, c.pose[ia * 3 + 2] = pw;
  c.motion[ia * 3] = mv, c.motion[ia * 3 + 1] = ma, c.motion[ia * 3 + 2] = mw;
  if (win.hv != nullptr) {
    // the window of step t1 + 1 = record slots t1 .. t1 + W - 1 (slot W - 1 + s holds the state before step s + 1; this thread wrote
    // all of its agent's slots: the last one just above, the others in earlier launches)
    // (all of a chunk's loads first, then its stores: a load -> store -> load chain per slot cost 16 us per launch)
    const int W = c.window;
    for (int w0 = 0; w0 < W; w0 += 4) {
      uint8_t v[4];
      float p[4][3], m[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int w = w0 + q < W ? w0 + q : W - 1;
        const int64_t ir = ((int64_t)b * rec_T + (t1 + w)) * A + a;
        v[q] = c.rec_valid[ir];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[q][k] = c.rec_pose[ir * 3 + k], m[q][k] = c.rec_motion[ir * 3 + k];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (w0 + q >= W) break;
        const int64_t iw = ia * W + w0 + q;
        win.hv[iw] = v[q];
#pragma unroll
        for (int k = 0; k < 3; ++k) win.hp[iw * 3 + k] = p[q][k], win.hm[iw * 3 + k] = m[q][k];
      }
    }
    const int64_t ic = ((int64_t)b * rec_T + (t1 + W - 1)) * A + a;
    win.valid[ia] = c.rec_valid[ic];
    win.navi_valid[ia] = c.rec_navi_valid[ic];
  }
}

// d(mean) from d(reward), all steps, in reverse. Reads what the forward over [0, T) left behind: the records of the state
// before every step (slot window - 1 + t holds the state before step t + 1), the override flags and the predictions.
__global__ __launch_bounds__(64) void train_chain_bwd_kernel(const tbx_train_chain_t c, const float* __restrict__ mean, int64_t sn,
                                                             int64_t st, const float* __restrict__ d_reward,
                                                             float* __restrict__ d_mean) {
#pragma clang fp contract(off)
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (a >= c.n_ag) return;
  const int A = c.n_ag, T = c.n_step, Tg = c.n_step_gt;
  const int64_t ia = (int64_t)b * A + a;
  const float lim0 = c.lim[ia * 2], lim1 = c.lim[ia * 2 + 1], dt = c.dt;
  const int rec_T = T + c.window;
  float ax = 0.f, ay = 0.f, aw = 0.f, av = 0.f;  // adjoint of the state after step s (x, y, yaw, speed)
  for (int t = T - 1; t >= 0; --t) {
    const int s = t + 1;
    const int64_t it = ((int64_t)b * T + t) * A + a;
    const int64_t ir = ((int64_t)b * rec_T + (c.window - 1 + t)) * A + a;
    const bool valid = c.rec_valid[ir] != 0;
    float dm0 = 0.f, dm1 = 0.f;
    if (!valid) {  // the prediction is the constant 0: nothing flows to the state before or to the action
      ax = ay = aw = av = 0.f;
    } else {
      float dx = 0.f, dy = 0.f, dw = 0.f, dv = 0.f;  // adjoint of the prediction (x', y', yaw', speed')
      if (!c.ov[it]) dx = ax, dy = ay, dw = aw, dv = av;
      if (s < Tg && c.reward_valid[it]) {
        const int64_t ig = ((int64_t)b * A + a) * Tg + s;
        const float g = d_reward[it];
        const float nx = c.pred_pose[it * 3], ny = c.pred_pose[it * 3 + 1], nw = c.pred_pose[it * 3 + 2], nv = c.pred_motion[it * 3];
        dx += g * c.w_pos * sl1_grad(c.gt_pose[ig * 3] - nx);
        dy += g * c.w_pos * sl1_grad(c.gt_pose[ig * 3 + 1] - ny);
        dw += g * 0.5f * c.w_rot * sinf(c.gt_pose[ig * 3 + 2] - nw);
        dv += g * c.w_spd * sl1_grad(c.gt_motion[ig * 3] - nv);
      }
      const float pw = c.rec_pose[ir * 3 + 2], mv = c.rec_motion[ir * 3];
      const float u0 = tanhf(mean[b * sn + t * st + a * 2]), u1 = tanhf(mean[b * sn + t * st + a * 2 + 1]);
      const float acc = u0 * lim0, yr = u1 * lim1;
      const float v_t = mv + 0.5f * dt * acc, th_t = pw + 0.5f * dt * yr;
      const float cs = cosf(th_t), sn_ = sinf(th_t);
      const float d_vt = dt * (dx * cs + dy * sn_);
      const float d_th = dt * v_t * (dy * cs - dx * sn_);
      const float d_yr = dt * dw + 0.5f * dt * d_th;
      const float d_acc = dt * dv + 0.5f * dt * d_vt;
      ax = dx, ay = dy, aw = dw + d_th, av = dv + d_vt;
      dm0 = d_acc * lim0 * (1.f - u0 * u0);
      dm1 = d_yr * lim1 * (1.f - u1 * u1);
    }
    d_mean[it * 2] = dm0;
    d_mean[it * 2 + 1] = dm1;
  }
}

int check(const tbx_train_chain_t* c) {
  if (!c) return TBX_ERR_ARG;
  if (c->n_batch <= 0 || c->n_ag <= 0 || c->n_step <= 0 || c->n_step_gt <= 0 || c->n_node < 0 || c->window < 1) return TBX_ERR_ARG;
  const void* ptrs[] = {c->gt_valid, c->gt_pose, c->gt_motion, c->tf_mask, c->lim, c->dest_thresh, c->dest_kind, c->boundary, c->valid,
                        c->disabled, c->navi_valid, c->outside, c->reached, c->pose, c->motion, c->rec_valid, c->rec_pose, c->rec_motion,
                        c->rec_navi_valid, c->pred_valid, c->tf, c->ov, c->reward_valid, c->pred_pose, c->pred_motion, c->reward};
  for (const void* p : ptrs)
    if (!p) return TBX_ERR_ARG;
  if (c->n_node > 0 && (!c->dest_pos || !c->dest_dir || !c->dest_invalid)) return TBX_ERR_ARG;
  return TBX_OK;
}

}  // namespace

extern "C" int tbx_train_chain_fwd(const tbx_train_chain_t* c, const float* mean, int64_t mean_stride_n, int64_t mean_stride_t, int t0,
                                   int t1, void* stream) {
  const int rc = check(c);
  if (rc != TBX_OK) return rc;
  if (!mean || t0 < 0 || t1 > c->n_step || t0 >= t1) return TBX_ERR_ARG;
  const WinOut none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(train_chain_fwd_kernel, dim3((c->n_ag + 63) / 64, c->n_batch), dim3(64), 0, (hipStream_t)stream, *c, mean,
                     mean_stride_n, mean_stride_t, t0, t1, none);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_train_chain_fwd_windows(const tbx_train_chain_t* c, const float* mean, int64_t mean_stride_n, int64_t mean_stride_t, int t0,
                                           int t1, uint8_t* hv, float* hp, float* hm, uint8_t* valid, uint8_t* navi_valid, void* stream) {
  const int rc = check(c);
  if (rc != TBX_OK) return rc;
  if (t0 < 0 || t1 > c->n_step || t0 > t1 || (t0 < t1 && !mean) || !hv || !hp || !hm || !valid || !navi_valid) return TBX_ERR_ARG;
  const WinOut win = {hv, hp, hm, valid, navi_valid};
  hipLaunchKernelGGL(train_chain_fwd_kernel, dim3((c->n_ag + 63) / 64, c->n_batch), dim3(64), 0, (hipStream_t)stream, *c, mean,
                     mean_stride_n, mean_stride_t, t0, t1, win);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_train_chain_bwd(const tbx_train_chain_t* c, const float* mean, int64_t mean_stride_n, int64_t mean_stride_t,
                                   const float* d_reward, float* d_mean, void* stream) {
  const int rc = check(c);
  if (rc != TBX_OK) return rc;
  if (!mean || !d_reward || !d_mean) return TBX_ERR_ARG;
  hipLaunchKernelGGL(train_chain_bwd_kernel, dim3((c->n_ag + 63) / 64, c->n_batch), dim3(64), 0, (hipStream_t)stream, *c, mean,
                     mean_stride_n, mean_stride_t, d_reward, d_mean);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
