// tbx_rule_*: the metric-only traffic-rule checks of a closed-loop rollout, batched over every
// (rollout, step) frame of the rollout log instead of once per Python step: collided (separating edge test), collided_wosac
// (signed distance of the Minkowski difference of two rounded boxes), run_road_edge (box edge x road-edge segment crossing),
// run_red_light (stop point leaves the footprint within 0.1 s) and passive (slow vehicle on a lane with nothing ahead).
// Restates utils/traffic_rule_checker.py:122-298,453-505 and utils/wosac_collision.py:20-257 of the reference; the outputs
// are booleans, so every expression keeps the reference's operation order (no FMA contraction; torch's 2-norm is
// sqrt(fma(y, y, x*x)) on the CPU, measured) and the CPU oracle (oracle/rule_checks.py) is matched bit for bit.
//
// One wavefront per (frame, agent): the frame's agents (pose, boxes) are staged once per workgroup in LDS, lanes stride over
// the other agents / the lights / the scene's compacted road-edge and lane-centre tables (static per scene, L2-resident,
// 16-byte coalesced loads), wave ballots give the any() reductions. HBM-light, latency / VALU bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_AG = 256;
constexpr float BIG = 1e10f;

__device__ __forceinline__ float norm2(float x, float y) { return __fsqrt_rn(__fmaf_rn(y, y, __fmul_rn(x, x))); }

// ---------------------------------------------------------------------------------------------- static tables
// traffic_rule_checker.py:453-501: road-edge node segments (types 4, 5, 7) and lane-centre nodes (types 0, 1, 2) of one
// scene, compacted (their order is irrelevant: every consumer is an any()). One workgroup per scene.
__global__ void rule_tables_kernel(const uint8_t* __restrict__ mp_valid, const uint8_t* __restrict__ mp_type_idx,
                                   const float* __restrict__ mp_pos, const float* __restrict__ mp_dir, int ld_xy, int n_mp,
                                   int n_node, float* __restrict__ seg, int32_t* __restrict__ n_seg, float* __restrict__ lane,
                                   int32_t* __restrict__ n_lane) {
  __shared__ int cnt[2];
  const int b = blockIdx.x;
  const int cap = n_mp * n_node;
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < cap; i += blockDim.x) {
    const int64_t g = (int64_t)b * cap + i;
    if (!mp_valid[g]) continue;
    const int ty = mp_type_idx[(int64_t)b * n_mp + i / n_node];
    const float px = mp_pos[g * ld_xy], py = mp_pos[g * ld_xy + 1];
    if (ty == 4 || ty == 5 || ty == 7) {
      const int s = atomicAdd(&cnt[0], 1);
      float4 v = make_float4(px, py, px + mp_dir[g * ld_xy], py + mp_dir[g * ld_xy + 1]);
      reinterpret_cast<float4*>(seg)[(int64_t)b * cap + s] = v;
    } else if (ty <= 2) {
      const int s = atomicAdd(&cnt[1], 1);
      reinterpret_cast<float2*>(lane)[(int64_t)b * cap + s] = make_float2(px, py);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    n_seg[b] = cnt[0];
    n_lane[b] = cnt[1];
  }
}

// ---------------------------------------------------------------------------------------------- a uniform grid over a scene's tables
// tbx_rule_grid (round 5). rule_check_kernel scanned ALL of a scene's road-edge segments (3,363 at 1024 polylines) and lane nodes
// (3,008) for every (frame, vehicle): 1.4 ms per 56 frames x 4096 agents, 8 % on top of the WOSAC-shape rollout. Both consumers are
// any() reductions of LOCAL predicates - a segment can only cross a box whose circumscribed circle it touches, a lane node only
// counts within 2 m - so the tables are sorted into the cells of a GRID x GRID raster over their bounding box (by segment midpoint /
// node position) and a wave visits the cell rows its query disc overlaps (+ one cell of slack on every side): the same predicates on
// a superset of every element that can satisfy them - bit-identical flags, ~30 instead of ~6,400 tests per vehicle.
// One workgroup per scene: bounding box + longest segment, per-cell counts (LDS atomics), prefix sums, scatter.
constexpr int GRID = 32;
constexpr int CELLS = GRID * GRID;

__device__ __forceinline__ int cell_of(float v, float v0, float inv) {
  const float c = fminf(fmaxf((v - v0) * inv, 0.f), (float)(GRID - 1));
  return (int)c;
}

template <int STRIDE>  // 4: segments (x0, y0, x1, y1), keyed by their midpoint; 2: points
__global__ __launch_bounds__(256) void rule_grid_kernel(const float* __restrict__ src, const int32_t* __restrict__ n_items, int cap,
                                                        float* __restrict__ dst, int32_t* __restrict__ start, float* __restrict__ hdr) {
  __shared__ float red[5][256];
  __shared__ int cnt[CELLS], cur[CELLS];
  __shared__ float gx0, gy0, ginv;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_items[b];
  const float* in = src + (int64_t)b * cap * STRIDE;
  float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY, lmax = 0.f;
  for (int i = tid; i < n; i += 256) {
    float kx, ky;
    if (STRIDE == 4) {
      const float4 v = reinterpret_cast<const float4*>(in)[i];
      kx = 0.5f * (v.x + v.z), ky = 0.5f * (v.y + v.w);
      lmax = fmaxf(lmax, sqrtf((v.z - v.x) * (v.z - v.x) + (v.w - v.y) * (v.w - v.y)));
    } else {
      const float2 v = reinterpret_cast<const float2*>(in)[i];
      kx = v.x, ky = v.y;
    }
    mnx = fminf(mnx, kx), mny = fminf(mny, ky), mxx = fmaxf(mxx, kx), mxy = fmaxf(mxy, ky);
  }
  red[0][tid] = mnx, red[1][tid] = mny, red[2][tid] = mxx, red[3][tid] = mxy, red[4][tid] = lmax;
  for (int c = tid; c < CELLS; c += 256) cnt[c] = 0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      red[0][tid] = fminf(red[0][tid], red[0][tid + st]);
      red[1][tid] = fminf(red[1][tid], red[1][tid + st]);
      red[2][tid] = fmaxf(red[2][tid], red[2][tid + st]);
      red[3][tid] = fmaxf(red[3][tid], red[3][tid + st]);
      red[4][tid] = fmaxf(red[4][tid], red[4][tid + st]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float ext = n > 0 ? fmaxf(fmaxf(red[2][0] - red[0][0], red[3][0] - red[1][0]), 1.f) : 1.f;
    gx0 = n > 0 ? red[0][0] : 0.f, gy0 = n > 0 ? red[1][0] : 0.f;
    ginv = (float)GRID / (ext * 1.0001f);
    float* h = hdr + (int64_t)b * 4;
    h[0] = gx0, h[1] = gy0, h[2] = ginv, h[3] = 0.5f * red[4][0];  // origin, cells per metre, half of the longest segment
  }
  __syncthreads();
  auto cell = [&](int i) {
    float kx, ky;
    if (STRIDE == 4) {
      const float4 v = reinterpret_cast<const float4*>(in)[i];
      kx = 0.5f * (v.x + v.z), ky = 0.5f * (v.y + v.w);
    } else {
      const float2 v = reinterpret_cast<const float2*>(in)[i];
      kx = v.x, ky = v.y;
    }
    return cell_of(ky, gy0, ginv) * GRID + cell_of(kx, gx0, ginv);
  };
  for (int i = tid; i < n; i += 256) atomicAdd(&cnt[cell(i)], 1);
  __syncthreads();
  if (tid == 0) {
    int32_t* st_ = start + (int64_t)b * (CELLS + 1);
    int run = 0;
    for (int c = 0; c < CELLS; ++c) {
      st_[c] = run, cur[c] = run;
      run += cnt[c];
    }
    st_[CELLS] = run;
  }
  __syncthreads();
  float* out = dst + (int64_t)b * cap * STRIDE;
  for (int i = tid; i < n; i += 256) {
    const int pos = atomicAdd(&cur[cell(i)], 1);
    if (STRIDE == 4) reinterpret_cast<float4*>(out)[pos] = reinterpret_cast<const float4*>(in)[i];
    else reinterpret_cast<float2*>(out)[pos] = reinterpret_cast<const float2*>(in)[i];
  }
}

// ---------------------------------------------------------------------------------------------- per-frame checks
struct Frame {
  float x[MAX_AG], y[MAX_AG], c[MAX_AG], s[MAX_AG];
  float bb[MAX_AG][8];  // collision box (size x scale), corners 0..3 (x, y)
  float wb[MAX_AG][8];  // wosac core box (scaled size shrunk by the rounding radius)
  float shrink[MAX_AG];
  uint8_t valid[MAX_AG], type[MAX_AG];
};

// wosac_collision.py:20-47
__device__ __forceinline__ void box_corners(float x, float y, float c, float s, float l, float w, float* o) {
  const float hl = 0.5f * l, hw = 0.5f * w;
  const float ofx = hl * c, ofy = hl * s;
  const float orx = hw * s, ory = hw * (-c);
  o[0] = x + (ofx - orx);
  o[1] = y + (ofy - ory);
  o[2] = x + (-ofx - orx);
  o[3] = y + (-ofy - ory);
  o[4] = x + (-ofx + orx);
  o[5] = y + (-ofy + ory);
  o[6] = x + (ofx + orx);
  o[7] = y + (ofy + ory);
}

// traffic_rule_checker.py:127-148: some edge line of box p has all four corners of box q strictly on its outer side
__device__ __forceinline__ bool separated_by_edge_of(const float* p, const float* q) {
  bool any_e = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float cx = p[2 * e], cy = p[2 * e + 1];
    const float nx = p[2 * ((e + 1) & 3)], ny = p[2 * ((e + 1) & 3) + 1];
    const float la = ny - cy, lb = cx - nx, lc = nx * cy - ny * cx;
    bool all_out = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) all_out = all_out && (((la * q[2 * k] + lb * q[2 * k + 1]) + lc) > 0.f);
    any_e = any_e || all_out;
  }
  return any_e;
}

// wosac_collision.py:144-179: index of the lowest corner (first on ties) and the unit direction of the edge leaving it
__device__ __forceinline__ int downmost(const float* b, float sgn, float* dx, float* dy) {
  int i0 = 0;
  float best = sgn * b[1];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float v = sgn * b[2 * k + 1];
    if (v < best) {
      best = v;
      i0 = k;
    }
  }
  const int i1 = (i0 + 1) & 3;
  const float ex = sgn * b[2 * i1] - sgn * b[2 * i0], ey = sgn * b[2 * i1 + 1] - sgn * b[2 * i0 + 1];
  const float ln = norm2(ex, ey);
  *dx = ex / ln;
  *dy = ey / ln;
  return i0;
}

// wosac_collision.py:50-116,182-250: signed distance from the origin to the Minkowski sum of box1 and (-box2)
__device__ __forceinline__ float wosac_pair_distance(const float* b1, const float* b2) {
  float d1x, d1y, d2x, d2y;
  const int s1 = downmost(b1, 1.f, &d1x, &d1y);
  const int s2 = downmost(b2, -1.f, &d2x, &d2y);
  const bool cond = (d1x * d2y - d1y * d2x) >= 0.f;
  float px[8], py[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int oa = k >> 1, ob = ((k + 1) >> 1) & 3;  // [0,0,1,1,2,2,3,3] and [0,1,1,2,2,3,3,0]
    const int i1 = ((cond ? ob : oa) + s1) & 3;
    const int i2 = ((cond ? oa : ob) + s2) & 3;
    px[k] = b1[2 * i1] + (-b2[2 * i2]);
    py[k] = b1[2 * i1 + 1] + (-b2[2 * i2 + 1]);
  }
  bool inside = true;
  float m = INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int k1 = (k + 1) & 7;
    const float ex = px[k1] - px[k], ey = py[k1] - py[k];
    const float ln = norm2(ex, ey);
    const float tx = ex / ln, ty = ey / ln;
    const float nx = -ty, ny = tx;
    const float qx = 0.f - px[k], qy = 0.f - py[k];
    const float dv = norm2(qx, qy);
    const float perp = (-nx) * qx + (-ny) * qy;
    inside = inside && (perp <= 0.f);
    const float prop = (tx * qx + ty * qy) / ln;
    const float de = (prop >= 0.f && prop <= 1.f) ? fabsf(perp) : (0.f + BIG);
    m = fminf(m, fminf(de, dv));
  }
  return inside ? -m : m;
}

// traffic_rule_checker.py:504-505
__device__ __forceinline__ bool ccw(float ax, float ay, float bx, float by, float cx, float cy) {
  return (cy - ay) * (bx - ax) > (by - ay) * (cx - ax);
}

struct RuleArgs {
  tbx_rule_ctx_t c;
  const uint8_t* valid;
  const float* pose;
  const float* motion;
  const uint8_t* tl_state;
  uint8_t* flags;
  int ld_t, t0, n_t, tiles;
};

__global__ __launch_bounds__(256) void rule_check_kernel(const RuleArgs a) {
  __shared__ Frame F;
  const tbx_rule_ctx_t& c = a.c;
  const int A = c.n_ag, L = c.n_tl;
  const int tile = blockIdx.x % a.tiles;
  const int frame = blockIdx.x / a.tiles;
  const int b = frame / a.n_t, t = a.t0 + frame % a.n_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- stage the frame's agents
  for (int j = threadIdx.x; j < A; j += blockDim.x) {
    const int64_t r = ((int64_t)b * A + j) * a.ld_t + t;
    const float x = a.pose[r * 3], y = a.pose[r * 3 + 1], yaw = a.pose[r * 3 + 2];
    const float cs = cosf(yaw), sn = sinf(yaw);
    const float* sz = c.ag_size + ((int64_t)b * A + j) * 3;
    const float l = sz[0] * c.collision_size_scale, w = sz[1] * c.collision_size_scale;
    F.x[j] = x;
    F.y[j] = y;
    F.c[j] = cs;
    F.s[j] = sn;
    box_corners(x, y, cs, sn, l, w, F.bb[j]);
    const float sh = fminf(l, w) * 0.7f / 2.0f;
    F.shrink[j] = sh;
    box_corners(x, y, cs, sn, l - 2.0f * sh, w - 2.0f * sh, F.wb[j]);
    F.valid[j] = a.valid[r];
    F.type[j] = c.ag_type_idx[(int64_t)b * A + j];
  }
  __syncthreads();
  const int i = tile * 4 + wave;
  if (i >= A) return;
  const int64_t ri = ((int64_t)b * A + i) * a.ld_t + t;
  const bool valid_i = F.valid[i] != 0;
  const int type_i = F.type[i];
  const bool veh = type_i == 0;
  const float xi = F.x[i], yi = F.y[i], ci = F.c[i], si = F.s[i];
  const float spd = a.motion[ri * 3];
  float bi[8], wi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bi[k] = F.bb[i][k];
    wi[k] = F.wb[i][k];
  }
  const float shrink_i = F.shrink[i];
  // ---- agent pairs: collided / collided_wosac / an agent ahead (passive)
  bool collided = false, wosac = false, ag_ahead = false;
  for (int j = lane; j < A; j += 64) {
    const bool pair_ok = valid_i && F.valid[j] != 0 && j != i;
    if (!pair_ok) continue;
    float bj[8], wj[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bj[k] = F.bb[j][k];
      wj[k] = F.wb[j][k];
    }
    if (!(type_i == 1 && F.type[j] == 1))  // no pedestrian-pedestrian collisions (traffic_rule_checker.py:47-49)
      collided = collided || !(separated_by_edge_of(bi, bj) || separated_by_edge_of(bj, bi));
    const float sd = (wosac_pair_distance(wi, wj) - F.shrink[j]) - shrink_i;
    wosac = wosac || (sd < 0.f);
    const float wx = F.x[j] - xi, wy = F.y[j] - yi;
    const float wn = norm2(wx, wy);
    ag_ahead = ag_ahead || ((wn < 10.f) && (((ci * wx + si * wy) / wn) > 0.95f));
  }
  // ---- lights: run_red_light / a non-green light ahead (passive)
  bool red = false, red_ahead = false;
  {
    const float* sz = c.ag_size + ((int64_t)b * A + i) * 3;
    const float half_len = sz[0] * 0.5f * 0.6f, half_wid = sz[1] * 0.5f * 1.8f;
    const float rx = si, ry = -ci;
    const float adv = 0.1f * spd;
    const float x1 = xi + adv * ci, y1 = yi + adv * si;
    for (int k = lane; k < L; k += 64) {
      const int64_t lk = (int64_t)b * L + k;
      if (!c.tl_valid[lk]) continue;
      const uint8_t st = a.tl_state[lk * a.ld_t + t];
      const float tx = c.tl_pose[lk * 3], ty = c.tl_pose[lk * 3 + 1];
      if ((st & 2) && valid_i && veh) {
        const float d0x = tx - xi, d0y = ty - yi, d1x = tx - x1, d1y = ty - y1;
        const bool in0 = (fabsf(d0x * ci + d0y * si) < half_len) && (fabsf(d0x * rx + d0y * ry) < half_wid);
        const bool in1 = (fabsf(d1x * ci + d1y * si) < half_len) && (fabsf(d1x * rx + d1y * ry) < half_wid);
        red = red || (in0 && !in1);
      }
      if (st & 0x17) {  // UNKNOWN | STOP | CAUTION | FLASHING
        const float vx = tx - xi, vy = ty - yi;
        const float vn = norm2(vx, vy);
        red_ahead = red_ahead || ((vn < 10.f) && (((ci * vx + si * vy) / vn) > 0.95f));
      }
    }
  }
  collided = __any(collided);
  wosac = __any(wosac);
  ag_ahead = __any(ag_ahead);
  red = __any(red);
  red_ahead = __any(red_ahead);
  // ---- the scene's static tables (vehicles only)
  bool edge = false, passive = false;
  if (valid_i && veh) {
    const int sc = b / c.map_batch_div;
    const float4* seg = reinterpret_cast<const float4*>(c.seg) + (int64_t)sc * c.cap;
    auto seg_range = [&](int k_begin, int k_end) {  // any segment of [k_begin, k_end) crossing an edge of the box?
      for (int k0 = k_begin; k0 < k_end; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < k_end) {
          const float4 s = seg[k];  // C = (x, y), D = (z, w)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ax = bi[2 * e], ay = bi[2 * e + 1];
            const float bx = bi[2 * ((e + 1) & 3)], by = bi[2 * ((e + 1) & 3) + 1];
            hit = hit || ((ccw(ax, ay, s.x, s.y, s.z, s.w) != ccw(bx, by, s.x, s.y, s.z, s.w)) &&
                          (ccw(ax, ay, bx, by, s.x, s.y) != ccw(ax, ay, bx, by, s.z, s.w)));
          }
        }
        if (__any(hit)) return true;
      }
      return false;
    };
    const float2* lc = reinterpret_cast<const float2*>(c.lane) + (int64_t)sc * c.cap;
    auto lane_range = [&](int k_begin, int k_end) {  // any lane node of [k_begin, k_end) within 2 m?
      for (int k0 = k_begin; k0 < k_end; k0 += 64) {
        const int k = k0 + lane;
        bool near_lane = false;
        if (k < k_end) {
          const float2 p = lc[k];
          near_lane = norm2(xi - p.x, yi - p.y) < 2.f;
        }
        if (__any(near_lane)) return true;
      }
      return false;
    };
    // the cell rows a query disc of radius r around (xi, yi) overlaps, one cell of slack on every side (tbx_rule_grid)
    auto cells = [&](const float* h, float r, int& cx0, int& cx1, int& cy0, int& cy1) {
      const float x0 = h[0], y0 = h[1], inv = h[2];
      cx0 = max(cell_of(xi - r, x0, inv) - 1, 0), cx1 = min(cell_of(xi + r, x0, inv) + 1, GRID - 1);
      cy0 = max(cell_of(yi - r, y0, inv) - 1, 0), cy1 = min(cell_of(yi + r, y0, inv) + 1, GRID - 1);
    };
    if (c.seg_start != nullptr) {
      // a segment that crosses an edge of the box has a point inside the box's circumscribed circle; its midpoint is at most half
      // of the scene's longest segment further away
      const float* h = c.seg_grid + (int64_t)sc * 4;
      float r2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) r2 = fmaxf(r2, (bi[2 * e] - xi) * (bi[2 * e] - xi) + (bi[2 * e + 1] - yi) * (bi[2 * e + 1] - yi));
      const float r = sqrtf(r2) * 1.001f + h[3] + 1e-3f;
      const int32_t* st_ = c.seg_start + (int64_t)sc * (CELLS + 1);
      int cx0, cx1, cy0, cy1;
      cells(h, r, cx0, cx1, cy0, cy1);
      for (int cy = cy0; cy <= cy1 && !edge; ++cy) edge = seg_range(st_[cy * GRID + cx0], st_[cy * GRID + cx1 + 1]);
    } else {
      edge = seg_range(0, c.n_seg[sc]);
    }
    if ((spd < 5.f) && !red_ahead && !ag_ahead) {
      if (c.lane_start != nullptr) {
        const float* h = c.lane_grid + (int64_t)sc * 4;
        const int32_t* st_ = c.lane_start + (int64_t)sc * (CELLS + 1);
        int cx0, cx1, cy0, cy1;
        cells(h, 2.f * 1.001f + 1e-3f, cx0, cx1, cy0, cy1);
        for (int cy = cy0; cy <= cy1 && !passive; ++cy) passive = lane_range(st_[cy * GRID + cx0], st_[cy * GRID + cx1 + 1]);
      } else {
        passive = lane_range(0, c.n_lane[sc]);
      }
    }
  }
  if (lane == 0)
    a.flags[ri] = (uint8_t)((collided ? TBX_RULE_COLLIDED : 0) | (wosac ? TBX_RULE_COLLIDED_WOSAC : 0) |
                            (edge ? TBX_RULE_RUN_ROAD_EDGE : 0) | (red ? TBX_RULE_RUN_RED_LIGHT : 0) |
                            (passive ? TBX_RULE_PASSIVE : 0));
}

// traffic_rule_checker.py:293-296,352-404: the counter turns raw passivity into the passive flag, every flag is OR-ed
// over time. One thread per (rollout, agent), sequential over the steps of the range.
__global__ void rule_accumulate_kernel(const uint8_t* raw, int n_rows, int ld_t, int t0, int n_t, uint8_t* __restrict__ acc_state,
                                       float* __restrict__ counter, uint8_t* out_now, uint8_t* __restrict__ out_acc) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  uint8_t acc = acc_state[r];
  float cnt = counter[r];
  for (int t = t0; t < t0 + n_t; ++t) {
    const uint8_t f = raw[(int64_t)r * ld_t + t];
    const float p = (f & TBX_RULE_PASSIVE) ? 1.f : 0.f;
    cnt = (cnt + p) * p;
    const uint8_t now = (uint8_t)((f & ~TBX_RULE_PASSIVE) | (cnt > 20.f ? TBX_RULE_PASSIVE : 0));
    acc |= now;
    out_now[(int64_t)r * ld_t + t] = now;
    out_acc[(int64_t)r * ld_t + t] = acc;
  }
  acc_state[r] = acc;
  counter[r] = cnt;
}


// ---------------------------------------------------------------------------------------------- WOSAC rollout filter
// data_modules/wosac_post_processing.py:31-64: score_k = sum_a role_a * any_t collided[k,a,t] + w * sum_a role_a * any_t
// run_road_edge[k,a,t] (t >= t_start), keep the n_keep rollouts with the smallest score. One workgroup per scene: a wave per
// rollout counts its flagged agents (lanes over agents, bytes along t are contiguous), then every rollout ranks itself by
// (score, index) against the K scores in LDS; the kept indices come out in ascending (score, index) order.
constexpr int MAX_K = 1024;

__global__ __launch_bounds__(256) void filter_score_kernel(const uint8_t* __restrict__ flags, int col_bit,
                                                           const uint8_t* __restrict__ role_any, int K, int A, int ld_t,
                                                           int t_start, float w_road_edge, int n_keep,
                                                           float* __restrict__ score, int32_t* __restrict__ idx) {
  __shared__ float sc[MAX_K];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < K; k += 4) {
    int n_col = 0, n_edge = 0;
    for (int a = lane; a < A; a += 64) {
      if (!role_any[(int64_t)b * A + a]) continue;
      const uint8_t* f = flags + (((int64_t)b * K + k) * A + a) * ld_t;
      uint8_t m = 0;
      for (int t = t_start; t < ld_t; ++t) m |= f[t];
      n_col += (m & col_bit) ? 1 : 0;
      n_edge += (m & TBX_RULE_RUN_ROAD_EDGE) ? 1 : 0;
    }
    n_col = (int)tbx::wave_sum((float)n_col);
    n_edge = (int)tbx::wave_sum((float)n_edge);
    if (lane == 0) {
      const float v = (float)n_col + (float)n_edge * w_road_edge;
      sc[k] = v;
      score[(int64_t)b * K + k] = v;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float v = sc[k];
    int rank = 0;
    for (int j = 0; j < K; ++j) rank += (sc[j] < v || (sc[j] == v && j < k)) ? 1 : 0;
    if (rank < n_keep) idx[(int64_t)b * n_keep + rank] = k;
  }
}

__global__ void filter_gather_kernel(const float* __restrict__ pose, const int32_t* __restrict__ idx, int K, int A, int ld_t,
                                     int t_start, int n_keep, int64_t total, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int per = (ld_t - t_start) * 3;  // floats of one (rollout, agent) future
  const int64_t row = i / per;           // (scene, kept, agent)
  const int c = (int)(i % per);
  const int a = (int)(row % A);
  const int64_t sj = row / A;
  const int b = (int)(sj / n_keep);
  const int k = idx[sj];
  out[i] = pose[((((int64_t)b * K + k) * A + a) * ld_t + t_start) * 3 + c];
}

// ---------------------------------------------------------------------------------------------- outside-map / destination-reached
// TrafficRuleChecker._check_outside_map / _check_dest_reached (traffic_rule_checker.py:109-120,300-330) for ONE step: the two checks
// that feed back into the simulation. The rollout engine evaluates them inside tbx_sim_step; this stand-alone form serves
// `TrafficRuleChecker.check` as the reference's `rollout` calls it between two `WaymoMotion.forward`s. One thread per agent.
__global__ void rule_navi_kernel(const uint8_t* __restrict__ valid, const float* __restrict__ pose, const float* __restrict__ boundary,
                                 int map_batch_div, const uint8_t* __restrict__ dest_invalid, const float* __restrict__ dest_pos,
                                 const float* __restrict__ dest_dir, const uint8_t* __restrict__ dest_kind,
                                 const float* __restrict__ dest_thresh, const float* __restrict__ goal, const float* __restrict__ goal_thresh,
                                 int n_rows, int n_ag, int n_node, uint8_t* __restrict__ acc, uint8_t* __restrict__ out_now) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const bool v = valid[i] != 0;
  const float x = pose[i * 3], y = pose[i * 3 + 1], yaw = pose[i * 3 + 2];
  const float* bd = boundary + (int64_t)(i / n_ag / map_batch_div) * 4;  // xmin, xmax, ymin, ymax
  const bool out_now_ = v && (x > bd[1] || x < bd[0] || y > bd[3] || y < bd[2]);
  bool reach_now = false;
  if (dest_invalid != nullptr) {
    const float hx = cosf(yaw), hy = sinf(yaw);
    const float thresh = dest_thresh[i];
    bool pos_ok = false, rot_ok = false;
    for (int k = 0; k < n_node; ++k) {
      const int64_t d = (int64_t)i * n_node + k;
      if (dest_invalid[d]) continue;
      pos_ok = pos_ok || norm2(__fsub_rn(x, dest_pos[d * 2]), __fsub_rn(y, dest_pos[d * 2 + 1])) < thresh;
      rot_ok = rot_ok || __fadd_rn(__fmul_rn(hx, dest_dir[d * 2]), __fmul_rn(hy, dest_dir[d * 2 + 1])) > 0.8660254037844387f;
    }
    const uint8_t kind = dest_kind[i];  // bit 0: lane destination (position and heading), bit 1: road-edge destination (position)
    reach_now = acc[n_rows + i] == 0 && v && (((kind & 1) && pos_ok && rot_ok) || ((kind & 2) && pos_ok));
  }
  bool goal_now = false;
  if (goal != nullptr) {  // _check_goal_reached (:277-288): within 8 agent lengths and 15 degrees of (x, y, yaw) of the goal
    const float* g = goal + (int64_t)i * 4;
    const bool pos_ok = norm2(__fsub_rn(x, g[0]), __fsub_rn(y, g[1])) < goal_thresh[i];
    // cast_rad (transform_utils.py:9-11): torch's remainder = fmod, moved into the divisor's sign
    const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
    float m = fmodf(__fadd_rn(__fsub_rn(yaw, g[2]), pi), two_pi);
    if (m != 0.f && m < 0.f) m = __fadd_rn(m, two_pi);
    const bool rot_ok = fabsf(__fsub_rn(m, pi)) < 0.2617993877991494f;
    goal_now = pos_ok && rot_ok && v && acc[2 * n_rows + i] == 0;
  }
  out_now[i] = out_now_ ? 1 : 0;
  out_now[n_rows + i] = reach_now ? 1 : 0;
  out_now[2 * n_rows + i] = goal_now ? 1 : 0;
  if (out_now_) acc[i] = 1;
  if (reach_now) acc[n_rows + i] = 1;
  if (goal_now) acc[2 * n_rows + i] = 1;
}

}  // namespace

extern "C" int tbx_rule_navi_check(const uint8_t* valid, const float* pose, const float* boundary, int map_batch_div,
                                   const uint8_t* dest_invalid, const float* dest_pos, const float* dest_dir, const uint8_t* dest_kind,
                                   const float* dest_thresh, const float* goal, const float* goal_thresh, int n_batch, int n_ag, int n_node,
                                   uint8_t* acc, uint8_t* out_now, void* stream) {
  if (!valid || !pose || !boundary || !acc || !out_now) return TBX_ERR_ARG;
  if ((goal != nullptr) != (goal_thresh != nullptr)) return TBX_ERR_ARG;
  if (n_batch <= 0 || n_ag <= 0 || map_batch_div <= 0 || n_batch % map_batch_div) return TBX_ERR_ARG;
  const bool dest = dest_invalid != nullptr;
  if (dest != (dest_pos != nullptr) || dest != (dest_dir != nullptr) || dest != (dest_kind != nullptr) || dest != (dest_thresh != nullptr))
    return TBX_ERR_ARG;
  if (dest && n_node <= 0) return TBX_ERR_ARG;
  const int64_t rows = (int64_t)n_batch * n_ag;
  if (rows > 0x7fffffff) return TBX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(rule_navi_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, (hipStream_t)stream, valid, pose, boundary,
                     map_batch_div, dest_invalid, dest_pos, dest_dir, dest_kind, dest_thresh, goal, goal_thresh, (int)rows, n_ag, n_node, acc, out_now);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_rule_tables(const uint8_t* mp_valid, const uint8_t* mp_type_idx, const float* mp_pos, const float* mp_dir,
                               int ld_xy, int n_scene, int n_mp, int n_node, float* seg, int32_t* n_seg, float* lane,
                               int32_t* n_lane, void* stream) {
  if (!mp_valid || !mp_type_idx || !mp_pos || !mp_dir || !seg || !n_seg || !lane || !n_lane) return TBX_ERR_ARG;
  if (n_scene <= 0 || n_mp <= 0 || n_node <= 0 || ld_xy < 2) return TBX_ERR_ARG;
  if (((uintptr_t)seg & 15) || ((uintptr_t)lane & 7)) return TBX_ERR_ALIGN;
  hipLaunchKernelGGL(rule_tables_kernel, dim3(n_scene), dim3(1024), 0, (hipStream_t)stream, mp_valid, mp_type_idx, mp_pos,
                     mp_dir, ld_xy, n_mp, n_node, seg, n_seg, lane, n_lane);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_rule_grid(const float* seg, const int32_t* n_seg, const float* lane, const int32_t* n_lane, int n_scene, int cap,
                             float* seg_sorted, int32_t* seg_start, float* seg_grid, float* lane_sorted, int32_t* lane_start, float* lane_grid,
                             void* stream) {
  if (!seg || !n_seg || !lane || !n_lane || !seg_sorted || !seg_start || !seg_grid || !lane_sorted || !lane_start || !lane_grid) return TBX_ERR_ARG;
  if (n_scene <= 0 || cap <= 0) return TBX_ERR_ARG;
  if (((uintptr_t)seg & 15) || ((uintptr_t)seg_sorted & 15) || ((uintptr_t)lane & 7) || ((uintptr_t)lane_sorted & 7)) return TBX_ERR_ALIGN;
  hipLaunchKernelGGL(rule_grid_kernel<4>, dim3(n_scene), dim3(256), 0, (hipStream_t)stream, seg, n_seg, cap, seg_sorted, seg_start, seg_grid);
  if (hipGetLastError() != hipSuccess) return TBX_ERR_LAUNCH;
  hipLaunchKernelGGL(rule_grid_kernel<2>, dim3(n_scene), dim3(256), 0, (hipStream_t)stream, lane, n_lane, cap, lane_sorted, lane_start, lane_grid);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_rule_grid_cells(void) { return CELLS; }

extern "C" int tbx_rule_check(const tbx_rule_ctx_t* ctx, const uint8_t* valid, const float* pose, const float* motion,
                              const uint8_t* tl_state, int ld_t, int t0, int n_t, uint8_t* flags, void* stream) {
  if (!ctx || !valid || !pose || !motion || !tl_state || !flags) return TBX_ERR_ARG;
  const tbx_rule_ctx_t& c = *ctx;
  if (!c.seg || !c.n_seg || !c.lane || !c.n_lane || !c.ag_size || !c.ag_type_idx || !c.tl_valid || !c.tl_pose) return TBX_ERR_ARG;
  if (c.n_batch <= 0 || c.n_ag <= 0 || c.n_tl <= 0 || c.map_batch_div <= 0 || c.cap <= 0 || ld_t <= 0 || t0 < 0 || n_t <= 0 ||
      t0 + n_t > ld_t || c.n_batch % c.map_batch_div)
    return TBX_ERR_ARG;
  if (c.n_ag > MAX_AG) return TBX_ERR_UNSUPPORTED;
  if (((uintptr_t)c.seg & 15) || ((uintptr_t)c.lane & 7)) return TBX_ERR_ALIGN;
  if ((c.seg_start != nullptr) != (c.seg_grid != nullptr) || (c.lane_start != nullptr) != (c.lane_grid != nullptr)) return TBX_ERR_ARG;
  RuleArgs a{c, valid, pose, motion, tl_state, flags, ld_t, t0, n_t, (c.n_ag + 3) / 4};
  const int64_t blocks = (int64_t)c.n_batch * n_t * a.tiles;
  if (blocks > 0x7fffffff) return TBX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(rule_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_rule_accumulate(const uint8_t* raw, int n_rows, int ld_t, int t0, int n_t, uint8_t* acc_state,
                                   float* passive_counter, uint8_t* out_now, uint8_t* out_acc, void* stream) {
  if (!raw || !acc_state || !passive_counter || !out_now || !out_acc) return TBX_ERR_ARG;
  if (n_rows <= 0 || ld_t <= 0 || t0 < 0 || n_t <= 0 || t0 + n_t > ld_t) return TBX_ERR_ARG;
  hipLaunchKernelGGL(rule_accumulate_kernel, dim3((n_rows + 127) / 128), dim3(128), 0, (hipStream_t)stream, raw, n_rows, ld_t,
                     t0, n_t, acc_state, passive_counter, out_now, out_acc);
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}

extern "C" int tbx_filter_futures(const uint8_t* flags, int col_bit, const uint8_t* ag_role_any, int n_scene, int n_k, int n_ag,
                                  int ld_t, int t_start, float w_road_edge, int n_keep, float* score, int32_t* idx,
                                  const float* pred_pose, float* trajs, void* stream) {
  if (!flags || !ag_role_any || !score || !idx) return TBX_ERR_ARG;
  if (n_scene <= 0 || n_k <= 0 || n_ag <= 0 || ld_t <= 0 || t_start < 0 || t_start > ld_t || n_keep <= 0 || n_keep > n_k)
    return TBX_ERR_ARG;
  if (col_bit != TBX_RULE_COLLIDED && col_bit != TBX_RULE_COLLIDED_WOSAC) return TBX_ERR_ARG;
  if (n_k > MAX_K) return TBX_ERR_UNSUPPORTED;
  if ((pred_pose == nullptr) != (trajs == nullptr)) return TBX_ERR_ARG;
  hipStream_t hs = (hipStream_t)stream;
  hipLaunchKernelGGL(filter_score_kernel, dim3(n_scene), dim3(256), 0, hs, flags, col_bit, ag_role_any, n_k, n_ag, ld_t, t_start,
                     w_road_edge, n_keep, score, idx);
  if (hipGetLastError() != hipSuccess) return TBX_ERR_LAUNCH;
  if (pred_pose && ld_t > t_start) {
    const int64_t total = (int64_t)n_scene * n_keep * n_ag * (ld_t - t_start) * 3;
    hipLaunchKernelGGL(filter_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, hs, pred_pose, idx, n_k, n_ag,
                       ld_t, t_start, n_keep, total, trajs);
  }
  return hipGetLastError() == hipSuccess ? TBX_OK : TBX_ERR_LAUNCH;
}
