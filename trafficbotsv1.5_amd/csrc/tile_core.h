// Device building blocks of the tile kernels (tile_layer.hip, tile_heads.hip, tile_window.hip): LINEAR stages on the split-bf16
// matrix path over 16-row tiles whose activations live in LDS as bf16 hi / lo planes. See tile_layer.hip for the design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbx_hip.h"
#include "tbx_common.h"

// TBX_TILE_SINGLE (tile_layer / tile_heads / tile_window are compiled a second time with it: the *_bf16 entry points,
// Schedule.linear_bf16): ONE bf16 product per LINEAR - weights and activations rounded to bfloat16, fp32 accumulation; the lo halves of
// the weight units are not fetched (a unit's 8 KiB per wave go through the CU's 64 B/clk L1 port whether they hit or not) and no lo
// planes are written or read.
#ifndef TBX_TILE_SINGLE
#define TBX_TILE_SINGLE 0
#endif
#if TBX_TILE_SINGLE
#define TBX_TILE_ENTRY(name) name##_bf16
#else
#define TBX_TILE_ENTRY(name) name
#endif

namespace tbx_tile {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define TBX_GLOBAL __attribute__((address_space(1)))

constexpr int NWAVE = 8, NT = NWAVE * 64, D = 128;
constexpr int UNIT = TBX_MFMA32_UNIT_FLOATS;  // 2112 floats: 4 groups x (hi 1 KiB | lo 1 KiB) + 4 x 16 bias floats

// bf16 planes of R rows x (32 * STEPS) channels: element (row, k) of a plane lives at byte
//   ((k >> 3) & 3) * PREG + row * PRS + (k >> 5) * 16 + (k & 7) * 2:
// the four 8-element octets of a 32-k step go to four regions (a multiple of 256 B apart, so the octet does not move the bank),
// inside a region a row is PRS bytes with PRS / 16 odd: the 16 rows of an operand read land in 16 distinct 16-byte bank groups
// whichever 16 lanes the LDS serves together (MI355X_MICROARCH.md, LDS: ds_read_b128 = 4 groups of 16 lanes). The lo plane
// follows the hi plane at + PLANE.
template <int R, int STEPS>
struct Planes {
  static constexpr int PRS = 16 * (STEPS | 1);  // odd number of 16-byte slots per row
  static constexpr int PREG = ((R * PRS + 255) / 256) * 256;
  static constexpr int PLANE = 4 * PREG;
  static __device__ __forceinline__ int off(int row, int k) { return ((k >> 3) & 3) * PREG + row * PRS + (k >> 5) * 16 + (k & 7) * 2; }
  // the (octet, row) offset of MFMA lane l for the row tile starting at row r0: B operand element [k-octet l >> 4][row l & 15]
  static __device__ __forceinline__ int lane_off(int lane, int r0) { return (lane >> 4) * PREG + (r0 + (lane & 15)) * PRS; }
};

struct Entry {
  const float* img;
  int32_t unit0, pad;
};

// one wave's unit of weights: 4 groups of (hi, lo) A fragments (lane l, element e = W[tile * 16 + (l & 15)][step * 32 + (l >> 4) * 8 + e])
struct W {
  bf16x8 hi[4], lo[4];
  f32x4 bias;  // group 0's tile: the lane's 4 output channels
};

__device__ __forceinline__ const TBX_GLOBAL float* unit_ptr(const float* img, int unit) {
  return (const TBX_GLOBAL float*)img + (int64_t)unit * UNIT;
}
__device__ __forceinline__ void load_unit(W& w, const float* img, int unit, int lane) {
  const TBX_GLOBAL float* base = unit_ptr(img, unit);
#ifdef TBX_ABL_NOLOAD  // (ablation builds, tools/scratch/tile_ablation.sh: every unit is the image's first 8 KiB - L1-resident)
  base = unit_ptr(img, 0);
#endif
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    w.hi[s] = *(const TBX_GLOBAL bf16x8*)(base + s * 512 + lane * 4);
#if !TBX_TILE_SINGLE
    w.lo[s] = *(const TBX_GLOBAL bf16x8*)(base + s * 512 + 256 + lane * 4);
#endif
  }
  w.bias = *(const TBX_GLOBAL f32x4*)(base + 2048 + (lane >> 4) * 4);
}
__device__ __forceinline__ void load_unit(W& w, const Entry& e, int wave, int lane) { load_unit(w, e.img, e.unit0 + wave, lane); }
// the lane's bias of group s (units whose groups are different tiles: k = 32 / 64 images)
__device__ __forceinline__ f32x4 unit_bias(const float* img, int unit, int s, int lane) {
  return *(const TBX_GLOBAL f32x4*)(unit_ptr(img, unit) + 2048 + s * 16 + (lane >> 4) * 4);
}

struct Acc {
  f32x4 hh, hl, lh;
  __device__ __forceinline__ void zero() { hh = hl = lh = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ f32x4 sum() const { return hh + (hl + lh); }
};

// D[channel][row] += W[channel][k] x[row][k] over the 32 k of one step: A = weight fragment, B = activation fragment read from
// the planes at `act_hi` (= hi plane + the lane's (octet, row) offset) + step * 16; the lo plane is PLANE bytes behind
template <int PLANE>
__device__ __forceinline__ void mfma_step(Acc& a, const bf16x8 whi, const bf16x8 wlo, const char* act_hi, int step) {
  const bf16x8 xh = *(const bf16x8*)(act_hi + step * 16);
#if TBX_TILE_SINGLE
  a.hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xh, a.hh, 0, 0, 0);
  return;
#endif
  const bf16x8 xl = *(const bf16x8*)(act_hi + PLANE + step * 16);
#ifdef TBX_ABL_NOMFMA  // (ablation builds: one product instead of three)
  a.hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xh, a.hh, 0, 0, 0);
  a.hl[0] += (float)xl[0];
  a.lh[0] += (float)wlo[0];
#else
  a.hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xh, a.hh, 0, 0, 0);
  a.hl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xl, a.hl, 0, 0, 0);
  a.lh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, a.lh, 0, 0, 0);
#endif
}

// 4 fp32 values -> bf16 hi (RNE) and lo = bf16(v - hi), 8 bytes each
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  const bf16x4 h = __builtin_convertvector(v, bf16x4);
  const f32x4 r = v - __builtin_convertvector(h, f32x4);
  const bf16x4 l = __builtin_convertvector(r, bf16x4);
  hi = __builtin_bit_cast(u32x2, h);
  lo = __builtin_bit_cast(u32x2, l);
}

// 4 consecutive channels [c, c + 4) (c % 4 == 0) of plane row `row` into a plane pair (hi at p, lo at p + PLANE)
template <class PL>
__device__ __forceinline__ void planes_write4(char* p, int row, int c, const f32x4 v) {
  const int o = PL::off(row, c);
#if TBX_TILE_SINGLE
  *(u32x2*)(p + o) = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
#else
  u32x2 hi, lo;
  split4(v, hi, lo);
  *(u32x2*)(p + o) = hi;
  *(u32x2*)(p + PL::PLANE + o) = lo;
#endif
}

// tbx_keyed_dropout's mask (csrc/dropout.hip, rowchain.hip op_dropout) on the lane's 4 consecutive columns [c, c + 4) of global
// row `row` of an n-wide block: keep = hash(seed, site, step, row * n + column) >= thresh
struct DropKey4 {
  uint32_t lo, hi, thresh;
  float scale;
  __device__ __forceinline__ void init(const uint64_t* seed, uint32_t site, uint32_t step, uint32_t thresh_, float scale_) {
    const uint64_t sd = *(const TBX_GLOBAL uint64_t*)seed;
    lo = (uint32_t)sd ^ (site * 0x85EBCA6Bu) ^ (step * 0x27D4EB2Fu);
    hi = (uint32_t)(sd >> 32) + site * 0xC2B2AE35u + step * 0x165667B1u;
    thresh = thresh_, scale = scale_;
  }
  __device__ __forceinline__ f32x4 apply(f32x4 v, int64_t row, int c, int n) const {
    const uint32_t base = (uint32_t)row * (uint32_t)n + (uint32_t)c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t x = (base + (uint32_t)r) ^ lo;
      x *= 0x9E3779B1u;
      x ^= hi;
      x ^= x >> 16;
      x *= 0x7feb352du;
      x ^= x >> 15;
      x *= 0x846ca68bu;
      x ^= x >> 16;
      v[r] = x >= thresh ? v[r] * scale : 0.f;
    }
    return v;
  }
};

__device__ __forceinline__ f32x4 gld4(const float* p) { return *(const TBX_GLOBAL f32x4*)p; }
__device__ __forceinline__ void gst4(float* p, const f32x4 v) { *(TBX_GLOBAL f32x4*)p = v; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v[0] = fmaxf(v[0], 0.f), v[1] = fmaxf(v[1], 0.f), v[2] = fmaxf(v[2], 0.f), v[3] = fmaxf(v[3], 0.f);
  return v;
}


// ---- ONE row as the B operand (all 16 columns read the same row; column 0 = lanes with (lane & 15) == 0 is kept): bf16 planes of a
// single row, hi at P + 2 k, lo `lo` bytes behind. Used where a workgroup owns one or two rows (csrc/front.hip; csrc/dec_mid.hip
// has its own copies next to the kernel they were written for).
namespace row1 {
__device__ __forceinline__ void put4(char* P, int lo, int c, const f32x4 v) {
#if TBX_TILE_SINGLE
  *(u32x2*)(P + c * 2) = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4));
#else
  u32x2 hi, l;
  split4(v, hi, l);
  *(u32x2*)(P + c * 2) = hi;
  *(u32x2*)(P + lo + c * 2) = l;
#endif
}
__device__ __forceinline__ void step(Acc& acc, const bf16x8 whi, const bf16x8 wlo, const char* P, int lo, int st, int g4) {
  const bf16x8 xh = *(const bf16x8*)(P + (st * 32 + g4 * 8) * 2);
  acc.hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xh, acc.hh, 0, 0, 0);
#if !TBX_TILE_SINGLE
  const bf16x8 xl = *(const bf16x8*)(P + lo + (st * 32 + g4 * 8) * 2);
  acc.hl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xl, acc.hl, 0, 0, 0);
  acc.lh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, acc.lh, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 gemv4(const W& w, const char* P, int lo, int st0, int g4) {
  Acc acc;
  acc.zero();
#pragma unroll
  for (int st = 0; st < 4; ++st) step(acc, w.hi[st], w.lo[st], P, lo, st0 + st, g4);
  return acc.sum();
}
// LayerNorm of the 128-float row `src` (LDS) by one wavefront, in rowchain.hip's ln_row order -> planes (lo 256 bytes behind)
__device__ __forceinline__ void ln_planes(const float* src, char* P, int lane, float eps, const float* gamma, const float* beta) {
  float v[2], gm[2], bt[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    v[q] = src[lane + 64 * q];
    gm[q] = *(const TBX_GLOBAL float*)(gamma + lane + 64 * q);
    bt[q] = *(const TBX_GLOBAL float*)(beta + lane + 64 * q);
  }
  const float mean = tbx::wave_sum(v[0] + v[1]) / 128.f;
  const float d0 = v[0] - mean, d1 = v[1] - mean;
  const float var = tbx::wave_sum(d0 * d0 + d1 * d1) / 128.f;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float y = (v[q] - mean) * rstd * gm[q] + bt[q];
    const __bf16 h = (__bf16)y;
    *(__bf16*)(P + 2 * (lane + 64 * q)) = h;
#if !TBX_TILE_SINGLE
    *(__bf16*)(P + 256 + 2 * (lane + 64 * q)) = (__bf16)(y - (float)h);
#endif
  }
}
}  // namespace row1

// the rider's tile (tbx_layer_tile_t.rider_*): four 128 -> 128 stages on 16 rows of its own, planes ping-pong Pa <-> Pb
// (PL = the caller's plane geometry for 16 rows, K >= 128; X = 16 rows of XLD floats of LDS; Pa, Pb = plane pairs)
template <class PL, int XLD>
__device__ __forceinline__ void rider_tile(const tbx_layer_tile_t& t, int tile, float* X, char* Pa, char* Pb) {
  constexpr int ROWS = 16;
  constexpr int PLANE = PL::PLANE;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)tile * ROWS;
  const int nv = (t.rider_rows - row0) < ROWS ? (int)(t.rider_rows - row0) : ROWS;
  const bool row_ok = j < nv;
  const int64_t grow = row0 + (row_ok ? j : 0);
  const int aoff = PL::lane_off(lane, 0);
  const int c_out = 16 * wave + 4 * g;
  W wb[2];
  load_unit(wb[0], t.rider_images[0], wave, lane);
  load_unit(wb[1], t.rider_images[1], wave, lane);
  const f32x4 add = gld4(t.rider_add + grow * D + c_out);
  const bool ok = *(const TBX_GLOBAL uint8_t*)(t.rider_valid + grow) != 0;
  {
    const int r = tid >> 5, c4 = tid & 31;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t.rider_pose3 != nullptr) {  // the rows' pose embeddings (tbx_common.h pose_emb_write: the stand-alone kernel's values)
      if (r < nv) {
        const TBX_GLOBAL float* p3 = (const TBX_GLOBAL float*)t.rider_pose3 + (row0 + r) * 3;
        tbx::pose_emb_write(X + r * XLD, D, p3[0], p3[1], p3[2], t.rider_freqs_xy, t.rider_freqs_yaw, c4, 32);
      }
      __syncthreads();
      if (r < nv) v = *(const f32x4*)(X + r * XLD + c4 * 4);
    } else if (r < nv) {
      v = gld4(t.rider_in + (row0 + r) * D + c4 * 4);
    }
    planes_write4<PL>(Pa, r, c4 * 4, v);
  }
  __syncthreads();
  {
    Acc acc;
    acc.zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, wb[0].hi[s], wb[0].lo[s], Pa + aoff, s);
    planes_write4<PL>(Pb, j, c_out, add + (acc.sum() + wb[0].bias));
    load_unit(wb[0], t.rider_images[2], wave, lane);
  }
  __syncthreads();
  {
    Acc acc;
    acc.zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, wb[1].hi[s], wb[1].lo[s], Pb + aoff, s);
    planes_write4<PL>(Pa, j, c_out, relu4(acc.sum() + wb[1].bias));
    load_unit(wb[1], t.rider_images[3], wave, lane);
  }
  __syncthreads();
  {
    Acc acc;
    acc.zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, wb[0].hi[s], wb[0].lo[s], Pa + aoff, s);
    planes_write4<PL>(Pb, j, c_out, relu4(acc.sum() + wb[0].bias));
  }
  __syncthreads();
  {
    Acc acc;
    acc.zero();
#pragma unroll
    for (int s = 0; s < 4; ++s) mfma_step<PLANE>(acc, wb[1].hi[s], wb[1].lo[s], Pb + aoff, s);
    f32x4 v = relu4(acc.sum() + wb[1].bias);
    if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (row_ok) gst4(t.rider_out + grow * D + c_out, v);
  }
}


}  // namespace tbx_tile
