// Shared device helpers (wave64 reductions, error strings). gfx950 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <atomic>

namespace tbx {

// Host side: hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel, and launches come from several
// host threads (a training step is captured on a thread of its own). One of these per launch site: `set` runs once per device
// (a bit per device ordinal; two threads racing both set the same value - idempotent), failures are reported every time.
struct PerDeviceOnce {
  std::atomic<uint64_t> done{0};
  template <class F>
  bool operator()(F&& set) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set();
    const uint64_t bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (!set()) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
  }
};

// Cross-lane exchange without the LDS crossbar: DPP modifiers inside a 16-lane row (they fuse into the consuming
// v_add / v_max) and the gfx950 v_permlane{16,32}_swap for the two cross-row steps. Each helper returns exactly what the
// __shfl_xor butterfly it replaces returns (same operands per add, commuted at most), so results are bit-identical.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <- lane 7-i of its 8-lane half row (== xor 4 once quads are uniform)
constexpr int DPP_MIRROR = 0x140;      // lane i <- lane 15-i of its row (== xor 8 once 8-lane groups are uniform)
constexpr int DPP_XOR8 = 0x128;        // row_ror:8

// (v[l], v[l ^ 16]) and (v[l], v[l ^ 32]) pairs, one of them being the lane's own value. v_permlane16_swap exchanges the
// odd rows of its first register with the even rows of its second in place, v_permlane32_swap the upper half of the
// first with the lower half of the second; fed two copies of v they leave (rows 0,0,2,2 | 1,1,3,3) and (lo,lo | hi,hi).
// Inline asm with two read-write operands: the clang 22 builtin folds both results into one register. The s_nops cover
// the VALU-write -> permlane-swap and permlane-swap -> VALU-read wait states the assembler does not insert in asm.
__device__ __forceinline__ void swap16(float v, float* a, float* b) {
  float x = v, y = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  *a = x;
  *b = y;
}
__device__ __forceinline__ void swap32(float v, float* a, float* b) {
  float x = v, y = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  *a = x;
  *b = y;
}

// the same two instructions on two DIFFERENT registers: afterwards x = (x rows 0, y rows 0, x rows 2, y rows 2), y = (x rows 1, y rows 1,
// x rows 3, y rows 3) / x = (x lower half | y lower half), y = (x upper half | y upper half)
__device__ __forceinline__ void pswap16(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void pswap32(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}

// slot_sum / slot_max of FOUR values at once: the result of value q lands in the lanes of quarter q (lanes 16 q .. 16 q + 15; lanes l
// and l ^ 8 hold the same number). Same butterfly as four slot_sum calls - (l + l^8), then + (^16), then + (^32), the same operands
// per add - so the sums are bit-identical; 10 instructions instead of ~50: the two cross-row steps exchange rows of two DIFFERENT
// values (both halves of a swap are useful) instead of a value with a copy of itself.
__device__ __forceinline__ float slot_sum4(float a0, float a1, float a2, float a3) {
  a0 += dpp<DPP_XOR8>(a0);
  a1 += dpp<DPP_XOR8>(a1);
  a2 += dpp<DPP_XOR8>(a2);
  a3 += dpp<DPP_XOR8>(a3);
  pswap16(a0, a1);
  float u01 = a0 + a1;  // rows: a0 (r0 + r1) | a1 (r0 + r1) | a0 (r2 + r3) | a1 (r2 + r3)
  pswap16(a2, a3);
  float u23 = a2 + a3;
  pswap32(u01, u23);    // (a0', a1' | a2', a3') and (a0'', a1'' | a2'', a3'')
  return u01 + u23;
}
__device__ __forceinline__ float slot_max4(float a0, float a1, float a2, float a3) {
  a0 = fmaxf(a0, dpp<DPP_XOR8>(a0));
  a1 = fmaxf(a1, dpp<DPP_XOR8>(a1));
  a2 = fmaxf(a2, dpp<DPP_XOR8>(a2));
  a3 = fmaxf(a3, dpp<DPP_XOR8>(a3));
  pswap16(a0, a1);
  float u01 = fmaxf(a0, a1);
  pswap16(a2, a3);
  float u23 = fmaxf(a2, a3);
  pswap32(u01, u23);
  return fmaxf(u01, u23);
}

// sum / max over lanes l ^ {8, 16, 32} (the 8 lanes that share l & 7)
__device__ __forceinline__ float slot_sum(float v) {
  float a, b;
  v += dpp<DPP_XOR8>(v);
  swap16(v, &a, &b);
  v = a + b;
  swap32(v, &a, &b);
  return a + b;
}
__device__ __forceinline__ float slot_max(float v) {
  float a, b;
  v = fmaxf(v, dpp<DPP_XOR8>(v));
  swap16(v, &a, &b);
  v = fmaxf(a, b);
  swap32(v, &a, &b);
  return fmaxf(a, b);
}

// sum over aligned groups of 8 lanes
__device__ __forceinline__ float group8_sum(float v) {
  v += dpp<DPP_XOR1>(v);
  v += dpp<DPP_XOR2>(v);
  v += dpp<DPP_HALF_MIRROR>(v);
  return v;
}

// Workgroup b of a 1-D grid runs on XCD b % 8 (MI300 / MI355X dispatch order). xcd_block(b, grid) = a virtual block index under
// which XCD x owns the CONTIGUOUS range [x * per + min(x, rem), ...) of the grid's blocks (per = grid / 8, rem = grid % 8): kernels
// whose consecutive blocks read the same tables (the rows of one rollout / scene / time-batched scene) then keep a table in ONE
// XCD's L2 instead of all eight.
__device__ __forceinline__ int xcd_block(const int b, const int grid) {
  const int xcd = b & 7, slot = b >> 3;
  const int per = grid >> 3, rem = grid & 7;
  return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = group8_sum(v);
  v += dpp<DPP_MIRROR>(v);
  float a, b;
  swap16(v, &a, &b);
  v = a + b;
  swap32(v, &a, &b);
  return a + b;
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp<DPP_XOR1>(v));
  v = fmaxf(v, dpp<DPP_XOR2>(v));
  v = fmaxf(v, dpp<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp<DPP_MIRROR>(v));
  float a, b;
  swap16(v, &a, &b);
  v = fmaxf(a, b);
  swap32(v, &a, &b);
  return fmaxf(a, b);
}

// 128-d (or 64-d) pe_xy_yaw channel c of pose (x, y, yaw): utils/pose_emb.py:50-55, utils/positional_emb.py:25,53.
// freqs_* are the reference's repeat-interleaved buffers; cos uses the even entries, sin the odd ones.
__device__ __forceinline__ float pose_emb_channel(int c, int pe_dim, float x, float y, float yaw,
                                                  const float* __restrict__ fxy, const float* __restrict__ fyaw) {
  const int nxy = pe_dim >> 3;  // frequencies per cos/sin block of x or y
  const int nyw = pe_dim >> 2;  // yaw harmonics
  if (c < nxy) return cosf(x * fxy[2 * c]);
  c -= nxy;
  if (c < nxy) return sinf(x * fxy[2 * c + 1]);
  c -= nxy;
  if (c < nxy) return cosf(y * fxy[2 * c]);
  c -= nxy;
  if (c < nxy) return sinf(y * fxy[2 * c + 1]);
  c -= nxy;
  if (c < nyw) return cosf(yaw * fyaw[2 * c]);
  c -= nyw;
  return sinf(yaw * fyaw[2 * c + 1]);
}

// Writes the pe_dim-d pe_xy_yaw embedding of (x, y, yaw) to e[0:pe_dim], cooperatively over `nl` lanes (lane id `l`).
// The pe_dim channels are pe_dim/2 (cos, sin) pairs of the SAME argument (x f_i | y f_i | k yaw), so each lane evaluates
// one sincosf per argument and writes two channels: no divergent trig branches. The reference's `freqs` buffers are
// repeat-interleaved ([f0,f0,f1,f1,..], positional_emb.py:13,41); the even entry is used for both channels of a pair.
__device__ __forceinline__ void pose_emb_write(float* __restrict__ e, int pe_dim, float x, float y, float yaw,
                                               const float* __restrict__ fxy, const float* __restrict__ fyaw, int l, int nl) {
  const int nxy = pe_dim >> 3, nyw = pe_dim >> 2;
  for (int a = l; a < 2 * nxy + nyw; a += nl) {
    float arg;
    int cc, sc;
    if (a < nxy) {
      arg = x * fxy[2 * a];
      cc = a;
      sc = a + nxy;
    } else if (a < 2 * nxy) {
      const int i = a - nxy;
      arg = y * fxy[2 * i];
      cc = 2 * nxy + i;
      sc = 3 * nxy + i;
    } else {
      const int i = a - 2 * nxy;
      arg = yaw * fyaw[2 * i];
      cc = 4 * nxy + i;
      sc = 4 * nxy + nyw + i;
    }
    float sn, cs;
    sincosf(arg, &sn, &cs);
    e[cc] = cs;
    e[sc] = sn;
  }
}

}  // namespace tbx
