// How fast can ONE workgroup (512 threads) pull an L2-resident weight image? Loads of 16 B per lane, N KiB in flight per step.
// usage: l2_stream_probe   (prints cycles and GB/s for several in-flight depths and grid sizes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DEPTH>  // float4 loads per thread in flight per step
__global__ __launch_bounds__(512) void probe(const float4* __restrict__ img, int steps, float4* out, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
    float4 r[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) r[i] = img[(s * DEPTH + i) * 512 + tid];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) acc.x += r[i].x, acc.y += r[i].y, acc.z += r[i].z, acc.w += r[i].w;
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 512 + tid] = acc;
}

// the same stream by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a 64 KiB LDS slot, DEPTH instructions per
// wave per step, then "s_waitcnt vmcnt(0); s_barrier" - the live-row chain's weight path
template <int DEPTH>
__global__ __launch_bounds__(512) void probe_dma(const float4* __restrict__ img, int steps, float4* out, unsigned long long* cyc) {
  extern __shared__ float4 slot[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)slot);
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      const float4* src = img + ((s * DEPTH + i) * 8 + wave) * 64 + lane;
      const uint32_t dst = lds0 + (uint32_t)((i % 8) * 8 + wave) * 1024u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 512 + tid] = slot[tid];
}
template <int DEPTH>
void run_dma(const float4* img, int total_f4_per_thread, int grid, float4* out, unsigned long long* cyc) {
  const int steps = total_f4_per_thread / DEPTH;
  CK(hipFuncSetAttribute((const void*)probe_dma<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_dma<DEPTH>, dim3(grid), dim3(512), 65536, 0, img, steps, out, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long c;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double bytes = (double)total_f4_per_thread * 512 * 16;
  printf("DMA  grid %4d depth %2d (%3d KiB per step per WG): %8llu ticks for %4.0f KiB -> %.1f B/tick\n", grid, DEPTH, DEPTH * 8, c, bytes / 1024, bytes / c);
}

// linear_gemv in isolation: two 66 KiB LDS slots; per chunk (33 rows x 2 KiB): [wait + barrier] [DMA of the next chunk into the other
// slot, pieces dealt to the 8 waves] [threads 0..127: bias + 128-long k-ordered fma chain over the slot's 32 weight rows and an
// activation row in LDS]. COMPUTE = false: the same without the chain.
template <bool COMPUTE, int UNR>
__global__ __launch_bounds__(512) void probe_gemv(const float* __restrict__ img, int chunks, float* out, unsigned long long* cyc) {
  extern __shared__ float lds_f[];
  float* slots = lds_f;              // 2 x 33 x 512 floats
  float* x = lds_f + 2 * 33 * 512;   // 128 activations
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 128) x[tid] = 0.001f * tid;
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)slots);
  auto dma = [&](int chunk, int slot) {
    const float* base = img + (size_t)chunk * 33 * 512;
    const int w0 = (UNR >= 100) ? 2 : 0;  // UNR >= 100: the two multiplying waves issue no DMA (their lgkmcnt waits would include it?)
    if (wave < w0) return;
    for (int p = wave - w0; p < 66; p += 8 - w0) {
      const float* src = base + p * 256 + lane * 4;
      const uint32_t dst = lds0 + (uint32_t)slot * 33u * 2048u + (uint32_t)p * 1024u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };
  __syncthreads();
  const unsigned long long t0 = clock64();
  dma(0, 0);
  float acc = 0.f;
  for (int i = 0; i < chunks; ++i) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (i + 1 < chunks) dma(i + 1, (i + 1) & 1);
    if (COMPUTE && tid < 128) {
      const float* w = slots + (i & 1) * 33 * 512 + tid * 4;
      float a = w[0];
      const float4* x4 = (const float4*)x;
      const float4* w4 = (const float4*)(w + 512);
      if (UNR == 399) {  // one float4 of activations per lane per k-block (lane & 3 picks which), the other three through DPP quad broadcasts
#define QB(V, G) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, V), (G) * 0x55, 0xf, 0xf, true))
#define RDQ(XQ, W, KB) XQ = x4[(KB) * 4 + (lane & 3)]; W[0] = w4[((KB) * 4) * 128], W[1] = w4[((KB) * 4 + 1) * 128], W[2] = w4[((KB) * 4 + 2) * 128], W[3] = w4[((KB) * 4 + 3) * 128]
#define FD(XC, WC, G) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #G "," #G "," #G "," #G "] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(XC), "v"(WC))
#define FMQ(XQ, W)                                                                     \
  FD(XQ.x, W[0].x, 0); FD(XQ.x, W[0].y, 1); FD(XQ.x, W[0].z, 2); FD(XQ.x, W[0].w, 3);      \
  FD(XQ.y, W[1].x, 0); FD(XQ.y, W[1].y, 1); FD(XQ.y, W[1].z, 2); FD(XQ.y, W[1].w, 3);      \
  FD(XQ.z, W[2].x, 0); FD(XQ.z, W[2].y, 1); FD(XQ.z, W[2].z, 2); FD(XQ.z, W[2].w, 3);      \
  FD(XQ.w, W[3].x, 0); FD(XQ.w, W[3].y, 1); FD(XQ.w, W[3].z, 2); FD(XQ.w, W[3].w, 3)
        float4 qa, qb, wa[4], wb[4];
        RDQ(qa, wa, 0);
#pragma unroll
        for (int kb = 0; kb < 8; kb += 2) {
          RDQ(qb, wb, kb + 1);
          FMQ(qa, wa);
          if (kb + 2 < 8) { RDQ(qa, wa, kb + 2); }
          FMQ(qb, wb);
        }
#undef QB
#undef RDQ
#undef FMQ
      } else if (UNR == 299) {  // activations as SGPR operands: lane l holds x[l] and x[64 + l], v_readlane per fma; weights pipelined one k-block ahead
        const float xv0 = x[lane], xv1 = x[64 + lane];
        const float4* w4q = w4;
#define XK(K) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (K) < 64 ? xv0 : xv1), (K) & 63))
#define RDW(W, KB) W[0] = w4q[((KB) * 4) * 128], W[1] = w4q[((KB) * 4 + 1) * 128], W[2] = w4q[((KB) * 4 + 2) * 128], W[3] = w4q[((KB) * 4 + 3) * 128]
#define FMW(W, KB)                                                                                                                   \
  a = __builtin_fmaf(XK((KB) * 16 + 0), W[0].x, a); a = __builtin_fmaf(XK((KB) * 16 + 4), W[0].y, a); a = __builtin_fmaf(XK((KB) * 16 + 8), W[0].z, a); a = __builtin_fmaf(XK((KB) * 16 + 12), W[0].w, a); \
  a = __builtin_fmaf(XK((KB) * 16 + 1), W[1].x, a); a = __builtin_fmaf(XK((KB) * 16 + 5), W[1].y, a); a = __builtin_fmaf(XK((KB) * 16 + 9), W[1].z, a); a = __builtin_fmaf(XK((KB) * 16 + 13), W[1].w, a); \
  a = __builtin_fmaf(XK((KB) * 16 + 2), W[2].x, a); a = __builtin_fmaf(XK((KB) * 16 + 6), W[2].y, a); a = __builtin_fmaf(XK((KB) * 16 + 10), W[2].z, a); a = __builtin_fmaf(XK((KB) * 16 + 14), W[2].w, a); \
  a = __builtin_fmaf(XK((KB) * 16 + 3), W[3].x, a); a = __builtin_fmaf(XK((KB) * 16 + 7), W[3].y, a); a = __builtin_fmaf(XK((KB) * 16 + 11), W[3].z, a); a = __builtin_fmaf(XK((KB) * 16 + 15), W[3].w, a)
        float4 wa[4], wb[4];
        RDW(wa, 0);
#pragma unroll
        for (int kb = 0; kb < 8; kb += 2) {
          RDW(wb, kb + 1);
          FMW(wa, kb);
          if (kb + 2 < 8) { RDW(wa, kb + 2); }
          FMW(wb, kb + 1);
        }
#undef XK
#undef RDW
#undef FMW
      } else if (UNR == 99 || UNR == 199) {  // software pipeline: the 8 operand reads of k-block kb + 1 are issued before the 16 fmas of k-block kb
#define RD(X, W, KB)                                                                                                   \
  X[0] = x4[(KB) * 4], X[1] = x4[(KB) * 4 + 1], X[2] = x4[(KB) * 4 + 2], X[3] = x4[(KB) * 4 + 3];                           \
  W[0] = w4[((KB) * 4) * 128], W[1] = w4[((KB) * 4 + 1) * 128], W[2] = w4[((KB) * 4 + 2) * 128], W[3] = w4[((KB) * 4 + 3) * 128]
#define FM(X, W)                                                                                                       \
  a = __builtin_fmaf(X[0].x, W[0].x, a); a = __builtin_fmaf(X[1].x, W[0].y, a); a = __builtin_fmaf(X[2].x, W[0].z, a); a = __builtin_fmaf(X[3].x, W[0].w, a); \
  a = __builtin_fmaf(X[0].y, W[1].x, a); a = __builtin_fmaf(X[1].y, W[1].y, a); a = __builtin_fmaf(X[2].y, W[1].z, a); a = __builtin_fmaf(X[3].y, W[1].w, a); \
  a = __builtin_fmaf(X[0].z, W[2].x, a); a = __builtin_fmaf(X[1].z, W[2].y, a); a = __builtin_fmaf(X[2].z, W[2].z, a); a = __builtin_fmaf(X[3].z, W[2].w, a); \
  a = __builtin_fmaf(X[0].w, W[3].x, a); a = __builtin_fmaf(X[1].w, W[3].y, a); a = __builtin_fmaf(X[2].w, W[3].z, a); a = __builtin_fmaf(X[3].w, W[3].w, a)
        float4 xa[4], wa[4], xb[4], wb[4];
        RD(xa, wa, 0);
#pragma unroll
        for (int kb = 0; kb < 8; kb += 2) {
          RD(xb, wb, kb + 1);
          FM(xa, wa);
          if (kb + 2 < 8) { RD(xa, wa, kb + 2); }
          FM(xb, wb);
        }
#undef RD
#undef FM
      } else {
#pragma unroll (UNR % 100)
      for (int kb = 0; kb < 8; ++kb) {
        const float4 x0 = x4[kb * 4], x1 = x4[kb * 4 + 1], x2 = x4[kb * 4 + 2], x3 = x4[kb * 4 + 3];
        const float4 w0 = w4[(kb * 4) * 128], w1 = w4[(kb * 4 + 1) * 128], w2 = w4[(kb * 4 + 2) * 128], w3 = w4[(kb * 4 + 3) * 128];
        a = __builtin_fmaf(x0.x, w0.x, a); a = __builtin_fmaf(x1.x, w0.y, a); a = __builtin_fmaf(x2.x, w0.z, a); a = __builtin_fmaf(x3.x, w0.w, a);
        a = __builtin_fmaf(x0.y, w1.x, a); a = __builtin_fmaf(x1.y, w1.y, a); a = __builtin_fmaf(x2.y, w1.z, a); a = __builtin_fmaf(x3.y, w1.w, a);
        a = __builtin_fmaf(x0.z, w2.x, a); a = __builtin_fmaf(x1.z, w2.y, a); a = __builtin_fmaf(x2.z, w2.z, a); a = __builtin_fmaf(x3.z, w2.w, a);
        a = __builtin_fmaf(x0.w, w3.x, a); a = __builtin_fmaf(x1.w, w3.y, a); a = __builtin_fmaf(x2.w, w3.z, a); a = __builtin_fmaf(x3.w, w3.w, a);
      }
      }
      acc += a;
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 512 + tid] = acc;
}
template <bool COMPUTE, int UNR>
void run_gemv(const float4* img, int grid, float4* out, unsigned long long* cyc, bool cold) {
  const int chunks = 15;  // ~1 MiB
  const size_t lds = (2 * 33 * 512 + 128) * 4;
  CK(hipFuncSetAttribute((const void*)probe_gemv<COMPUTE, UNR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  unsigned long long sum = 0;
  int n = 0;
  for (int rep = 0; rep < (cold ? 100 : 4); ++rep) {
    const float* base = (const float*)img + (cold ? (size_t)(rep % 100) * (1 << 18) : 0);  // cold: a different MiB every launch
    hipLaunchKernelGGL((probe_gemv<COMPUTE, UNR>), dim3(grid), dim3(512), lds, 0, base, chunks, (float*)out, cyc);
    if (rep >= (cold ? 96 : 1)) { unsigned long long c; CK(hipDeviceSynchronize()); CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); sum += c; ++n; }
  }
  CK(hipDeviceSynchronize());
  printf("GEMV unroll %d grid %4d compute %d %s: %8llu ticks for %d chunks of 66 KiB -> %.0f ticks per chunk, %.1f B/tick\n", UNR, grid, (int)COMPUTE, cold ? "cold" : "warm",
         sum / n, chunks, (double)(sum / n) / chunks, 15.0 * 33 * 2048 / (double)(sum / n));
}

template <int DEPTH>
void run(const float4* img, int total_f4_per_thread, int grid, float4* out, unsigned long long* cyc) {
  const int steps = total_f4_per_thread / DEPTH;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<DEPTH>, dim3(grid), dim3(512), 0, 0, img, steps, out, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long c;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double bytes = (double)total_f4_per_thread * 512 * 16;
  // clock64 = s_memtime: 100 MHz on gfx950? print both interpretations
  printf("grid %4d depth %2d (%3d KiB in flight per WG): %8llu ticks for %4.0f KiB -> %.1f B/tick\n", grid, DEPTH, DEPTH * 8, c, bytes / 1024, bytes / c);
}

// cold variant: every launch streams a different 1 MiB region of a 128 MiB buffer (L2-cold, Infinity-Cache-warm after one sweep);
// TOUCH: every thread first touches one dword per 128-byte line of the whole region (all misses in flight at once), then streams
template <int DEPTH, bool TOUCH>
__global__ __launch_bounds__(512) void probe_cold(const float4* __restrict__ img, int steps, float4* out, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  __syncthreads();
  const unsigned long long t0 = clock64();
  if (TOUCH) {
    const float* f = (const float*)img;
    float t = 0.f;
    const int lines = steps * DEPTH * 512 * 16 / 128;
    for (int l = tid; l < lines; l += 512) t += f[l * 32];
    acc.x += t;
  }
  const unsigned long long tm = clock64();
  for (int s = 0; s < steps; ++s) {
    float4 r[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) r[i] = img[(s * DEPTH + i) * 512 + tid];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) acc.x += r[i].x, acc.y += r[i].y, acc.z += r[i].z, acc.w += r[i].w;
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = tm - t0;
  out[blockIdx.x * 512 + tid] = acc;
}

template <int DEPTH, bool TOUCH>
void run_cold(const float4* img, int total_f4_per_thread, int grid, float4* out, unsigned long long* cyc) {
  const int steps = total_f4_per_thread / DEPTH;
  const size_t region = (size_t)total_f4_per_thread * 512;  // float4s per region
  unsigned long long c[2] = {0, 0}, sum = 0, sumt = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int reg = 0; reg < 128; ++reg) {  // 128 regions of 1 MiB
      hipLaunchKernelGGL((probe_cold<DEPTH, TOUCH>), dim3(grid), dim3(512), 0, 0, img + reg * region, steps, out, cyc);
      if (pass == 1 && reg >= 120) { CK(hipDeviceSynchronize()); CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost)); sum += c[0]; sumt += c[1]; }
    }
  CK(hipDeviceSynchronize());
  const double bytes = (double)total_f4_per_thread * 512 * 16;
  printf("COLD grid %4d depth %2d touch %d: %8llu ticks (touch phase %6llu) for %4.0f KiB -> %.1f B/tick\n", grid, DEPTH, (int)TOUCH, sum / 8, sumt / 8, bytes / 1024, bytes / (sum / 8.0));
}

// dependent v_fma latency: one wave (or all 8), N dependent fmas on register operands
__global__ __launch_bounds__(512) void probe_fma(float* out, unsigned long long* cyc, int active_threads) {
  const int tid = threadIdx.x;
  float a = tid * 1e-3f, x = 1.0001f, w = 0.5f + tid * 1e-6f;
  __syncthreads();
  const unsigned long long t0 = clock64();
  if (tid < active_threads) {
#pragma unroll
    for (int i = 0; i < 1024; ++i) a = __builtin_fmaf(x, w, a);
  }
  const unsigned long long t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 512 + tid] = a;
}

int main() {
  const int total = 128;  // float4 per thread = 1 MiB per WG
  float4* img; float4* out; unsigned long long* cyc;
  CK(hipMalloc(&img, (size_t)total * 512 * 16 * 128));
  CK(hipMemset(img, 0, (size_t)total * 512 * 16 * 128));
  CK(hipMalloc(&out, 1024 * 512 * 16));
  CK(hipMalloc(&cyc, 16));
  for (int act : {64, 128, 512}) {
    hipLaunchKernelGGL(probe_fma, dim3(64), dim3(512), 0, 0, (float*)out, cyc, act);
    CK(hipDeviceSynchronize());
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("FMA chain: %d active threads per WG: %llu ticks for 1024 dependent v_fma -> %.1f ticks each\n", act, c, c / 1024.0);
  }
  for (int grid : {64}) {
    run_gemv<false, 1>(img, grid, out, cyc, false);
    run_gemv<true, 1>(img, grid, out, cyc, false);
    run_gemv<true, 2>(img, grid, out, cyc, false);
    run_gemv<true, 4>(img, grid, out, cyc, false);
    run_gemv<true, 8>(img, grid, out, cyc, false);
    run_gemv<true, 99>(img, grid, out, cyc, false);
    run_gemv<false, 102>(img, grid, out, cyc, false);
    run_gemv<true, 102>(img, grid, out, cyc, false);
    run_gemv<true, 199>(img, grid, out, cyc, false);
    run_gemv<true, 199>(img, grid, out, cyc, true);
    run_gemv<true, 399>(img, grid, out, cyc, false);
    run_gemv<true, 299>(img, grid, out, cyc, false);
    run_gemv<true, 299>(img, grid, out, cyc, true);
  }
  for (int grid : {1, 64}) {
    run_dma<1>(img, total, grid, out, cyc);
    run_dma<4>(img, total, grid, out, cyc);
    run_dma<8>(img, total, grid, out, cyc);
  }
  for (int grid : {1, 64, 192}) {
    run_cold<4, false>(img, total, grid, out, cyc);
    run_cold<8, false>(img, total, grid, out, cyc);
    run_cold<32, false>(img, total, grid, out, cyc);
    run_cold<8, true>(img, total, grid, out, cyc);
  }
  for (int grid : {1, 8, 64, 192}) {
    run<1>(img, total, grid, out, cyc);
    run<4>(img, total, grid, out, cyc);
    run<8>(img, total, grid, out, cyc);
    run<16>(img, total, grid, out, cyc);
    run<32>(img, total, grid, out, cyc);
  }
  return 0;
}
