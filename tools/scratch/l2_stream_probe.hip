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

int main() {
  const int total = 128;  // float4 per thread = 1 MiB per WG
  float4* img; float4* out; unsigned long long* cyc;
  CK(hipMalloc(&img, (size_t)total * 512 * 16 * 128));
  CK(hipMemset(img, 0, (size_t)total * 512 * 16 * 128));
  CK(hipMalloc(&out, 1024 * 512 * 16));
  CK(hipMalloc(&cyc, 16));
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
