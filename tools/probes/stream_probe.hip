// Streaming-rate probe (round 6): what a read-X / write-Y pass over [rows, 128] fp32 rows achieves on this device as a function of the
// access width per lane, the cache policy (default vs non-temporal) and the grid - the pattern of ln_fwd / the glue kernels /
// tall_linear's row traffic.   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int W, bool NT_LD, bool NT_ST, int UNROLL>
__global__ __launch_bounds__(1024) void copy_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n_vec) {
  // n_vec vectors of W floats; thread-strided (every wave instruction touches 64 * W * 4 contiguous bytes)
  typedef float vec __attribute__((ext_vector_type(W)));
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  const vec* xv = (const vec*)x;
  vec* yv = (vec*)y;
  for (int64_t i = tid; i < n_vec; i += nt * UNROLL) {
    vec v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t j = i + u * nt;
      if (j < n_vec) v[u] = NT_LD ? __builtin_nontemporal_load(xv + j) : xv[j];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t j = i + u * nt;
      if (j < n_vec) {
        vec o = v[u] * 1.0001f;
        if (NT_ST) __builtin_nontemporal_store(o, yv + j);
        else yv[j] = o;
      }
    }
  }
}

// read-only (sum) and write-only forms: where the asymmetry is
template <int W, bool NT>
__global__ __launch_bounds__(1024) void read_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n_vec) {
  typedef float vec __attribute__((ext_vector_type(W)));
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  const vec* xv = (const vec*)x;
  float acc = 0.f;
  for (int64_t i = tid; i < n_vec; i += nt * 4) {
    vec v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + u * nt;
      v[u] = j < n_vec ? (NT ? __builtin_nontemporal_load(xv + j) : xv[j]) : (vec)(0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u][0] + v[u][W - 1];
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int W, bool NT>
__global__ __launch_bounds__(1024) void write_kernel(float* __restrict__ y, int64_t n_vec) {
  typedef float vec __attribute__((ext_vector_type(W)));
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  vec* yv = (vec*)y;
  for (int64_t i = tid; i < n_vec; i += nt) {
    vec o = (vec)((float)i);
    if (NT) __builtin_nontemporal_store(o, yv + i);
    else yv[i] = o;
  }
}

template <class F>
static float time_us(F&& launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const int64_t rows_list[] = {184320, 2027520};
  for (int64_t rows : rows_list) {
    const int64_t n = rows * 128;
    float *x, *y;
    hipMalloc(&x, n * 4), hipMalloc(&y, n * 4);
    hipMemset(x, 0, n * 4);
    printf("rows %ld x 128 fp32 (%.1f MB each way)\n", (long)rows, n * 4 / 1e6);
    const double by2 = 2.0 * n * 4, by1 = 1.0 * n * 4;
    for (int grid : {256, 512, 1024, 2048}) {
      for (int threads : {256, 1024}) {
#define RUN(NAME, KERNEL, NV, BYTES)                                                                              \
  {                                                                                                                \
    const float us = time_us([&] { hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(threads), 0, 0, x, y, (int64_t)(NV)); }, 20); \
    printf("  grid %5d x %4d  %-34s %8.1f us  %6.0f GB/s  %.3f of 8 TB/s\n", grid, threads, NAME, us, BYTES / us / 1e3, BYTES / us / 1e3 / 8000.0); \
  }
        RUN("copy 8B/lane", (copy_kernel<2, false, false, 4>), n / 2, by2)
        RUN("copy 16B/lane", (copy_kernel<4, false, false, 4>), n / 4, by2)
        RUN("copy 16B/lane nt-load", (copy_kernel<4, true, false, 4>), n / 4, by2)
        RUN("copy 16B/lane nt-store", (copy_kernel<4, false, true, 4>), n / 4, by2)
        RUN("copy 16B/lane nt both", (copy_kernel<4, true, true, 4>), n / 4, by2)
        RUN("copy 16B/lane nt both unroll 8", (copy_kernel<4, true, true, 8>), n / 4, by2)
        RUN("copy 8B/lane nt both", (copy_kernel<2, true, true, 4>), n / 2, by2)
#undef RUN
      }
    }
    for (int grid : {512, 2048}) {
      {
        const float us = time_us([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(grid), dim3(1024), 0, 0, x, y, n / 4); }, 20);
        printf("  grid %5d  read-only 16B/lane            %8.1f us  %6.0f GB/s  %.3f\n", grid, us, by1 / us / 1e3, by1 / us / 1e3 / 8000.0);
      }
      {
        const float us = time_us([&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(grid), dim3(1024), 0, 0, x, y, n / 4); }, 20);
        printf("  grid %5d  read-only 16B/lane nt         %8.1f us  %6.0f GB/s  %.3f\n", grid, us, by1 / us / 1e3, by1 / us / 1e3 / 8000.0);
      }
      {
        const float us = time_us([&] { hipLaunchKernelGGL((write_kernel<4, false>), dim3(grid), dim3(1024), 0, 0, y, n / 4); }, 20);
        printf("  grid %5d  write-only 16B/lane           %8.1f us  %6.0f GB/s  %.3f\n", grid, us, by1 / us / 1e3, by1 / us / 1e3 / 8000.0);
      }
      {
        const float us = time_us([&] { hipLaunchKernelGGL((write_kernel<4, true>), dim3(grid), dim3(1024), 0, 0, y, n / 4); }, 20);
        printf("  grid %5d  write-only 16B/lane nt        %8.1f us  %6.0f GB/s  %.3f\n", grid, us, by1 / us / 1e3, by1 / us / 1e3 / 8000.0);
      }
    }
    hipFree(x), hipFree(y);
  }
  return 0;
}
