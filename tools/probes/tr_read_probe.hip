// ds_read_b64_tr_b16 semantics probe (gfx950): which LDS halfwords does lane l receive for a given per-lane address pattern?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// LDS halfword i holds the value i. Pattern A: lane l -> byte address l * 8 (a dense [16 rows][16 halfwords] image per 16-lane
// group... in natural order). Pattern B: lane i of a 16-lane group g -> row (g * 4 + (i >> 2)) of a [rows][144-halfword] image,
// halfwords 4 * (i & 3) .. + 4 (the padded row layout the attention kernel uses).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint16_t* out, int pattern) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;
  if (pattern == 0) {
    addr = l * 8;
  } else {
    const int g = l >> 4, i = l & 15;
    addr = ((g * 4 + (i >> 2)) * 144 + 4 * (i & 3)) * 2;
  }
  addr += (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (uint16_t)(v.x & 0xffff);
  out[l * 4 + 1] = (uint16_t)(v.x >> 16);
  out[l * 4 + 2] = (uint16_t)(v.y & 0xffff);
  out[l * 4 + 3] = (uint16_t)(v.y >> 16);
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5u %5u %5u %5u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 1) ? "\n" : "   |   ");
  }
  return 0;
}
