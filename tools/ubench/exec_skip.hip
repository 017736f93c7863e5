// Microbenchmark: does a VALU instruction get cheaper when whole 16-lane quarters of the wavefront are masked off?
// (decides whether sub-wave groups with their own row counters could run the streaming DP divergently for free)
// hipcc --offload-arch=gfx950 -O3 -o exec_skip exec_skip.hip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned long long mask) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = (threadIdx.x + 1) * 2654435761u + i * 0x9E3779B9u;
  if ((mask >> lane) & 1ull) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a[0] = __builtin_amdgcn_bitop3_b32(a[0], a[1], a[2], 0x96); a[2] = __builtin_amdgcn_bitop3_b32(a[2], a[3], a[4], 0x96);
        a[4] = __builtin_amdgcn_bitop3_b32(a[4], a[5], a[6], 0x96); a[6] = __builtin_amdgcn_bitop3_b32(a[6], a[7], a[0], 0x96);
        a[1] = __builtin_amdgcn_bitop3_b32(a[1], a[2], a[3], 0xE8); a[3] = __builtin_amdgcn_bitop3_b32(a[3], a[4], a[5], 0xE8);
        a[5] = __builtin_amdgcn_bitop3_b32(a[5], a[6], a[7], 0xE8); a[7] = __builtin_amdgcn_bitop3_b32(a[7], a[0], a[1], 0xE8);
      }
    }
  }
  unsigned r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const unsigned long long masks[] = {~0ull, 0x0000FFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0xFFFFFFFF00000000ull, 0x000000000000FFFFull,
                                      0x0000FFFF0000FFFFull, 0x00000000FFFF0000ull, 0x1ull, 0x0001000100010001ull, 0x5555555555555555ull};
  for (unsigned long long m : masks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 16, m);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, 4096, m);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("exec mask %016llx (%2d lanes): %8.3f ms\n", m, __builtin_popcountll(m), ms);
  }
  return 0;
}
