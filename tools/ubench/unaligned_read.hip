// Microbenchmark: does a misaligned global_load_dwordx4 stream (the reverse-strand scan would read the
// forward text backwards, 16-byte pieces at n - L - 16 with n arbitrary) cost bandwidth?
// Lane-chunk access pattern of the filter kernels, byte offset `mis` added to every address.
// hipcc --offload-arch=gfx950 -O3 -o unaligned_read unaligned_read.hip && ./unaligned_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));

__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ text, uint64_t n16, uint32_t* out, uint32_t bpl,
                                         uint32_t n_iter, uint32_t mis) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
  uint32_t acc = 0;
  const uint64_t chunk0 = wave * 64;
  uint64_t off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) off[i] = (chunk0 + (uint64_t)i * 8 + lane / 8) * (uint64_t)bpl * 4 + (lane % 8);
  for (uint32_t it = 0; it + 2 <= n_iter; it += 2) {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t o = off[i] + (uint64_t)it * 4;
      v[i] = o + 1 < n16 ? *reinterpret_cast<const u32x4*>(text + o * 16 + mis) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const uint64_t n = 3000000000ull;
  uint8_t* d; uint32_t* o;
  hipMalloc(&d, n + 4096); hipMalloc(&o, 4);
  hipMemset(d, 1, n + 4096);
  const uint32_t wpc = 16;
  const uint64_t lanes = 256ull * wpc * 64 * 2;
  const uint64_t blocks = n / 64;
  uint32_t bpl = (uint32_t)((blocks + lanes - 1) / lanes); bpl += bpl & 1;
  const uint64_t chunks = (blocks + bpl - 1) / bpl;
  const uint32_t grid = (uint32_t)((chunks + 255) / 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (uint32_t mis : {0u, 1u, 3u, 4u, 8u, 13u}) {
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, n / 16, o, bpl, bpl, mis);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("lane chunks, misalignment %2u bytes: %.3f ms  %.0f GB/s\n", mis, best, n / best / 1e6);
  }
  return 0;
}
