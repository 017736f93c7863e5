// Microbenchmark: HBM read rate of the access patterns a text-streaming kernel can use.
//   mode 0  linear: a wave reads 64 x 16 B = 1 KiB contiguous per instruction, U instructions (U KiB
//           contiguous) per step, waves own contiguous regions
//   mode 1  lane chunks (filter_dna_kernel today): every lane owns a chunk of `bpl` blocks; one
//           instruction fetches 8 lanes x 16 B = one 128-byte line of each of 8 chunks, 8 instructions
//           = 64 lines scattered bpl*64 bytes apart
//   mode 2  linear, grid-stride interleaved: wave w reads KiB w, w+W, w+2W, ...
// hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip && ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int MODE, int U>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ text, uint64_t n16, uint32_t* out,
                                         uint32_t bpl, uint32_t n_iter) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
  const uint64_t n_waves = (uint64_t)gridDim.x * 4;
  uint32_t acc = 0;
  if (MODE == 0) {
    const uint64_t per_wave = n16 / n_waves;  // 16-byte units
    const uint4* p = text + wave * per_wave + lane;
    for (uint64_t i = 0; i + (uint64_t)U * 64 <= per_wave; i += (uint64_t)U * 64) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + (uint64_t)u * 64];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  } else if (MODE == 2) {
    for (uint64_t i = wave * (uint64_t)U * 64; i + (uint64_t)U * 64 <= n16; i += n_waves * (uint64_t)U * 64) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = text[i + (uint64_t)u * 64 + lane];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  } else {
    // lane chunk c = wave*64 + owner; instruction i covers owners 8i..8i+7, lane%8 = 16-byte slot of
    // the 128-byte line; two blocks (128 B) per step
    const uint64_t chunk0 = wave * 64;
    uint64_t off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) off[i] = (chunk0 + (uint64_t)i * 8 + lane / 8) * (uint64_t)bpl * 4 + (lane % 8);
    for (uint32_t it = 0; it + 2 <= n_iter; it += 2) {
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint64_t o = off[i] + (uint64_t)it * 4;
        v[i] = o < n16 ? text[o] : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int U>
void run(const char* name, const uint4* d, uint64_t n, uint32_t* d_out, int waves_per_cu) {
  const uint32_t grid = 256u * waves_per_cu / 4;
  const uint64_t n16 = n / 16;
  const uint64_t n_chunks = (uint64_t)grid * 256;
  uint32_t bpl = (uint32_t)((n / 64 + n_chunks - 1) / n_chunks);
  bpl += bpl & 1;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, U>), dim3(grid), dim3(256), 0, 0, d, n16, d_out, bpl, bpl);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  printf("%-44s waves/CU=%2d  %.3f ms  %.0f GB/s\n", name, waves_per_cu, best, n / (best * 1e-3) / 1e9);
}

int main() {
  const uint64_t n = 3000000000ull;
  uint4* d; uint32_t* d_out;
  hipMalloc(&d, n + 4096); hipMalloc(&d_out, 64);
  hipMemset(d, 1, n + 4096);
  for (int w : {8, 16, 32}) {
    run<0, 4>("linear, 4 KiB per step, contiguous regions", d, n, d_out, w);
    run<0, 8>("linear, 8 KiB per step, contiguous regions", d, n, d_out, w);
    run<2, 4>("linear, 4 KiB per step, interleaved waves", d, n, d_out, w);
    run<2, 8>("linear, 8 KiB per step, interleaved waves", d, n, d_out, w);
    run<1, 8>("lane chunks, 64 x 128 B lines per step", d, n, d_out, w);
  }
  return 0;
}
