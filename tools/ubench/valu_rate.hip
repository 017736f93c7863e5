// Microbenchmark: issue rate of the integer VALU instructions the scan kernel is made of.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
  unsigned long long w[4];
  for (int i = 0; i < 4; ++i) w[i] = ((unsigned long long)a[2*i] << 32) | a[2*i+1];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) { a[0] = (a[0] & a[1]) ; a[2] = a[2] & a[3]; a[4] = a[4] & a[5]; a[6] = a[6] & a[7]; a[1] = a[1] & a[2]; a[3] = a[3] & a[4]; a[5] = a[5] & a[6]; a[7] = a[7] & a[0]; }
      if (OP == 1) { a[0] = __builtin_amdgcn_bitop3_b32(a[0], a[1], a[2], 0x96); a[2] = __builtin_amdgcn_bitop3_b32(a[2], a[3], a[4], 0x96); a[4] = __builtin_amdgcn_bitop3_b32(a[4], a[5], a[6], 0x96); a[6] = __builtin_amdgcn_bitop3_b32(a[6], a[7], a[0], 0x96); a[1] = __builtin_amdgcn_bitop3_b32(a[1], a[2], a[3], 0xE8); a[3] = __builtin_amdgcn_bitop3_b32(a[3], a[4], a[5], 0xE8); a[5] = __builtin_amdgcn_bitop3_b32(a[5], a[6], a[7], 0xE8); a[7] = __builtin_amdgcn_bitop3_b32(a[7], a[0], a[1], 0xE8); }
      if (OP == 2) { a[0] = __builtin_amdgcn_alignbit(a[0], a[1], 31); a[2] = __builtin_amdgcn_alignbit(a[2], a[3], 31); a[4] = __builtin_amdgcn_alignbit(a[4], a[5], 31); a[6] = __builtin_amdgcn_alignbit(a[6], a[7], 31); a[1] = __builtin_amdgcn_alignbit(a[1], a[2], 31); a[3] = __builtin_amdgcn_alignbit(a[3], a[4], 31); a[5] = __builtin_amdgcn_alignbit(a[5], a[6], 31); a[7] = __builtin_amdgcn_alignbit(a[7], a[0], 31); }
      if (OP == 3) { w[0] += w[1]; w[1] += w[2]; w[2] += w[3]; w[3] += w[0]; w[0] += w[2]; w[1] += w[3]; w[2] += w[0]; w[3] += w[1]; }
      if (OP == 4) { a[0] = __builtin_amdgcn_udot4(a[0], a[1], a[2], false); a[2] = __builtin_amdgcn_udot4(a[2], a[3], a[4], false); a[4] = __builtin_amdgcn_udot4(a[4], a[5], a[6], false); a[6] = __builtin_amdgcn_udot4(a[6], a[7], a[0], false); a[1] = __builtin_amdgcn_udot4(a[1], a[2], a[3], false); a[3] = __builtin_amdgcn_udot4(a[3], a[4], a[5], false); a[5] = __builtin_amdgcn_udot4(a[5], a[6], a[7], false); a[7] = __builtin_amdgcn_udot4(a[7], a[0], a[1], false); }
      if (OP == 5) { a[0] = (a[0] << 1) | a[1]; a[2] = (a[2] << 1) | a[3]; a[4] = (a[4] << 1) | a[5]; a[6] = (a[6] << 1) | a[7]; a[1] = (a[1] << 1) | a[2]; a[3] = (a[3] << 1) | a[4]; a[5] = (a[5] << 1) | a[6]; a[7] = (a[7] << 1) | a[0]; }
      if (OP == 6) { a[0] = (a[0] >> 5) & 1u; a[0] += a[1]; a[2] = (a[2] >> 7) & 1u; a[2] += a[3]; a[4] = (a[4] >> 9) & 1u; a[4] += a[5]; a[6] = (a[6] >> 3) & 1u; a[6] += a[7]; }
      if (OP == 7) { a[0] = __popc(a[0]) + a[1]; a[2] = __popc(a[2]) + a[3]; a[4] = __popc(a[4]) + a[5]; a[6] = __popc(a[6]) + a[7]; a[1] = __popc(a[1]) + a[2]; a[3] = __popc(a[3]) + a[4]; a[5] = __popc(a[5]) + a[6]; a[7] = __popc(a[7]) + a[0]; }
      if (OP == 8) { float f0 = __uint_as_float(a[0]), f1 = __uint_as_float(a[1]), f2 = __uint_as_float(a[2]), f3 = __uint_as_float(a[3]); f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f0); f3 = fmaf(f3, f0, f1); f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f0); f3 = fmaf(f3, f0, f1); a[0] = __float_as_uint(f0); a[1] = __float_as_uint(f1); a[2] = __float_as_uint(f2); a[3] = __float_as_uint(f3); }
    }
  }
  unsigned r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i];
  for (int i = 0; i < 4; ++i) r ^= (unsigned)w[i] ^ (unsigned)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> double run(const char* name, int per_iter, unsigned* d_out) {
  const int iters = 4096, grid = 256 * 8;  // 8 blocks of 256 per CU = 8 waves / SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, 16, 1u);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, 1u);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)grid * 4 * iters * 16.0 * per_iter;  // wave-instructions
  double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
  printf("%-28s %8.3f ms  %7.2f G wave-instr/s/SIMD  -> %.2f cycles/instr @2.4GHz (%.1f T lane-ops/s)\n", name, ms,
         per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, winstr * 64 / (ms * 1e-3) / 1e12);
  return ms;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_and_b32 (VOP2)", 8, d);
  run<1>("v_bitop3_b32", 8, d);
  run<2>("v_alignbit_b32", 8, d);
  run<3>("64-bit add", 8, d);
  run<4>("v_dot4_u32_u8", 8, d);
  run<5>("v_lshl_or_b32", 8, d);
  run<6>("v_bfe_u32 + v_add", 8, d);
  run<7>("v_bcnt_u32_b32", 8, d);
  run<8>("v_fma_f32", 8, d);
  return 0;
}
