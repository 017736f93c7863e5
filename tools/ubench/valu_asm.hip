// Microbenchmark with explicit registers: cycles per wave64 instruction for the integer ops of
// the DP step, and the effect of VGPR bank placement (bank = register index mod 4) on
// three-source instructions.   hipcc --offload-arch=gfx950 -O3 -o valu_asm valu_asm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
// each BODY string must contain exactly 8 independent instructions on v100..v131
#define DEFK(NAME, BODY)                                                               \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters) {                \
    asm volatile("v_mov_b32 v100, 1\n v_mov_b32 v101, 2\n v_mov_b32 v102, 3\n v_mov_b32 v103, 4\n" \
                 "v_mov_b32 v104, 5\n v_mov_b32 v105, 6\n v_mov_b32 v106, 7\n v_mov_b32 v107, 8\n" \
                 "v_mov_b32 v108, 9\n v_mov_b32 v109, 10\n v_mov_b32 v110, 11\n v_mov_b32 v111, 12\n" \
                 "v_mov_b32 v112, 13\n v_mov_b32 v113, 14\n v_mov_b32 v114, 15\n v_mov_b32 v115, 16\n" \
                 "v_mov_b32 v116, 9\n v_mov_b32 v117, 10\n v_mov_b32 v118, 11\n v_mov_b32 v119, 12\n" \
                 "v_mov_b32 v120, 13\n v_mov_b32 v121, 14\n v_mov_b32 v122, 15\n v_mov_b32 v123, 16\n" \
                 ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123"); \
    for (int it = 0; it < iters; ++it) {                                                 \
      asm volatile(R16(BODY) ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","vcc"); \
    }                                                                                    \
    unsigned r;                                                                          \
    asm volatile("v_xor_b32 %0, v100, v104" : "=v"(r));                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
  }
// 2-source VOP2, distinct destinations
DEFK(k_and2, "v_and_b32 v100, v101, v102\n v_or_b32 v104, v105, v106\n v_xor_b32 v108, v109, v110\n v_and_b32 v112, v113, v114\n"
             "v_or_b32 v101, v102, v103\n v_xor_b32 v105, v106, v107\n v_and_b32 v109, v110, v111\n v_or_b32 v113, v114, v115\n")
// bitop3, sources in 3 different banks (101,102,103 -> banks 1,2,3)
DEFK(k_bitop3_ok, "v_bitop3_b32 v100, v101, v102, v103 bitop3:0x96\n v_bitop3_b32 v104, v105, v106, v107 bitop3:0x96\n v_bitop3_b32 v108, v109, v110, v111 bitop3:0x96\n v_bitop3_b32 v112, v113, v114, v115 bitop3:0x96\n"
                  "v_bitop3_b32 v116, v117, v118, v119 bitop3:0xe8\n v_bitop3_b32 v120, v121, v122, v123 bitop3:0xe8\n v_bitop3_b32 v100, v105, v110, v115 bitop3:0xe8\n v_bitop3_b32 v104, v109, v114, v119 bitop3:0xe8\n")
// bitop3, all three sources in the same bank (101,105,109 -> bank 1)
DEFK(k_bitop3_conf, "v_bitop3_b32 v100, v101, v105, v109 bitop3:0x96\n v_bitop3_b32 v104, v102, v106, v110 bitop3:0x96\n v_bitop3_b32 v108, v103, v107, v111 bitop3:0x96\n v_bitop3_b32 v112, v113, v117, v121 bitop3:0x96\n"
                    "v_bitop3_b32 v116, v114, v118, v122 bitop3:0xe8\n v_bitop3_b32 v120, v115, v119, v123 bitop3:0xe8\n v_bitop3_b32 v100, v101, v105, v113 bitop3:0xe8\n v_bitop3_b32 v104, v102, v106, v114 bitop3:0xe8\n")
// bitop3 with only two distinct VGPR sources + one repeated
DEFK(k_bitop3_2src, "v_bitop3_b32 v100, v101, v102, v102 bitop3:0x96\n v_bitop3_b32 v104, v105, v106, v106 bitop3:0x96\n v_bitop3_b32 v108, v109, v110, v110 bitop3:0x96\n v_bitop3_b32 v112, v113, v114, v114 bitop3:0x96\n"
                    "v_bitop3_b32 v116, v117, v118, v118 bitop3:0xe8\n v_bitop3_b32 v120, v121, v122, v122 bitop3:0xe8\n v_bitop3_b32 v100, v105, v110, v110 bitop3:0xe8\n v_bitop3_b32 v104, v109, v114, v114 bitop3:0xe8\n")
DEFK(k_alignbit, "v_alignbit_b32 v100, v101, v102, 31\n v_alignbit_b32 v104, v105, v106, 31\n v_alignbit_b32 v108, v109, v110, 31\n v_alignbit_b32 v112, v113, v114, 31\n"
                 "v_alignbit_b32 v116, v117, v118, 31\n v_alignbit_b32 v120, v121, v122, 31\n v_alignbit_b32 v101, v102, v103, 31\n v_alignbit_b32 v105, v106, v107, 31\n")
DEFK(k_lshl_or, "v_lshl_or_b32 v100, v101, 1, v102\n v_lshl_or_b32 v104, v105, 1, v106\n v_lshl_or_b32 v108, v109, 1, v110\n v_lshl_or_b32 v112, v113, 1, v114\n"
                "v_lshl_or_b32 v116, v117, 1, v118\n v_lshl_or_b32 v120, v121, 1, v122\n v_lshl_or_b32 v101, v102, 1, v103\n v_lshl_or_b32 v105, v106, 1, v107\n")
DEFK(k_add64, "v_lshl_add_u64 v[100:101], v[102:103], 0, v[104:105]\n v_lshl_add_u64 v[106:107], v[108:109], 0, v[110:111]\n v_lshl_add_u64 v[112:113], v[114:115], 0, v[116:117]\n v_lshl_add_u64 v[118:119], v[120:121], 0, v[122:123]\n"
              "v_lshl_add_u64 v[102:103], v[104:105], 0, v[106:107]\n v_lshl_add_u64 v[108:109], v[110:111], 0, v[112:113]\n v_lshl_add_u64 v[114:115], v[116:117], 0, v[118:119]\n v_lshl_add_u64 v[120:121], v[122:123], 0, v[100:101]\n")
// 64-bit add as add_co + addc (counts as 8 instrs = 4 adds)
DEFK(k_addc, "v_add_co_u32 v100, vcc, v101, v102\n v_addc_co_u32 v103, vcc, v104, v105, vcc\n v_add_co_u32 v106, vcc, v107, v108\n v_addc_co_u32 v109, vcc, v110, v111, vcc\n"
             "v_add_co_u32 v112, vcc, v113, v114\n v_addc_co_u32 v115, vcc, v116, v117, vcc\n v_add_co_u32 v118, vcc, v119, v120\n v_addc_co_u32 v121, vcc, v122, v123, vcc\n")
DEFK(k_bfe, "v_bfe_u32 v100, v101, 5, 1\n v_bfe_u32 v104, v105, 6, 1\n v_bfe_u32 v108, v109, 7, 1\n v_bfe_u32 v112, v113, 8, 1\n"
            "v_bfe_u32 v116, v117, 9, 1\n v_bfe_u32 v120, v121, 10, 1\n v_bfe_u32 v102, v103, 11, 1\n v_bfe_u32 v106, v107, 12, 1\n")
DEFK(k_dot4, "v_dot4_u32_u8 v100, v101, v102, v103\n v_dot4_u32_u8 v104, v105, v106, v107\n v_dot4_u32_u8 v108, v109, v110, v111\n v_dot4_u32_u8 v112, v113, v114, v115\n"
             "v_dot4_u32_u8 v116, v117, v118, v119\n v_dot4_u32_u8 v120, v121, v122, v123\n v_dot4_u32_u8 v101, v102, v103, v104\n v_dot4_u32_u8 v105, v106, v107, v108\n")
DEFK(k_dot4_s, "v_dot4_u32_u8 v100, v101, s4, v103\n v_dot4_u32_u8 v104, v105, s4, v107\n v_dot4_u32_u8 v108, v109, s4, v111\n v_dot4_u32_u8 v112, v113, s4, v115\n"
               "v_dot4_u32_u8 v116, v117, s4, v119\n v_dot4_u32_u8 v120, v121, s4, v123\n v_dot4_u32_u8 v101, v102, s4, v104\n v_dot4_u32_u8 v105, v106, s4, v108\n")
DEFK(k_and_or, "v_and_or_b32 v100, v101, v102, v103\n v_and_or_b32 v104, v105, v106, v107\n v_and_or_b32 v108, v109, v110, v111\n v_and_or_b32 v112, v113, v114, v115\n"
               "v_or3_b32 v116, v117, v118, v119\n v_or3_b32 v120, v121, v122, v123\n v_or3_b32 v101, v102, v103, v104\n v_or3_b32 v105, v106, v107, v108\n")
DEFK(k_bcnt, "v_bcnt_u32_b32 v100, v101, v102\n v_bcnt_u32_b32 v104, v105, v106\n v_bcnt_u32_b32 v108, v109, v110\n v_bcnt_u32_b32 v112, v113, v114\n"
             "v_bcnt_u32_b32 v116, v117, v118\n v_bcnt_u32_b32 v120, v121, v122\n v_bcnt_u32_b32 v101, v102, v103\n v_bcnt_u32_b32 v105, v106, v107\n")
DEFK(k_fma, "v_fma_f32 v100, v101, v102, v103\n v_fma_f32 v104, v105, v106, v107\n v_fma_f32 v108, v109, v110, v111\n v_fma_f32 v112, v113, v114, v115\n"
            "v_fma_f32 v116, v117, v118, v119\n v_fma_f32 v120, v121, v122, v123\n v_fma_f32 v101, v102, v103, v104\n v_fma_f32 v105, v106, v107, v108\n")
DEFK(k_lshl2, "v_lshlrev_b32 v100, 1, v101\n v_lshrrev_b32 v104, 31, v105\n v_lshlrev_b32 v108, 1, v109\n v_lshrrev_b32 v112, 31, v113\n"
              "v_lshlrev_b32 v116, 1, v117\n v_lshrrev_b32 v120, 31, v121\n v_lshlrev_b32 v102, 1, v103\n v_lshrrev_b32 v106, 31, v107\n")

DEFK(k_shr64, "v_lshrrev_b64 v[100:101], v102, v[104:105]\n v_lshrrev_b64 v[106:107], v108, v[110:111]\n v_lshrrev_b64 v[112:113], v114, v[116:117]\n v_lshrrev_b64 v[118:119], v120, v[122:123]\n"
              "v_lshrrev_b64 v[102:103], v104, v[106:107]\n v_lshrrev_b64 v[108:109], v110, v[112:113]\n v_lshrrev_b64 v[114:115], v116, v[118:119]\n v_lshrrev_b64 v[120:121], v122, v[100:101]\n")
DEFK(k_min3, "v_min3_u32 v100, v101, v102, v103\n v_min3_u32 v104, v105, v106, v107\n v_min3_u32 v108, v109, v110, v111\n v_min3_u32 v112, v113, v114, v115\n"
             "v_min3_u32 v116, v117, v118, v119\n v_min3_u32 v120, v121, v122, v123\n v_min3_u32 v101, v102, v103, v104\n v_min3_u32 v105, v106, v107, v108\n")
DEFK(k_alignbit_v, "v_alignbit_b32 v100, v101, v102, v103\n v_alignbit_b32 v104, v105, v106, v107\n v_alignbit_b32 v108, v109, v110, v111\n v_alignbit_b32 v112, v113, v114, v115\n"
                   "v_alignbit_b32 v116, v117, v118, v119\n v_alignbit_b32 v120, v121, v122, v123\n v_alignbit_b32 v101, v102, v103, v104\n v_alignbit_b32 v105, v106, v107, v108\n")
DEFK(k_mov64, "v_mov_b64 v[100:101], v[102:103]\n v_mov_b64 v[104:105], v[106:107]\n v_mov_b64 v[108:109], v[110:111]\n v_mov_b64 v[112:113], v[114:115]\n"
              "v_mov_b64 v[116:117], v[118:119]\n v_mov_b64 v[120:121], v[122:123]\n v_mov_b64 v[102:103], v[104:105]\n v_mov_b64 v[106:107], v[108:109]\n")

template <typename K> void run(const char* name, K kern, unsigned* d, int waves_per_simd) {
  const int iters = 2048, grid = 256 * waves_per_simd;  // blocks of 256 = 4 waves = 1 per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 8);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double winstr = (double)grid * 4 * iters * 16.0 * 8;
  double rate = winstr / (ms * 1e-3) / 1024.0;
  printf("%-34s w/SIMD=%d %8.3f ms  %6.3f G winstr/s/SIMD  (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ms, rate / 1e9, 2.4e9 / rate);
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run("VOP2 and/or/xor (2 src)", k_and2, d, w);
    run("v_bitop3 (3 banks)", k_bitop3_ok, d, w);
    run("v_bitop3 (same bank)", k_bitop3_conf, d, w);
    run("v_bitop3 (2 distinct src)", k_bitop3_2src, d, w);
    run("v_alignbit", k_alignbit, d, w);
    run("v_alignbit (shift in a VGPR)", k_alignbit_v, d, w);
    run("v_lshrrev_b64", k_shr64, d, w);
    run("v_min3_u32", k_min3, d, w);
    run("v_mov_b64", k_mov64, d, w);
    run("v_lshl_or", k_lshl_or, d, w);
    run("v_lshl_add_u64", k_add64, d, w);
    run("v_add_co+v_addc_co (per instr)", k_addc, d, w);
    run("v_bfe_u32 (imm)", k_bfe, d, w);
    run("v_dot4 (3 vgpr)", k_dot4, d, w);
    run("v_dot4 (sgpr weight)", k_dot4_s, d, w);
    run("v_and_or / v_or3", k_and_or, d, w);
    run("v_bcnt_u32_b32", k_bcnt, d, w);
    run("v_fma_f32", k_fma, d, w);
    run("v_lshlrev/v_lshrrev (VOP2)", k_lshl2, d, w);
    printf("\n");
  }
  return 0;
}
