// trace_kernel.hip -- K3: traceback of the reported end positions on the device.
//
// One thread per report.  Same rules as the reference's traceback (reference:
// src/search.rs:1477-1478 for the window text[e-(m+k) .. e), src/trace.rs:80-103 for the local
// matrix with L[j][0] = j and L[0][i] = 0, src/trace.rs:337-365 for the greedy walk preferring
// '=', then 'X', 'D', 'I').  Only the diagonals a <=k alignment can touch are computed: every
// cell the walk visits lies on an optimal alignment ending in (m, e), hence on window diagonals
// [dend-k, dend+k]; one more diagonal on each side covers the neighbours the walk compares, and
// values are saturated at k+1 (DESIGN.md "traceback").  The band lives in a per-thread slice of
// a global scratch buffer; reports are rare, so this kernel is microseconds.
#include <hip/hip_runtime.h>

#include "common.h"

namespace sassy_hip {

__device__ __forceinline__ uint32_t d_iupac_code(uint32_t c) {
  // letter (5 low bits) -> base set, 255 = not a letter (reference: src/profiles/iupac.rs:281-317)
  const uint32_t i = c & 31u;
  return i == 1 ? 1 : i == 3 ? 2 : i == 20 ? 4 : i == 21 ? 4 : i == 7 ? 8 : i == 14 ? 15
       : i == 18 ? 9 : i == 25 ? 6 : i == 19 ? 10 : i == 23 ? 5 : i == 11 ? 12 : i == 13 ? 3
       : i == 2 ? 14 : i == 4 ? 13 : i == 8 ? 7 : i == 22 ? 11 : i == 24 ? 0 : 255;
}
__device__ __forceinline__ bool d_scan_eq(uint32_t pr, uint32_t p, uint32_t t) {
  if (pr == PROFILE_DNA) return ((p >> 1) & 3u) == ((t >> 1) & 3u);
  if (pr == PROFILE_IUPAC) return ((d_iupac_code(p) & d_iupac_code(t)) & 0x0Fu) != 0;
  return p == t;
}
__device__ __forceinline__ bool d_is_match(uint32_t pr, uint32_t p, uint32_t t) {
  if (pr == PROFILE_DNA) return (p | 0x20u) == (t | 0x20u);
  if (pr == PROFILE_IUPAC) return (d_iupac_code(p) & d_iupac_code(t)) > 0;
  return p == t;
}

// IN_LDS: the band of every thread of the block fits into LDS (64 * scratch_stride bytes of dynamic
// shared memory) -- the usual case; otherwise it lives in a global scratch slice.
template <typename Cell, bool IN_LDS>
__global__ __launch_bounds__(64) void trace_kernel(const TraceParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char trace_smem[];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nthreads = gridDim.x * blockDim.x;
  uint32_t count = *P.cand_count;
  if (count > P.cand_cap) count = P.cand_cap;
  const long m = (long)P.m, k = (long)P.k;
  const long bw = 2 * k + 3;
  const int inf = (int)k + 1;
  Cell* L = IN_LDS ? reinterpret_cast<Cell*>(trace_smem + (size_t)threadIdx.x * P.scratch_stride)
                   : reinterpret_cast<Cell*>(P.scratch + (uint64_t)tid * P.scratch_stride);

  for (uint32_t c = tid; c < count; c += nthreads) {
    const uint64_t e = P.cand[c].pos;                 // global end position
    const uint64_t fill = (uint64_t)(m + k);
    const uint64_t o = e > fill ? e - fill : 0;        // global window start
    const uint64_t we = e < P.total_len ? e : P.total_len;
    const long wl = (long)(we - o);
    const uint8_t* win = P.text + (o - P.global_offset);
    const long dend = wl - m, dlo = dend - k - 1, dhi = dend + k + 1;
    auto at = [&](long j, long i) -> int {
      if (i < 0 || i > wl) return inf;
      const long d = i - j;
      if (d < dlo || d > dhi) return inf;
      return (int)L[j * bw + (d - dlo)];
    };
    for (long j = 0; j <= m; ++j) {
      long ilo = j + dlo, ihi = j + dhi;
      if (ilo < 0) ilo = 0;
      if (ihi > wl) ihi = wl;
      const uint32_t pc = j > 0 ? P.pattern[j - 1] : 0u;
      for (long i = ilo; i <= ihi; ++i) {
        int v;
        if (j == 0) v = 0;
        else if (i == 0) v = j < inf ? (int)j : inf;
        else {
          v = at(j - 1, i - 1) + (d_scan_eq(P.profile, pc, win[i - 1]) ? 0 : 1);
          const int l = at(j, i - 1) + 1, u = at(j - 1, i) + 1;
          v = v < l ? v : l;
          v = v < u ? v : u;
          v = v < inf ? v : inf;
        }
        L[j * bw + (i - j - dlo)] = (Cell)v;
      }
    }
    TraceRec r;
    r.text_start = 0;
    r.text_end = we;
    r.cost = 0;
    r.nops = 0;
    r.ok = 0;
    r.cand = c;
    long j = m, i = wl;
    int g = at(j, i);
    r.cost = g;
    uint8_t* ops = P.out_ops + (uint64_t)c * P.ops_stride;
    uint32_t nops = 0;
    bool ok = g <= (int)k;
    while (ok && j > 0) {
      if (nops >= P.ops_stride) { ok = false; break; }
      if (i > 0 && at(j - 1, i - 1) == g && d_is_match(P.profile, P.pattern[j - 1], win[i - 1])) {
        ops[nops++] = '='; --j; --i; continue;
      }
      g -= 1;
      if (g < 0) { ok = false; break; }
      if (i > 0 && at(j - 1, i - 1) == g) { ops[nops++] = 'X'; --j; --i; continue; }
      if (i > 0 && at(j, i - 1) == g) { ops[nops++] = 'D'; --i; continue; }
      if (at(j - 1, i) == g) { ops[nops++] = 'I'; --j; continue; }
      ok = false;  // the reference panics here ("Trace failed! No ancestor found")
    }
    if (ok && g != 0) ok = false;
    r.text_start = o + (uint64_t)i;
    r.nops = nops;  // written end -> start; the host reverses while run-length encoding
    r.ok = ok ? 1u : 0u;
    P.out[c] = r;
  }
}

hipError_t launch_trace(const TraceParams& P, uint32_t nblocks, hipStream_t stream) {
  const size_t lds = (size_t)64 * P.scratch_stride;
  const bool in_lds = lds <= 64 * 1024;
  if (P.k + 1 <= 255) {
    if (in_lds) hipLaunchKernelGGL((trace_kernel<uint8_t, true>), dim3(nblocks), dim3(64), lds, stream, P);
    else hipLaunchKernelGGL((trace_kernel<uint8_t, false>), dim3(nblocks), dim3(64), 0, stream, P);
  } else {
    if (in_lds) hipLaunchKernelGGL((trace_kernel<uint16_t, true>), dim3(nblocks), dim3(64), lds, stream, P);
    else hipLaunchKernelGGL((trace_kernel<uint16_t, false>), dim3(nblocks), dim3(64), 0, stream, P);
  }
  return hipGetLastError();
}

}  // namespace sassy_hip
