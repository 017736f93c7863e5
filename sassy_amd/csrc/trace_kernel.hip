// trace_kernel.hip -- K3: traceback of the reported end positions on the device.
//
// One thread per report.  Same rules as the reference's traceback (reference:
// src/search.rs:1477-1478 for the window text[e-(m+k) .. e), src/trace.rs:80-103 for the local
// matrix with L[j][0] = j and L[0][i] = 0, src/trace.rs:337-365 for the greedy walk preferring
// '=', then 'X', 'D', 'I').  Only the diagonals a <=k alignment can touch are computed: every
// cell the walk visits lies on an optimal alignment ending in (m, e), hence on window diagonals
// [dend-k, dend+k]; one more diagonal on each side covers the neighbours the walk compares, and
// values are saturated at k+1 (DESIGN.md "traceback").
//
// Band coordinates: cell (j, i) of the window matrix is stored at row j, column b = i - j - dlo
// (0 <= b < 2k+3).  Its neighbours are (j-1, i-1) -> same b, (j, i-1) -> b-1, (j-1, i) -> b+1.
// Cells outside the window store k+1, so no position checks are needed when reading.
//
// Per-thread slice = band | window bytes | ops (end -> start), in LDS when 64 slices fit, else in
// a global scratch buffer.  Reports are few (thousands), so the kernel is a latency chain, not a
// throughput problem: the window arrives with a handful of independent 16-byte loads, and for
// small k the band row and the text bytes under it stay in registers while the band is filled
// (LDS only records the rows for the walk).  The thread finishes its record completely:
// run-length encoded cigar text (pa-types' Cigar::to_string form, reference src/search.rs:75-98)
// and a sassy_hip_Match-layout row, so the host only copies.
#include <hip/hip_runtime.h>

#include "common.h"

namespace sassy_hip {

// Letter (5 low bits of the byte) -> IUPAC base set, 255 = not a letter
// (reference: src/profiles/iupac.rs:281-317).
__constant__ uint8_t kIupacCode[32] = {
    255, 1, 14, 2, 13, 255, 255, 8, 7, 255, 255, 12, 255, 3, 15, 255,
    255, 255, 9, 10, 4, 4, 11, 5, 0, 6, 255, 255, 255, 255, 255, 255};

// Characters are compared through one formula for all profiles, so that the fill and walk loops
// stay small (the kernel runs once per call on a few dozen waves: instruction fetch of cold code
// is a visible part of its time).  Stored byte s(c): Dna / Ascii the raw byte, Iupac its base
// set.  With X = iupac ? p & t : p ^ t:
//   scan equality   (profile eq of the scan)        : ((X & emask) != 0) == iupac
//   display match   (Profile::is_match, '=' vs 'X') : ((X & mmask) != 0) == iupac
// Dna: emask 6 ((c>>1)&3 codes, src/profiles/dna.rs:19-60), mmask 0xDF (case-insensitive byte);
// Iupac: emask 15 (non-letters = 255 act as N in the scan), mmask 255; Ascii: both 255.
struct CharRule {
  uint32_t iupac, emask, mmask;
};
__device__ __forceinline__ CharRule char_rule(uint32_t profile) {
  CharRule r;
  r.iupac = profile == PROFILE_IUPAC ? 1u : 0u;
  r.emask = profile == PROFILE_DNA ? 6u : profile == PROFILE_IUPAC ? 15u : 255u;
  r.mmask = profile == PROFILE_DNA ? 0xDFu : 255u;
  return r;
}
__device__ __forceinline__ bool rule_hit(const CharRule& r, uint32_t p, uint32_t t, uint32_t mask) {
  const uint32_t x = r.iupac ? (p & t) : (p ^ t);
  return ((x & mask) != 0u) == (r.iupac != 0u);
}

// Window bytes of one report into the thread's slice: aligned 16-byte chunks, all loads of a batch
// in flight together.  Only chunks that contain a valid text byte are touched (the text buffer is
// 16-byte aligned, so such a chunk never leaves the allocation).  Returns the offset of window
// byte 0 inside the slice's window area.
__device__ __forceinline__ uint32_t load_window(const uint8_t* text, uint64_t o_rel, int wl,
                                                unsigned char* wbuf) {
  const uint64_t a0 = o_rel & ~15ull;
  const uint32_t skew = (uint32_t)(o_rel - a0);
  const uint32_t nch = wl > 0 ? (skew + (uint32_t)wl + 15u) / 16u : 0u;
  const uint4* src = reinterpret_cast<const uint4*>(text + a0);
  uint32_t* dst = reinterpret_cast<uint32_t*>(wbuf);
  for (uint32_t q0 = 0; q0 < nch; q0 += 4) {
    uint4 v[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) v[u] = q0 + u < nch ? src[q0 + u] : uint4{0, 0, 0, 0};
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      if (q0 + u < nch) {
        dst[4 * (q0 + u) + 0] = v[u].x;
        dst[4 * (q0 + u) + 1] = v[u].y;
        dst[4 * (q0 + u) + 2] = v[u].z;
        dst[4 * (q0 + u) + 3] = v[u].w;
      }
    }
  }
  return skew;
}

// Cigar text of a finished walk into `sbuf` (slice memory): run-length encoded, start -> end
// (pa-types' Cigar::to_string form).  Returns its length; the text is NUL-padded to a dword.
__device__ __forceinline__ uint32_t rle_text(const unsigned char* ops, uint32_t nops, bool ok, unsigned char* sbuf) {
  uint32_t w = 0;
  int idx = ok ? (int)nops - 1 : -1;
  while (idx >= 0) {
    const unsigned char op = ops[idx];
    uint32_t run = 1;
    while (idx - (int)run >= 0 && ops[idx - (int)run] == op) ++run;
    idx -= (int)run;
    uint32_t p10 = 1;
    while (p10 * 10 <= run) p10 *= 10;
    while (p10) { sbuf[w++] = (unsigned char)('0' + run / p10); run %= p10; p10 /= 10; }
    sbuf[w++] = op;
  }
  sbuf[w] = 0; sbuf[w + 1] = 0; sbuf[w + 2] = 0; sbuf[w + 3] = 0;
  return w;
}
// Window of a report: text[o .. we) with o = e - (m+k) clipped at the start of the text and we = e
// clipped at its end (reference: src/search.rs:1477-1478).  In a multi-text buffer "the text" is
// the one the report belongs to (its index travels in the candidate's flags); `base` is that text's
// first byte, so that the record carries text-relative coordinates.
struct Window {
  uint64_t o, we, base;
  uint32_t text_idx;
  bool skip;
};
__device__ __forceinline__ Window report_window(const TraceParams& P, const Candidate& cd, uint32_t c = 0) {
  Window w;
  const uint64_t fill = (uint64_t)P.m + P.k;
  const uint64_t e = cd.pos;
  w.skip = (cd.flags & kCandDrop) != 0;
  if (P.texts.n) {
    w.text_idx = P.report_text ? P.report_text[c] : cd.flags >> kCandTextShift;
    w.base = P.texts.start[w.text_idx];
    const uint64_t te = w.base + P.texts.len[w.text_idx];
    w.o = e > w.base + fill ? e - fill : w.base;
    w.we = e < te ? e : te;
  } else {
    w.text_idx = 0;
    w.base = 0;
    w.o = e > fill ? e - fill : 0;
    w.we = e < P.total_len ? e : P.total_len;
  }
  return w;
}

// Overhang: cost of having the first j pattern characters hang over the start of the text,
// floor(min(j, max_overhang) * alpha) + max(0, j - max_overhang) in f32 (reference: src/trace.rs:36-47).
__device__ __forceinline__ int ov_left(const TraceParams& P, int j) {
  const int a = (uint32_t)j < P.max_overhang ? j : (int)P.max_overhang;
  return __float2int_rd((float)a * P.alpha) + (j - a);
}

__device__ __forceinline__ MatchOut make_row(const TraceParams& P, uint32_t c, const Window& win, uint64_t text_start,
                                             int cost, uint32_t len, bool ok, uint32_t pattern_start = 0,
                                             uint32_t pattern_end = 0xFFFFFFFFu, uint32_t pattern_idx = 0) {
  MatchOut r;
  r.pattern_idx = pattern_idx;
  r.text_idx = win.text_idx;
  r.text_start = text_start - win.base;
  r.text_end = win.we - win.base;
  r.pattern_start = pattern_start;
  r.pattern_end = pattern_end == 0xFFFFFFFFu ? P.m : pattern_end;
  r.cost = cost;
  r.strand = 0;
  r.pad_[0] = ok ? 0 : kTraceFailed;
  r.pad_[1] = r.pad_[2] = 0;
  r.cigar_off = c * P.str_stride;
  r.cigar_len = len;
  return r;
}

// KT >= 0: k is the compile-time constant KT (band row in registers); KT < 0: any k.
template <typename Cell, bool IN_LDS, int KT>
__global__ __launch_bounds__(64) void trace_kernel(const TraceParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char trace_smem[];
  uint32_t count = *P.cand_count;
  if (count <= P.count_min || count > P.count_max) return;  // the other kernel shape handles it
  if (count > P.cand_cap) count = P.cand_cap;
  const int m = (int)P.m, k = KT >= 0 ? KT : (int)P.k;
  const int bw = 2 * k + 3;
  const int inf = k + 1;
  const uint32_t tid = threadIdx.x;
  // block-shared pattern copy behind the 64 slices (LDS mode)
  unsigned char* spat = trace_smem + (size_t)64 * P.scratch_stride;
  const CharRule rule = char_rule(P.profile);
  // (pattern_stride != 0: many patterns -- a report brings its own, named in its flags' upper bits; the threads of a
  // workgroup then read their rows from the patterns' array, a few KB that stay in L1 / L2)
  const bool many = P.pattern_stride != 0;
  if (IN_LDS && !many) {
    for (uint32_t x = tid; x < P.m; x += 64) {
      const uint32_t ch = P.pattern[x];
      spat[x] = (unsigned char)(rule.iupac ? kIupacCode[ch & 31u] : ch);
    }
    __syncthreads();
  }
  const uint8_t* my_pat = P.pattern;
  auto pat_at = [&](int j) -> uint32_t {
    if (IN_LDS && !many) return spat[j];
    const uint32_t ch = my_pat[j];
    return rule.iupac ? kIupacCode[ch & 31u] : ch;
  };
  unsigned char* slice = IN_LDS ? trace_smem + (size_t)tid * P.scratch_stride
                                : P.scratch + ((uint64_t)blockIdx.x * 64 + tid) * P.scratch_stride;
  Cell* L = reinterpret_cast<Cell*>(slice);
  unsigned char* wbuf = slice + P.band_bytes;
  unsigned char* ops = slice + P.band_bytes + P.win_bytes;

  for (uint32_t c = blockIdx.x * 64 + tid; c < count; c += gridDim.x * 64) {
    const Candidate cd = P.cand[c];
    const Window W = report_window(P, cd, c);
    if (W.skip) continue;
    uint32_t pattern_idx = 0;
    if (many) {
      pattern_idx = cd.flags >> kCandTextShift;
      my_pat = P.pattern + (size_t)pattern_idx * P.pattern_stride;
    }
    const uint64_t o = W.o, we = W.we;                     // global window bounds
    const int wl = (int)(we - o);
    // Rc strand without a reversed copy: the window [o, o + wl) of the reversed text is the forward
    // bytes [n - o - wl, n - o), fetched the same way and turned around in place
    const uint32_t skew = load_window(P.text, P.rev_n ? P.rev_n - (o - P.global_offset) - (uint64_t)wl : o - P.global_offset,
                                      wl, wbuf);
    unsigned char* win = wbuf + skew;
    if (P.rev_n) {
      for (int x = 0; x < wl / 2; ++x) {
        const unsigned char a = win[x];
        win[x] = win[wl - 1 - x];
        win[wl - 1 - x] = a;
      }
    }
    if (rule.iupac)
      for (int x = 0; x < wl; ++x) win[x] = kIupacCode[win[x] & 31u];
    auto text_at = [&](int i) -> uint32_t { return win[i]; };

    // overhang: the end cell may lie past the text (columns wl+1 .. iend are 'N')
    const bool alpha_on = KT < 0 && P.use_alpha != 0;
    const int iend = alpha_on ? (int)(cd.pos - o) : wl;
    const uint32_t ncode = rule.iupac ? 15u : (uint32_t)'N';
    const int dend = iend - m, dlo = dend - k - 1;
    // ---- fill the band ----
    if constexpr (KT >= 0) {
      constexpr int BW = 2 * KT + 3;
      int prev[BW];
      uint32_t tx[BW];  // tx[b] = text byte compared in cell (j, i = j+dlo+b): text[i-1]
#pragma unroll
      for (int b = 0; b < BW; ++b) {
        const int i = dlo + b;
        prev[b] = (i < 0 || i > wl) ? inf : 0;
        L[b] = (Cell)prev[b];
        const int ti = dlo + b;  // row 1: i - 1 = 1 + dlo + b - 1
        tx[b] = (ti >= 0 && ti < wl) ? text_at(ti) : 0u;
      }
      for (int j = 1; j <= m; ++j) {
        const uint32_t pc = pat_at(j - 1);
        // the text byte that enters the band on the right in the next row
        const int tn = j + dlo + BW - 1;
        const uint32_t tnext = (tn >= 0 && tn < wl) ? text_at(tn) : 0u;
        Cell* row = L + (size_t)j * BW;
        int cur[BW];
        int left = inf;
#pragma unroll
        for (int b = 0; b < BW; ++b) {
          const int i = j + dlo + b;
          int v;
          if (i < 0 || i > wl) v = inf;
          else if (i == 0) v = j < inf ? j : inf;
          else {
            v = prev[b] + (rule_hit(rule, pc, tx[b], rule.emask) ? 0 : 1);
            const int l = left + 1;
            const int u = (b + 1 < BW ? prev[b + 1] : inf) + 1;
            v = v < l ? v : l;
            v = v < u ? v : u;
            v = v < inf ? v : inf;
          }
          cur[b] = v;
          left = v;
        }
#pragma unroll
        for (int b = 0; b < BW; ++b) {
          row[b] = (Cell)cur[b];
          prev[b] = cur[b];
          tx[b] = b + 1 < BW ? tx[b + 1] : tnext;
        }
      }
    } else {
      for (int j = 0; j <= m; ++j) {
        const uint32_t pc = j > 0 ? pat_at(j - 1) : 0u;
        Cell* row = L + (size_t)j * bw;
        const Cell* prev = row - bw;
        int left = inf;
        for (int b = 0; b < bw; ++b) {
          const int i = j + dlo + b;
          int v;
          if (i < 0 || i > iend) v = inf;
          else if (j == 0) v = 0;
          else if (i == 0) {
            const int lc = alpha_on ? ov_left(P, j) : j;
            v = lc < inf ? lc : inf;
          } else {
            const uint32_t tc = i - 1 < wl ? text_at(i - 1) : ncode;
            v = (int)prev[b] + (rule_hit(rule, pc, tc, rule.emask) ? 0 : 1);
            const int l = left + 1;
            const int u = (b + 1 < bw ? (int)prev[b + 1] : inf) + 1;
            v = v < l ? v : l;
            v = v < u ? v : u;
            v = v < inf ? v : inf;
          }
          row[b] = (Cell)v;
          left = v;
        }
      }
    }
    // ---- greedy walk from (m, iend) ----
    int j = m, i = iend;
    int g = (int)L[(size_t)m * bw + (k + 1)];
    int cost = g;
    uint32_t pattern_start = 0, pattern_end = P.m;
    const uint32_t max_ops = P.m + P.k + 1;
    uint32_t nops = 0;
    bool ok = g <= k;
    if (ok && i > wl) {  // the match ends past the text: step back along the diagonal (trace.rs:299-312)
      const int over = i - wl;
      if (over > m) ok = false;
      else {
        pattern_end -= (uint32_t)over;
        cost += __float2int_rd((float)over * P.alpha);
        i -= over;
        j -= over;
      }
    }
    while (ok && j > 0) {
      if (alpha_on && i == 0) {  // the rest of the pattern hangs over the text start (trace.rs:322-335)
        pattern_start = (uint32_t)j;
        g -= ov_left(P, j);
        break;
      }
      if (nops >= max_ops) { ok = false; break; }
      const int b = i - j - dlo;
      const Cell* row = L + (size_t)j * bw;
      const Cell* prev = row - bw;
      const int diag = i > 0 ? (int)prev[b] : inf;
      if (diag == g && rule_hit(rule, pat_at(j - 1), text_at(i - 1), rule.mmask)) {
        ops[nops++] = '='; --j; --i; continue;
      }
      g -= 1;
      if (g < 0) { ok = false; break; }
      if (diag == g) { ops[nops++] = 'X'; --j; --i; continue; }
      const int lft = (i > 0 && b > 0) ? (int)row[b - 1] : inf;
      if (lft == g) { ops[nops++] = 'D'; --i; continue; }
      const int up = (b + 1 < bw) ? (int)prev[b + 1] : inf;
      if (up == g) { ops[nops++] = 'I'; --j; continue; }
      ok = false;  // the reference panics here ("Trace failed! No ancestor found")
    }
    if (ok && g != 0) ok = false;
    // the reference asserts that the traced cost does not exceed the scanned one
    // (src/search.rs:1672-1685)
    if (cost > cd.cost) ok = false;

    // ---- cigar text and the finished row, to the device arrays and (head of the list) the host ----
    unsigned char* sbuf = ops + P.ops_bytes;
    const uint32_t w = rle_text(ops, nops, ok, sbuf);
    const MatchOut r = make_row(P, c, W, o + (uint64_t)i, cost, w, ok, pattern_start, pattern_end, pattern_idx);
    const uint32_t ndw = w / 4 + 1;
    uint32_t* dstr = reinterpret_cast<uint32_t*>(P.out_str + (uint64_t)c * P.str_stride);
    const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(sbuf);
    for (uint32_t x = 0; x < ndw; ++x) dstr[x] = ssrc[x];
    P.out[c] = r;
    if (!ok && P.host_flags) __hip_atomic_fetch_or(P.host_flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (c < P.host_cap) {
      uint32_t* hstr = reinterpret_cast<uint32_t*>(P.host_str + (uint64_t)c * P.str_stride);
      for (uint32_t x = 0; x < ndw; ++x) hstr[x] = ssrc[x];
      P.host_out[c] = r;
    }
  }
}

// ---------------------------------------------------------------- wave-per-report variant
// For wider bands (k > 6) one thread per report is a very long serial chain and its slice no
// longer fits LDS.  Here a whole wavefront fills the band of one report: lane b owns band column b
// (2k+3 <= 64 columns).  Within a row, cell b needs its left neighbour: v[b] = min(t[b], v[b-1]+1)
// with t[b] = min(diag + neq, up + 1), which unrolls to v[b] = b + prefix-min over b' <= b of
// (t[b'] - b') -- a 6-step DPP scan across the wave.  The diagonal neighbour is the lane's own
// previous value, the upper neighbour the next lane's (DPP wave shift).  The walk and the cigar
// text are wave-uniform (every lane follows the same path through LDS).
__device__ __forceinline__ int dpp_min_scan(int x) {
  constexpr int kId = 0x7FFFFFFF;
  // inclusive prefix minimum over the 64 lanes (lanes without a source keep the identity)
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x111, 0xF, 0xF, false));  // row_shr:1
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x112, 0xF, 0xF, false));  // row_shr:2
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x114, 0xF, 0xF, false));  // row_shr:4
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x118, 0xF, 0xF, false));  // row_shr:8
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
  return x;
}

// the same within the first 16 lanes (one DPP row): enough for a band of <= 16 columns
__device__ __forceinline__ int dpp_min_scan16(int x) {
  constexpr int kId = 0x7FFFFFFF;
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x111, 0xF, 0xF, false));  // row_shr:1
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x112, 0xF, 0xF, false));  // row_shr:2
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x114, 0xF, 0xF, false));  // row_shr:4
  x = min(x, __builtin_amdgcn_update_dpp(kId, x, 0x118, 0xF, 0xF, false));  // row_shr:8
  return x;
}

// ALPHA: overhang (use_alpha) -- the common searches get a fill loop without its branches
template <typename Cell, bool ALPHA>
__global__ __launch_bounds__(256) void trace_wave_kernel(const TraceParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char trace_smem[];
  uint32_t count = *P.cand_count;
  // self-ranking mode (no rank kernels in front): the control block goes to the host from here
  if (P.unsorted && blockIdx.x == 0 && threadIdx.x < 4 && P.host_ctl)
    P.host_ctl[threadIdx.x] = reinterpret_cast<const uint4*>(P.cand_count)[threadIdx.x];
  if (count <= P.count_min || count > P.count_max) return;  // the other kernel shape handles it
  if (count > P.cand_cap) count = P.cand_cap;
  const int m = (int)P.m, k = (int)P.k;
  const int bw = 2 * k + 3;  // <= 64
  const int inf = k + 1;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const CharRule rule = char_rule(P.profile);
  // LDS: per wave: pattern codes | then per wave: band rows | window | ops.  With many patterns
  // (pattern_stride != 0: the pattern-tiled search) a report brings its own pattern, loaded per report.
  const uint32_t pat_bytes = (P.m + 15u) & ~15u;
  unsigned char* spat = trace_smem + (size_t)wave * pat_bytes;
  auto load_pattern = [&](const uint8_t* src) {
    for (uint32_t x = lane; x < P.m; x += 64u) {
      const uint32_t ch = src[x];
      spat[x] = (unsigned char)(rule.iupac ? kIupacCode[ch & 31u] : ch);
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (P.pattern_stride == 0) load_pattern(P.pattern);
  unsigned char* slice = trace_smem + 4u * pat_bytes + (size_t)wave * P.scratch_stride;
  Cell* L = reinterpret_cast<Cell*>(slice);
  unsigned char* win = slice + P.band_bytes;
  unsigned char* ops = slice + P.band_bytes + P.win_bytes;
  const uint32_t n_waves = gridDim.x * 4;
  // self-ranking: the end positions of all reports, once per workgroup, into LDS (behind the four slices) -- every
  // wave then ranks its report against LDS instead of walking the list in L2 with dependent loads (12 us -> 2 us)
  unsigned long long* lpos = reinterpret_cast<unsigned long long*>(trace_smem + 4u * pat_bytes + 4u * (size_t)P.scratch_stride);
  const bool pos_in_lds = P.unsorted != nullptr && P.rank_lds != 0 && count <= P.rank_lds;
  if (pos_in_lds) {
    // (four loads in flight per thread: a dozen dependent round trips to L2 were 2 us of every report's prologue)
    uint32_t x = threadIdx.x;
    for (; x + 768u < count; x += 1024u) {
      const unsigned long long p0 = P.unsorted[x].pos, p1 = P.unsorted[x + 256u].pos, p2 = P.unsorted[x + 512u].pos,
                               p3 = P.unsorted[x + 768u].pos;
      lpos[x] = p0; lpos[x + 256u] = p1; lpos[x + 512u] = p2; lpos[x + 768u] = p3;
    }
    for (; x < count; x += 256u) lpos[x] = P.unsorted[x].pos;
    __syncthreads();
  }

  unsigned long long tp0 = P.probe ? wall_clock64() : 0ull;
  auto tick = [&](int slot) {
    if (P.probe) {
      const unsigned long long t = wall_clock64();
      if (lane == 0) P.probe[(size_t)(blockIdx.x * 4 + wave) * 8 + slot] += t - tp0;  // (the wave's own row)
      tp0 = t;
    }
  };
  for (uint32_t u = blockIdx.x * 4 + wave; u < count; u += n_waves) {
    uint32_t c = u;
    Candidate cd;
    uint32_t wpre = 0;       // self-ranking: the first 64 window bytes, requested before the ranking scan
    bool have_wpre = false;
    if (P.unsorted) {
      // The reports arrive in append order; the result order is by end position (unique per strand).
      // With a few thousand reports every wave ranks its own: it counts the reports that precede
      // it (64 lanes over the list, which sits in L2) and files the report, and everything it
      // derives from it, under that rank -- no ranking kernels, two launches fewer per search.
      cd = P.unsorted[u];
      if (P.texts.n == 0 && P.rev_n == 0) {  // (the window does not depend on the rank)
        const Window Wp = report_window(P, cd, 0);
        const int wlp = (int)(Wp.we - Wp.o);
        // the lane that made the report kept the block under it (TextStash): no walk through the page tables of a
        // multi-GB text for 40 bytes
        const uint8_t* src0 = P.text + (Wp.o - P.global_offset);
        const uint32_t slot1 = (P.stash != nullptr && P.pattern_stride == 0) ? (cd.flags >> kCandTextShift) : 0u;
        if (slot1 != 0 && slot1 <= P.stash_cap) {
          const TextStash* ts = P.stash + (slot1 - 1u);
          const uint64_t base = ts->base, orel = Wp.o - P.global_offset;
          // (a window may begin in front of the buffer -- reference-lane searches hand in a pointer into a larger text)
          if (Wp.o >= P.global_offset && orel >= base && orel + (uint64_t)wlp <= base + 64)
            src0 = reinterpret_cast<const uint8_t*>(ts->text) + (orel - base);
        }
        if ((int)lane < wlp) wpre = src0[lane];
        have_wpre = wlp <= 64;
      }
      // rank = reports with a smaller end position + (dedup) earlier copies of this very report: copies of one
      // position then fill consecutive slots, and every copy but the first is a kCandDrop record the host skips
      uint32_t r = 0;
      // (dedup) a conditional report is certain after all when a copy of it -- another window's view of the same end
      // position -- is not conditional: that window saw what settles the plateau state
      const bool want_twin = P.dedup && (cd.flags & kCandCond) != 0;  // wave-uniform, rare
      bool twin_uncond = false;
      if (want_twin)
        for (uint32_t v = lane; v < count; v += 64) {
          const Candidate o = P.unsorted[v];
          twin_uncond |= o.pos == cd.pos && v != u && !(o.flags & kCandCond);
        }
      if (pos_in_lds) {
        uint32_t v = lane;
        for (; v + 192 < count; v += 256) {  // (four LDS reads in flight)
          const unsigned long long p0 = lpos[v], p1 = lpos[v + 64], p2 = lpos[v + 128], p3 = lpos[v + 192];
          r += (p0 < cd.pos ? 1u : 0u) + (p1 < cd.pos ? 1u : 0u) + (p2 < cd.pos ? 1u : 0u) + (p3 < cd.pos ? 1u : 0u);
          if (P.dedup)
            r += ((p0 == cd.pos && v < u) ? 0x10000u : 0u) + ((p1 == cd.pos && v + 64 < u) ? 0x10000u : 0u) +
                 ((p2 == cd.pos && v + 128 < u) ? 0x10000u : 0u) + ((p3 == cd.pos && v + 192 < u) ? 0x10000u : 0u);
        }
        for (; v < count; v += 64) {
          const unsigned long long p0 = lpos[v];
          r += p0 < cd.pos ? 1u : 0u;
          if (P.dedup) r += (p0 == cd.pos && v < u) ? 0x10000u : 0u;
        }
      } else {
        // four independent loads in flight per lane (the list is L2 resident; the loop is latency bound)
        uint32_t v = lane;
        for (; v + 192 < count; v += 256) {
          const uint64_t p0 = P.unsorted[v].pos, p1 = P.unsorted[v + 64].pos, p2 = P.unsorted[v + 128].pos,
                         p3 = P.unsorted[v + 192].pos;
          r += (p0 < cd.pos ? 1u : 0u) + (p1 < cd.pos ? 1u : 0u) + (p2 < cd.pos ? 1u : 0u) + (p3 < cd.pos ? 1u : 0u);
          if (P.dedup)
            r += ((p0 == cd.pos && v < u) ? 0x10000u : 0u) + ((p1 == cd.pos && v + 64 < u) ? 0x10000u : 0u) +
                 ((p2 == cd.pos && v + 128 < u) ? 0x10000u : 0u) + ((p3 == cd.pos && v + 192 < u) ? 0x10000u : 0u);
        }
        for (; v < count; v += 64) {
          const uint64_t p0 = P.unsorted[v].pos;
          r += p0 < cd.pos ? 1u : 0u;
          if (P.dedup) r += (p0 == cd.pos && v < u) ? 0x10000u : 0u;
        }
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) r += __shfl_xor(r, d);
      // (count <= kTraceWaveMax = 8192 here: the two 16-bit fields cannot overflow)
      const uint32_t twins = r >> 16;
      c = (r & 0xFFFFu) + twins;
      const bool drop = twins != 0 || (P.dedup && cd.pos < P.min_pos);
      if (drop) cd.flags |= kCandDrop;
      if (want_twin && __any(twin_uncond)) cd.flags &= ~kCandCond;
      if (lane == 0) {
        const_cast<Candidate*>(P.cand)[c] = cd;
        if (c < P.host_cap && P.host_cand) P.host_cand[c] = cd;
        const uint32_t hf = (drop ? 4u : 0u) | ((cd.flags & kCandCond) ? 2u : 0u);
        if (hf && P.host_flags) __hip_atomic_fetch_or(P.host_flags, hf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (drop) continue;
    } else {
      cd = P.cand[c];                                      // wave-uniform
    }
    tick(0);  // prologue (pattern, positions into LDS) + ranking
    Window W = report_window(P, cd, c);
    uint32_t pattern_idx = 0;
    if (P.pattern_stride) {  // many patterns: the flags' upper bits name the pattern, not a text (report_text does)
      pattern_idx = cd.flags >> kCandTextShift;
      __builtin_amdgcn_wave_barrier();  // the previous report's walk is done with spat
      load_pattern(P.pattern + (size_t)pattern_idx * P.pattern_stride);
    }
    if (W.skip) continue;
    const uint64_t o = W.o, we = W.we;
    const int wl = (int)(we - o);
    // overhang: the end cell may lie past the text; columns wl+1 .. iend are virtual 'N'
    const bool alpha_on = ALPHA;
    const int iend = alpha_on ? (int)(cd.pos - o) : wl;
    {  // window -> LDS (coalesced bytes), Iupac letters -> base sets
      const uint8_t* src = P.text + (o - P.global_offset);
      const uint8_t* rsrc = P.text + (P.rev_n - 1 - (o - P.global_offset));  // reversed view: byte x = rsrc[-x]
      for (int x = (int)lane; x < iend; x += 64) {
        const uint32_t ch = x < wl ? ((have_wpre && x < 64) ? wpre : (P.rev_n ? rsrc[-(int64_t)x] : src[x])) : (uint32_t)'N';
        win[x] = (unsigned char)(rule.iupac ? kIupacCode[ch & 31u] : ch);
      }
    }
    __builtin_amdgcn_wave_barrier();
    tick(1);  // window
    const int dend = iend - m, dlo = dend - k - 1;
    const int b = (int)lane;
    const bool in_band = b < bw;
    // ---- fill: row 0, then rows 1..m ----
    int prev;
    {
      const int i = dlo + b;
      prev = (!in_band || i < 0 || i > iend) ? inf : 0;
      if (in_band) L[b] = (Cell)prev;
    }
    if constexpr (ALPHA) {
      for (int j = 1; j <= m; ++j) {
        const uint32_t pc = spat[j - 1];
        const int i = j + dlo + b;
        const bool valid = in_band && i >= 0 && i <= iend;
        // upper neighbour (j-1, i) = band column b+1 of the previous row
        const int up = __builtin_amdgcn_update_dpp(inf, prev, 0x130, 0xF, 0xF, false);  // wave_shl:1
        int t = inf;
        if (valid) {
          if (i == 0) {
            const int lc = ov_left(P, j);
            t = lc < inf ? lc : inf;
          } else {
            const uint32_t tc = win[i - 1];
            const int dg = prev + (rule_hit(rule, pc, tc, rule.emask) ? 0 : 1);
            const int u = up + 1;
            t = dg < u ? dg : u;
          }
        }
        // left dependency: v[b] = b + min_{b' <= b} (t[b'] - b'); invalid cells carry a large value
        int v = dpp_min_scan(valid ? t - b : 0x3FFFFFFF) + b;
        v = (valid && v < inf) ? v : inf;
        if (in_band) L[(size_t)j * bw + b] = (Cell)v;
        prev = v;
      }
    } else {
      // The reports of a call run side by side (three or four waves per SIMD), so a row costs what it issues.  Round 6:
      // the row in the "minus b" domain (q = value - b: what the prefix minimum works on anyway, so nothing is subtracted
      // before the scan or added after it), the compare of the profile and the width of the scan chosen once per report
      // (wave-uniform; four copies of the loop) instead of per row, one unsigned compare for "inside the window", the
      // window's first column a scalar: 33 -> 24 VALU per row and no branches.
      const int lo_j = -(dlo + b);                                     // the row in which this band column is window column 0
      const int n_valid = in_band ? iend - dlo - b - lo_j + 1 : 0;     // rows lo_j .. lo_j + n_valid - 1 lie inside the window
      const uint32_t cnt = n_valid > 0 ? (uint32_t)n_valid : 0u;
      constexpr int kBig = 0x3FFFFFFF;
      constexpr int kId = 0x7FFFFFFF;  // (the scan's identity must be INT_MAX: only then is a step ONE v_min_i32_dpp)
      const int infq = inf - b;
      // the text byte under cell (j, i) is win[i - 1] = wp[j - 1].  No clamping: a cell outside the window reads some
      // other byte of the wave's own slice (the band lies in front of the window, ops and text behind it, and
      // |dlo + b| stays below either size) and is overwritten.
      const unsigned char* wp = win + (dlo + b);
      const int dlo_u = __builtin_amdgcn_readfirstlane(dlo);  // (the report is the wave's: the compiler cannot know)
      auto rows = [&](auto iupac_tag, auto width_tag) {
        constexpr bool IUPAC = decltype(iupac_tag)::value;
        constexpr int WIDTH = decltype(width_tag)::value;  // lanes the band may reach: 16 / 32 / 64
        int prevq = prev - b;
        // lanes outside the band store into a cell of their own at the end of the slice (kTraceWaveDummy bytes nothing
        // reads) with stride 0: the store needs no exec mask, the row no scalar branch (29.8 -> 24.2 us of fill per report
        // at m = 200)
        Cell* Lrow = in_band ? L + bw + b : reinterpret_cast<Cell*>(slice + P.scratch_stride - kTraceWaveDummy) + lane;
        const int Lstride = in_band ? bw : 0;
        // MID: a row in which every band column lies inside the window and behind its first column -- no "inside the
        // window" test, no first-column value; the lanes outside the band are held at infq by the clamp's lower bound
        const int loq = in_band ? -kBig : infq;
        auto row = [&](int j, uint32_t pc, uint32_t tc, auto mid_tag) {
          constexpr bool MID = decltype(mid_tag)::value;
          // upper neighbour (j-1, i) = band column b+1 of the previous row: prevq of lane b+1, + 1 for the column, + 1 for the step
          // (lane 63 has no lane 64 to read: it gets 0 -- the band is 2k + 3 <= 63 columns, lane 63 is never inside it)
          const int uq = __builtin_amdgcn_update_dpp(0, prevq, 0x130, 0xF, 0xF, true) + 2;  // wave_shl:1
          int dgq;
          if constexpr (IUPAC) dgq = prevq + ((pc & tc) == 0u ? 1 : 0);  // (pattern letters are base sets <= 15: no mask)
          else dgq = prevq + (((pc ^ tc) & rule.emask) != 0u ? 1 : 0);
          int xq = dgq < uq ? dgq : uq;
          bool valid = true;
          if constexpr (!MID) {
            const uint32_t d = (uint32_t)(j - lo_j);
            const int firstq = (j < inf ? j : inf) + dlo_u + j;  // D[j][0] = j (capped) minus the band column that holds it: a scalar
            xq = d == 0u ? firstq : xq;
            valid = d < cnt;
            xq = valid ? xq : kBig;
          }
          xq = min(xq, __builtin_amdgcn_update_dpp(kId, xq, 0x111, 0xF, 0xF, false));  // row_shr:1
          xq = min(xq, __builtin_amdgcn_update_dpp(kId, xq, 0x112, 0xF, 0xF, false));  // row_shr:2
          xq = min(xq, __builtin_amdgcn_update_dpp(kId, xq, 0x114, 0xF, 0xF, false));  // row_shr:4
          xq = min(xq, __builtin_amdgcn_update_dpp(kId, xq, 0x118, 0xF, 0xF, false));  // row_shr:8
          if constexpr (WIDTH > 16) xq = min(xq, __builtin_amdgcn_update_dpp(kId, xq, 0x142, 0xA, 0xF, false));  // row_bcast:15
          if constexpr (WIDTH > 32) xq = min(xq, __builtin_amdgcn_update_dpp(kId, xq, 0x143, 0xC, 0xF, false));  // row_bcast:31
          int vq;
          if constexpr (MID) {
            vq = max(min(xq, infq), loq);  // (one v_med3_i32) min(xq, infq) inside the band, infq outside
          } else {
            xq = xq < infq ? xq : infq;
            vq = valid ? xq : infq;
          }
          *Lrow = (Cell)(vq + b);
          Lrow += Lstride;
          prevq = vq;
        };
        // rows j0 .. j1, two per iteration, each with its own pair of prefetch registers (loaded a row ahead; a read behind
        // the pattern's last row or the window stays inside the wave's slices and is not used)
        auto run = [&](int j0, int j1, auto mid_tag) {
          if (j0 > j1) return;
          uint32_t pa = spat[j0 - 1], ta = wp[j0 - 1];
          int j = j0;
          for (; j < j1; j += 2) {
            const uint32_t pb = spat[j], tb = wp[j];
            row(j, pa, ta, mid_tag);
            pa = spat[j + 1];
            ta = wp[j + 1];
            row(j + 1, pb, tb, mid_tag);
          }
          if (j == j1) row(j, pa, ta, mid_tag);
        };
        // middle rows: 1 <= j + dlo (every band column behind the window's first column) and j + dlo + bw - 1 <= iend
        const int iend_u = __builtin_amdgcn_readfirstlane(iend);
        int mid0 = 1 - dlo_u, mid1 = iend_u - dlo_u - bw + 1;
        mid0 = mid0 < 1 ? 1 : mid0;
        mid1 = mid1 > m ? m : mid1;
        if (mid0 > mid1) {  // (a window shorter than the band is wide)
          run(1, m, std::false_type{});
        } else {
          run(1, mid0 - 1, std::false_type{});
          run(mid0, mid1, std::true_type{});
          run(mid1 + 1, m, std::false_type{});
        }
      };
      const bool iu = rule.iupac != 0u;  // wave-uniform, like bw
      if (bw <= 16) { if (iu) rows(std::true_type{}, std::integral_constant<int, 16>{}); else rows(std::false_type{}, std::integral_constant<int, 16>{}); }
      else if (bw <= 32) { if (iu) rows(std::true_type{}, std::integral_constant<int, 32>{}); else rows(std::false_type{}, std::integral_constant<int, 32>{}); }
      else { if (iu) rows(std::true_type{}, std::integral_constant<int, 64>{}); else rows(std::false_type{}, std::integral_constant<int, 64>{}); }
    }
    // make the band and the window visible to every lane (same wave: LDS ops are in order, this
    // only keeps the compiler from reordering)
    __builtin_amdgcn_wave_barrier();
    tick(2);  // band fill
    // ---- greedy walk from (m, iend) ----
    // Every lane follows the same path, so the state lives in scalars (readfirstlane: the compiler cannot know that
    // the report is the same for the whole wave) and the branches are scalar branches.  Most steps are '=' along a
    // diagonal: lane t tests the step t cells further up, a ballot gives the length of the run and the wave takes it
    // in one round -- about 2k + 1 rounds of one LDS round trip each instead of m + k dependent steps.
    int j = __builtin_amdgcn_readfirstlane(m), i = __builtin_amdgcn_readfirstlane(iend);
    const int dlo_s = __builtin_amdgcn_readfirstlane(dlo);
    int g = __builtin_amdgcn_readfirstlane((int)L[(size_t)m * bw + (k + 1)]);
    int cost = g;
    uint32_t pattern_start = 0, pattern_end = P.m;
    const uint32_t max_ops = P.m + P.k + 1;
    uint32_t nops = 0;
    bool ok = g <= k;
    if (ok && i > wl) {  // the match ends past the text: step back along the diagonal (trace.rs:299-312)
      const int over = i - wl;
      if (over > m) ok = false;
      else {
        pattern_end -= (uint32_t)over;
        cost += __float2int_rd((float)over * P.alpha);
        i -= over;
        j -= over;
      }
    }
    while (ok && j > 0) {
      if (alpha_on && i == 0) {  // the rest of the pattern hangs over the text start (trace.rs:322-335)
        pattern_start = (uint32_t)j;
        g -= ov_left(P, j);
        break;
      }
      if (nops >= max_ops) { ok = false; break; }
      const int bb = i - j - dlo_s;
      const int t = (int)lane;
      const bool can = t < j && t < i;  // lane 0: i > 0
      const int tt = can ? t : 0;
      const int cell = can ? (int)L[(size_t)(j - 1 - tt) * bw + bb] : -1;
      const uint32_t pc = spat[j - 1 - tt], tc = win[i > 0 ? i - 1 - tt : 0];
      // the other two neighbours of (j, i), for the step that ends the run (same LDS round trip)
      const int lft_v = (i > 0 && bb > 0) ? (int)L[(size_t)j * bw + bb - 1] : inf;
      const int up_v = (bb + 1 < bw) ? (int)L[(size_t)(j - 1) * bw + bb + 1] : inf;
      const bool hit = can && cell == g && rule_hit(rule, pc, tc, rule.mmask);
      const unsigned long long bal = __ballot(hit);
      const uint32_t run = bal == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~bal);
      if (run) {
        if (nops + run > max_ops) { ok = false; break; }
        if (lane < run) ops[nops + lane] = '=';
        nops += run; j -= (int)run; i -= (int)run;
        continue;
      }
      g -= 1;
      if (g < 0) { ok = false; break; }
      const int diag = i > 0 ? __builtin_amdgcn_readfirstlane(cell) : inf;
      unsigned char op;
      if (diag == g) { op = 'X'; --j; --i; }
      else if (__builtin_amdgcn_readfirstlane(lft_v) == g) { op = 'D'; --i; }
      else if (__builtin_amdgcn_readfirstlane(up_v) == g) { op = 'I'; --j; }
      else { ok = false; break; }  // the reference panics here ("Trace failed! No ancestor found")
      if (lane == 0) ops[nops] = op;
      ++nops;
    }
    if (ok && g != 0) ok = false;
    if (cost > cd.cost) ok = false;  // src/search.rs:1672-1685
    __builtin_amdgcn_wave_barrier();
    tick(3);  // walk
    // ---- cigar text: run-length encoded, start -> end (the ops were recorded end -> start), by the whole wave:
    // run starts by ballot into the band's memory (the walk is done with it), then one lane per run ----
    unsigned char* sbuf = ops + P.ops_bytes;
    uint32_t w = 0;
    if (ok && nops) {
      uint16_t* starts = reinterpret_cast<uint16_t*>(slice);
      uint32_t nruns = 0;
      for (uint32_t x0 = 0; x0 < nops; x0 += 64u) {
        const uint32_t x = x0 + lane;
        const bool valid = x < nops;
        const uint32_t cur = valid ? ops[nops - 1u - x] : 0u;
        const uint32_t prv = (valid && x > 0) ? ops[nops - x] : 0x100u;
        const bool is_start = valid && cur != prv;
        const unsigned long long sb = __ballot(is_start);
        if (is_start) starts[nruns + (uint32_t)__popcll(sb & ((1ull << lane) - 1ull))] = (uint16_t)x;
        nruns += (uint32_t)__popcll(sb);
      }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t r0 = 0; r0 < nruns; r0 += 64u) {
        const uint32_t r = r0 + lane;
        const bool valid = r < nruns;
        const uint32_t s0 = valid ? starts[r] : 0u;
        const uint32_t e0 = valid ? (r + 1u < nruns ? (uint32_t)starts[r + 1u] : nops) : 0u;
        uint32_t len = e0 - s0;
        const unsigned char op = valid ? ops[nops - 1u - s0] : (unsigned char)0;
        const uint32_t nd = len >= 10000u ? 5u : len >= 1000u ? 4u : len >= 100u ? 3u : len >= 10u ? 2u : 1u;
        const uint32_t tl = valid ? nd + 1u : 0u;
        uint32_t incl = tl;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t v = __shfl_up(incl, d);
          if ((int)lane >= d) incl += v;
        }
        const uint32_t off = w + incl - tl;
        if (valid) {
          for (int q = (int)nd - 1; q >= 0; --q) { sbuf[off + q] = (unsigned char)('0' + len % 10u); len /= 10u; }
          sbuf[off + nd] = op;
        }
        w += __builtin_amdgcn_readlane(incl, 63);
      }
    }
    if (lane == 0) { sbuf[w] = 0; sbuf[w + 1] = 0; sbuf[w + 2] = 0; sbuf[w + 3] = 0; }
    __builtin_amdgcn_wave_barrier();
    {
      const uint32_t ndw = w / 4 + 1;
      const uint32_t* ssrc = reinterpret_cast<const uint32_t*>(sbuf);
      uint32_t* dstr = reinterpret_cast<uint32_t*>(P.out_str + (uint64_t)c * P.str_stride);
      uint32_t* hstr = reinterpret_cast<uint32_t*>(P.host_str + (uint64_t)c * P.str_stride);
      for (uint32_t x = lane; x < ndw; x += 64) {
        const uint32_t v = ssrc[x];
        dstr[x] = v;
        if (c < P.host_cap) hstr[x] = v;
      }
      if (lane == 0) {
        const MatchOut r = make_row(P, c, W, o + (uint64_t)i, cost, w, ok, pattern_start, pattern_end, pattern_idx);
        P.out[c] = r;
        if (c < P.host_cap) P.host_out[c] = r;
        if (!ok && P.host_flags) __hip_atomic_fetch_or(P.host_flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __builtin_amdgcn_wave_barrier();  // the slice is reused by the next report
    tick(4);  // cigar text + rows out
    if (P.probe && lane == 0) P.probe[(size_t)(blockIdx.x * 4 + wave) * 8 + 7] += 1ull;
  }
}

template <typename Cell, bool IN_LDS, int KT>
static void launch_one(const TraceParams& P, uint32_t nblocks, size_t lds, hipStream_t stream) {
  if (IN_LDS) {
    static DeviceOnce attr_set;  // per instantiation
    if (attr_set.need()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_kernel<Cell, IN_LDS, KT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_set.done();
    }
  }
  hipLaunchKernelGGL((trace_kernel<Cell, IN_LDS, KT>), dim3(nblocks), dim3(64), IN_LDS ? lds : 0, stream, P);
}

hipError_t launch_trace(const TraceParams& P, uint32_t nblocks, hipStream_t stream) {
  if (P.wave_mode) {  // one wavefront per report (host checked 2k+3 <= 64 and the LDS budget)
    const size_t lds = (size_t)4 * ((P.m + 15u) & ~15u) + (size_t)4 * P.scratch_stride + (size_t)P.rank_lds * 8u;
    if (P.k + 1 <= 255) {
      static DeviceOnce attr8;
      if (attr8.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_wave_kernel<uint8_t, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_wave_kernel<uint8_t, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr8.done();
      }
      if (P.use_alpha) hipLaunchKernelGGL((trace_wave_kernel<uint8_t, true>), dim3(nblocks), dim3(256), lds, stream, P);
      else hipLaunchKernelGGL((trace_wave_kernel<uint8_t, false>), dim3(nblocks), dim3(256), lds, stream, P);
    } else {
      static DeviceOnce attr16;
      if (attr16.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_wave_kernel<uint16_t, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_wave_kernel<uint16_t, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr16.done();
      }
      if (P.use_alpha) hipLaunchKernelGGL((trace_wave_kernel<uint16_t, true>), dim3(nblocks), dim3(256), lds, stream, P);
      else hipLaunchKernelGGL((trace_wave_kernel<uint16_t, false>), dim3(nblocks), dim3(256), lds, stream, P);
    }
    return hipGetLastError();
  }
  // LDS mode: 64 slices + the pattern
  const size_t lds = (size_t)64 * P.scratch_stride + ((P.m + 15) / 16) * 16;
  const bool in_lds = lds <= kTraceLdsLimit;
  if (P.k + 1 > 255) {
    if (in_lds) launch_one<uint16_t, true, -1>(P, nblocks, lds, stream);
    else launch_one<uint16_t, false, -1>(P, nblocks, lds, stream);
  } else if (!in_lds) {
    launch_one<uint8_t, false, -1>(P, nblocks, lds, stream);
  } else if (P.use_alpha) {  // overhang lives in the generic variant only
    launch_one<uint8_t, true, -1>(P, nblocks, lds, stream);
  } else {
    switch (P.k) {
      case 0: launch_one<uint8_t, true, 0>(P, nblocks, lds, stream); break;
      case 1: launch_one<uint8_t, true, 1>(P, nblocks, lds, stream); break;
      case 2: launch_one<uint8_t, true, 2>(P, nblocks, lds, stream); break;
      case 3: launch_one<uint8_t, true, 3>(P, nblocks, lds, stream); break;
      case 4: launch_one<uint8_t, true, 4>(P, nblocks, lds, stream); break;
      case 5: launch_one<uint8_t, true, 5>(P, nblocks, lds, stream); break;
      case 6: launch_one<uint8_t, true, 6>(P, nblocks, lds, stream); break;
      default: launch_one<uint8_t, true, -1>(P, nblocks, lds, stream); break;
    }
  }
  return hipGetLastError();
}

}  // namespace sassy_hip
