// tiled_kernel.hip -- K2: the pattern-tiled scan of search_encoded_patterns (reference "v2":
// src/pattern_tiling/search.rs:148-175 myers_step, :326-425 search_ranges, tqueries.rs:53-134 peq tables).
//
// The reference's v2 turns the DP around: bits run along the PATTERN (<= 64 rows in one word), one pattern per
// SIMD lane, every lane consumes the same text character per step.  On a wavefront that is: lane = one pattern
// (64 per wave), the character class of the text byte is wave-uniform, the lane fetches ITS pattern's match mask
// for that class (peq) from LDS and advances the classic Myers column step; the last-row cost is tracked per
// lane and every end position with cost <= k is appended as a candidate {pattern, position, cost}.
//
// Where the reference walks the whole text with each block of LANES patterns, the grid here is
// (text chunks) x (groups of 64 patterns): a wave starts m + k characters left of its chunk with the fresh
// column (Vp = 1^m, cost = m) -- after m + k characters every value <= k is exact -- and owns
// the end positions inside its chunk.  Runs of positions <= k are complete in the candidate list whatever the
// chunking (chunks own disjoint position ranges), so the host applies the report rule to each run exactly, with
// no seam bookkeeping (host.hip: search_encoded_tiled).
//
// Used where the pigeonhole prefilter path (filter_dna_multi_kernel: one pass per 64 patterns at HBM speed plus
// chains per pattern) does not apply or is launch-bound: many patterns on short / medium texts, texts with
// non-ACGT letters.  Cost: ~30 VALU per (character, 64 patterns) -- integer-VALU bound, no MFMA.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "tiled_step.h"

namespace sassy_hip {

// IUPAC letter (5 low bits) -> base set nibble; non-letters act as N (reference v1 scan semantics,
// src/profiles/iupac.rs:281-330), X = 0.
__constant__ uint8_t kTiledIupacNib[32] = {
    15, 1, 14, 2, 13, 15, 15, 8, 7, 15, 15, 12, 15, 3, 15, 15,
    15, 15, 9, 10, 4, 4, 11, 5, 0, 6, 15, 15, 15, 15, 15, 15};

__device__ __forceinline__ void tiled_emit(const TiledParams& P, uint64_t pos, int cost, uint32_t pat) {
  if (P.keep_bits && !((P.keep_bits[pos >> 5] >> (pos & 31u)) & 1u)) return;  // (a gathered buffer: context, separators)
  // (saturating: far beyond any capacity the host would retry with, the lanes stop counting -- the counter never wraps)
  if (*reinterpret_cast<volatile const uint32_t*>(P.cand_count) > P.cand_stop) return;
  const uint32_t idx = atomicAdd(P.cand_count, 1u);
  if (idx < P.cand_cap) P.cand[idx] = Candidate{pos, cost, pat << kCandTextShift};
}

// The same for eight consecutive end positions of every lane of the wave at once (called in wave-uniform control flow):
// bit i of `mask` = position pos + i of this lane's pattern is listed, with cost c8[i].  ONE counter update per call: in
// the inside of a run of N every pattern matches at every position -- 512 records per wave and step, and an atomic each
// (plus the saturation test's read of the counter) made the zones around the N runs of a genome the longest kernel of a
// guide-set search (119 of 210 ms).
// cand_chunk != 0 (the zones around the N runs of a genome: 10^8 records from a few MB of text): the wave takes its list
// slots cand_chunk at a time and keeps the range in `cur` -- one update of the one counter per cand_chunk records
// instead of one per call (2.7 M updates of one address were 20 of the scan's 34 ms).  What a wave has left of its
// last range when it ends it fills with empty records (position 0: tiled_release), so the list has holes and
// *cand_count counts slots.
struct EmitCursor {
  uint32_t cur = 0, end = 0;  // wave-uniform
  bool dead = false;          // the list is far beyond any capacity: stop
};
__device__ __forceinline__ void tiled_emit8(const TiledParams& P, uint64_t pos, uint32_t mask, const int (&c8)[8], uint32_t pat,
                                            EmitCursor& C) {
  if (P.keep_bits) {  // (a gathered buffer: context, separators)
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
      const uint64_t q = pos + i;
      if (((mask >> i) & 1u) && !((P.keep_bits[q >> 5] >> (q & 31u)) & 1u)) mask &= ~(1u << i);
    }
  }
  const uint32_t n = (uint32_t)__popc(mask), lane = __lane_id();
  uint32_t inc = n;  // inclusive prefix sum over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += t;
  }
  const uint32_t total = __shfl(inc, 63, 64);
  if (total == 0) return;
  uint32_t idx, wrap_at = 0xFFFFFFFFu, wrap_to = 0;  // slots from wrap_at on continue at wrap_to
  if (P.cand_chunk) {
    if (C.dead) return;
    idx = C.cur + inc - n;
    if (C.cur + total > C.end) {  // (wave-uniform) the next range; total <= 512 <= cand_chunk
      uint32_t nf = 0;
      if (lane == 0) nf = atomicAdd(P.cand_count, P.cand_chunk);
      nf = __shfl(nf, 0, 64);
      if (nf > P.cand_stop) { C.dead = true; return; }
      wrap_at = C.end;
      wrap_to = nf;
      C.cur = nf + (C.cur + total - C.end);
      C.end = nf + P.cand_chunk;
    } else {
      C.cur += total;
    }
  } else {
    uint32_t first = 0xFFFFFFFFu;
    if (lane == 0) {
      // (saturating: far beyond any capacity the host would retry with, the waves stop counting -- the counter never wraps)
      if (*reinterpret_cast<volatile const uint32_t*>(P.cand_count) <= P.cand_stop) first = atomicAdd(P.cand_count, total);
    }
    first = __shfl(first, 0, 64);
    if (first == 0xFFFFFFFFu) return;
    idx = first + inc - n;
  }
#pragma unroll
  for (uint32_t i = 0; i < 8; ++i) {
    if ((mask >> i) & 1u) {
      const uint32_t slot = idx >= wrap_at ? wrap_to + (idx - wrap_at) : idx;
      if (slot < P.cand_cap) P.cand[slot] = Candidate{pos + i, c8[i], pat << kCandTextShift};
      ++idx;
    }
  }
}
// the end of a wave that took its slots in ranges: empty records in what is left of the last one
__device__ __forceinline__ void tiled_release(const TiledParams& P, const EmitCursor& C) {
  for (uint32_t i = C.cur + __lane_id(); i < C.end && i < P.cand_cap; i += 64u) P.cand[i] = Candidate{0ull, 0, 0u};
}

// WORDS = 1: patterns of <= 32 rows (one 32-bit word), 2: 33 .. 64 rows.
//
// Coordinates: y = text position + skew indexes the 64-byte-aligned array text_aligned = text - skew, so that
// every wave step loads one aligned 64-byte block (one byte per lane); the chunks a wave owns are whole blocks
// of y.  Character y is followed by end position y - skew + 1.
template <int WORDS>
__global__ __launch_bounds__(256) void tiled_scan_kernel(const TiledParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsmem[];
  typedef typename std::conditional<WORDS == 1, uint32_t, unsigned long long>::type Word;
  constexpr uint32_t kShift = WORDS == 1 ? 8u : 9u;  // one class = 64 lanes x sizeof(Word) bytes of LDS
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t w = (uint64_t)blockIdx.x * kWavesPerGroup + wave;
  const uint64_t chunk = w / P.n_groups;
  const uint32_t group = (uint32_t)(w % P.n_groups);
  if (chunk >= P.n_chunks) return;  // wave-uniform
  const uint32_t pat = group * 64u + lane;
  const bool valid = pat < P.npat;
  // this wave's match masks: [class][lane]
  unsigned char* wave_lds = tsmem + ((size_t)wave * P.classes << kShift);
  Word* peq = reinterpret_cast<Word*>(wave_lds);
  for (uint32_t c = 0; c < P.classes; ++c) {
    const unsigned long long v = valid ? P.peq[(size_t)c * P.npat_padded + pat] : 0ull;
    peq[c * 64u + lane] = (Word)v;
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned char* lane_lds = wave_lds + lane * sizeof(Word);

  const uint32_t m = P.m;
  const int kk = valid ? (int)P.k : (int)0x80000000;  // lanes without a pattern never report
  const uint32_t skew = P.skew;
  const uint64_t end_y = (uint64_t)skew + P.text_len;
  const uint64_t own_lo = chunk * (uint64_t)P.chunk;
  const uint64_t own_hi = own_lo + P.chunk < end_y ? own_lo + P.chunk : end_y;
  const uint64_t warm = (uint64_t)P.warm_blocks * 64u;
  const uint64_t first = own_lo > warm ? own_lo - warm : 0;  // first block processed (0: the text starts in it)
  const uint32_t top_shift = (m - 1u) & 31u;
  TiledState<Word> S;
  S.vp = m >= 8 * sizeof(Word) ? (Word)~(Word)0 : (Word)(((Word)1 << m) - 1);
  S.vn = 0;
  S.cost = (int)m;
  if (chunk == 0 && S.cost <= kk) tiled_emit(P, 0ull, S.cost, pat);  // end position 0 (only when m <= k)
  EmitCursor cursor;

  for (uint64_t yb = first; yb < own_hi; yb += 64) {
    // 64 text bytes, one per lane -> LDS offsets of their character classes
    const uint32_t ch = P.text_aligned[yb + lane];
    uint32_t off;
    if (P.classes == 4) off = ((ch >> 1) & 3u) << kShift;   // Dna code (src/profiles/dna.rs:19-40)
    else off = (uint32_t)kTiledIupacNib[ch & 31u] << kShift;  // Iupac base set
    const uint32_t u0 = yb == 0 ? skew : 0u;
    const uint32_t u1 = end_y - yb < 64 ? (uint32_t)(end_y - yb) : 64u;
    const bool owned = yb >= own_lo;
    const uint64_t pos0 = yb - skew + 1;  // end position behind character u = 0 of this block
    if (u0 == 0 && u1 == 64) {
      // whole block: eight characters' masks are fetched ahead of the eight (dependent) column steps
#pragma unroll
      for (uint32_t g = 0; g < 64; g += 8) {
        Word eq[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i)
          eq[i] = *reinterpret_cast<const Word*>(lane_lds + (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(g + i)));
        if (owned) {
          // (one look per eight characters whether any of them ended a match: a compare and a branch per character were
          // a third of the loop's instructions)
          int c8[8];
          int lowest = 0x7FFFFFFF;
#pragma unroll
          for (uint32_t i = 0; i < 8; ++i) {
            tiled_step(S, eq[i], top_shift);
            c8[i] = S.cost;
            lowest = min(lowest, S.cost);
          }
          if (__any(lowest <= kk)) {  // (wave-uniform: the records of all lanes leave with one counter update)
            uint32_t mask = 0;
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) mask |= (c8[i] <= kk ? 1u : 0u) << i;
            tiled_emit8(P, pos0 + g, mask, c8, pat, cursor);
          }
        } else {  // warm-up: nothing is reported
#pragma unroll
          for (uint32_t i = 0; i < 8; ++i) tiled_step(S, eq[i], top_shift);
        }
      }
    } else {  // the block the text starts or ends in
      for (uint32_t u = u0; u < u1; ++u) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)u);
        tiled_step(S, *reinterpret_cast<const Word*>(lane_lds + o), top_shift);
        if (owned && S.cost <= kk) tiled_emit(P, pos0 + u, S.cost, pat);
      }
    }
  }
  if (P.cand_chunk) tiled_release(P, cursor);
}

// Overhang in one pass over a batch of texts (TiledParams::n_texts != 0): a wave = 64 patterns x a run of
// texts_per_wave texts, each text from its own overhang column to its last virtual 'N' column (the virtual columns are
// made here -- class 15 -- whatever the buffer holds behind a text).
// edge_cols != 0: ONLY what overhang changes -- the seeded search lists the end positions (edge_cols, len] of a text
// longer than edge_cols = m + k (no alignment with <= k edits that ends there reaches the text's first column, and the
// virtual columns lie behind them): this kernel adds [0, edge_cols] from the overhang column and (len, len + ov_steps]
// from a fresh column m + k characters in front of the text's end.
template <int WORDS>
__global__ __launch_bounds__(256) void tiled_pertext_kernel(const TiledParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsmem[];
  typedef typename std::conditional<WORDS == 1, uint32_t, unsigned long long>::type Word;
  constexpr uint32_t kShift = WORDS == 1 ? 8u : 9u;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t w = (uint64_t)blockIdx.x * kWavesPerGroup + wave;
  const uint64_t range = w / P.n_groups;
  const uint32_t group = (uint32_t)(w % P.n_groups);
  const uint64_t t_lo = range * P.texts_per_wave;
  if (t_lo >= P.n_texts) return;  // wave-uniform
  const uint64_t t_hi = t_lo + P.texts_per_wave < P.n_texts ? t_lo + P.texts_per_wave : P.n_texts;
  const uint32_t pat = group * 64u + lane;
  const bool valid = pat < P.npat;
  unsigned char* wave_lds = tsmem + ((size_t)wave * P.classes << kShift);
  Word* peq = reinterpret_cast<Word*>(wave_lds);
  for (uint32_t c = 0; c < P.classes; ++c) {
    const unsigned long long v = valid ? P.peq[(size_t)c * P.npat_padded + pat] : 0ull;
    peq[c * 64u + lane] = (Word)v;
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned char* lane_lds = wave_lds + lane * sizeof(Word);
  const int kk = valid ? (int)P.k : (int)0x80000000;  // lanes without a pattern never report
  const uint32_t top_shift = (P.m - 1u) & 31u;
  const uint32_t n_class = (P.classes == 4 ? 3u : 15u) << kShift;  // 'N' (overhang is Iupac's: 16 classes)
  EmitCursor cursor;  // (cand_chunk is 0 here: a counter update per call)
  for (uint64_t t = t_lo; t < t_hi; ++t) {
    const uint64_t start = P.texts_start[t], len = P.texts_len[t];
    if (len == 0) continue;  // no reports for an empty text (src/search.rs:1314-1316)
    const uint64_t end = len + P.ov_steps;
    const bool edges = P.edge_cols != 0 && len > P.edge_cols;
    // the segments of this text: characters [c0, c1) (virtual ones included), end positions from emit_from on
    for (int seg = 0; seg < (edges ? 2 : 1); ++seg) {
      const uint64_t c0 = seg == 0 ? 0 : len - P.edge_cols;
      const uint64_t c1 = edges && seg == 0 ? P.edge_cols : end;
      const uint64_t emit_from = seg == 0 ? 0 : len + 1;
      TiledState<Word> S;
      if (c0 == 0) {  // the overhang column
        S.vp = (Word)P.ov_vp;
        S.cost = P.ov_cost0;
        if (S.cost <= kk) tiled_emit(P, start, S.cost, pat);  // end position 0: the whole pattern hangs over the text's start
      } else {        // a fresh column inside the text
        S.vp = P.m >= 8 * sizeof(Word) ? (Word)~(Word)0 : (Word)(((Word)1 << P.m) - 1);
        S.cost = (int)P.m;
      }
      S.vn = 0;
      for (uint64_t yb = c0 & ~63ull; yb < c1; yb += 64) {
        // (a batch slot is a whole number of blocks that covers len + ov_steps + 2 characters; a single text -- search_encoded
        // with overhang -- ends where its buffer ends: nothing is read behind text_len, the virtual columns are made below)
        const uint64_t at = start + yb + lane;
        const uint32_t ch = at < P.text_len ? P.text_aligned[at] : (uint32_t)'N';
        uint32_t off;
        if (P.classes == 4) off = ((ch >> 1) & 3u) << kShift;
        else off = (uint32_t)kTiledIupacNib[ch & 31u] << kShift;
        const uint64_t pos0 = start + yb + 1;  // end position behind character u = 0 of this block
        if (yb >= c0 && yb + 64 <= c1 && yb + 64 <= len && yb + 1 >= emit_from) {  // a whole block inside the text, all of it listed
#pragma unroll
          for (uint32_t g = 0; g < 64; g += 8) {
            Word eq[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i)
              eq[i] = *reinterpret_cast<const Word*>(lane_lds + (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(g + i)));
            int c8[8];
            int lowest = 0x7FFFFFFF;
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
              tiled_step(S, eq[i], top_shift);
              c8[i] = S.cost;
              lowest = min(lowest, S.cost);
            }
            if (__any(lowest <= kk)) {
              uint32_t mask = 0;
#pragma unroll
              for (uint32_t i = 0; i < 8; ++i) mask |= (c8[i] <= kk ? 1u : 0u) << i;
              tiled_emit8(P, pos0 + g, mask, c8, pat, cursor);
            }
          }
        } else {  // a segment's first or last block, the block the text ends in, the virtual columns
          const uint32_t u0 = c0 > yb ? (uint32_t)(c0 - yb) : 0u;
          const uint32_t u1 = c1 - yb < 64 ? (uint32_t)(c1 - yb) : 64u;
          for (uint32_t u = u0; u < u1; ++u) {
            const uint64_t i = yb + u + 1;  // the end position behind this character
            const uint32_t o = i <= len ? (uint32_t)__builtin_amdgcn_readlane((int)off, (int)u) : n_class;
            tiled_step(S, *reinterpret_cast<const Word*>(lane_lds + o), top_shift);
            if (i >= emit_from) {
              // (f32 arithmetic, as the reference's add_overshoot_cost: src/search.rs:1274-1282)
              const int tot = S.cost + (i > len ? __float2int_rd(P.alpha * (float)(i - len)) : 0);
              if (tot <= kk) tiled_emit(P, start + i, tot, pat);
            }
          }
        }
      }
    }
  }
}

hipError_t launch_tiled_pertext(const TiledParams& P, hipStream_t stream) {
  if (P.n_texts == 0 || P.texts_per_wave == 0) return hipErrorInvalidValue;
  const uint64_t ranges = ((uint64_t)P.n_texts + P.texts_per_wave - 1) / P.texts_per_wave;
  const uint64_t waves = ranges * (uint64_t)P.n_groups;
  const uint64_t groups = (waves + kWavesPerGroup - 1) / kWavesPerGroup;
  if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
  if (P.m <= 32) {
    const size_t lds = (size_t)kWavesPerGroup * P.classes * 64u * 4u;
    hipLaunchKernelGGL((tiled_pertext_kernel<1>), dim3((uint32_t)groups), dim3(256), lds, stream, P);
  } else {
    const size_t lds = (size_t)kWavesPerGroup * P.classes * 64u * 8u;
    hipLaunchKernelGGL((tiled_pertext_kernel<2>), dim3((uint32_t)groups), dim3(256), lds, stream, P);
  }
  return hipGetLastError();
}

hipError_t launch_tiled_scan(const TiledParams& P, hipStream_t stream) {
  const uint64_t waves = P.n_chunks * (uint64_t)P.n_groups;
  const uint64_t groups = (waves + kWavesPerGroup - 1) / kWavesPerGroup;
  if (groups == 0) return hipSuccess;
  if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
  if (P.m <= 32) {
    const size_t lds = (size_t)kWavesPerGroup * P.classes * 64u * 4u;
    hipLaunchKernelGGL((tiled_scan_kernel<1>), dim3((uint32_t)groups), dim3(256), lds, stream, P);
  } else {
    const size_t lds = (size_t)kWavesPerGroup * P.classes * 64u * 8u;
    hipLaunchKernelGGL((tiled_scan_kernel<2>), dim3((uint32_t)groups), dim3(256), lds, stream, P);
  }
  return hipGetLastError();
}

}  // namespace sassy_hip
