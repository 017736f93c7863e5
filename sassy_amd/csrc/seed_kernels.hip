// seed_kernels.hip -- search_encoded_patterns for MANY patterns over a LONG text (BASELINE config 4: 10 000
// 20-mers, k = 2, 3 GB): seed, verify, report.  No pass over the text per pattern (or per 64 patterns), no kernel
// launch per pattern.
//
// Exactness (the pigeonhole argument the per-pattern prefilter uses, src of the idea: every alignment with <= k
// edits leaves one of k+1 disjoint pattern pieces intact): cut every pattern into k+1 pieces; an end position e
// with cost <= k has an alignment in which some piece p matches the text exactly, ending at a text position i with
// e in [i + rem_p - k, i + rem_p + k] (rem_p = pattern rows behind the piece).  So
//   seed     (seed_search_kernel, the pass over the text): ONE pass.  Every lane packs its characters to 2-bit Dna codes,
//            forms the L-gram ending at each position and looks it up in a direct-address table of all pieces of all
//            patterns (4^L entries -> list of (pattern, piece)); every hit goes into the wave's queue in LDS as
//            (position, index of the table entry) -- a tight loop that loads nothing;
//   test     (same kernel, whenever 64 hits are queued): the sub-piece test (test_issue / test_finish below): one of
//            k+1 disjoint sub-pieces of the pattern's other rows must be intact within k characters of the seed's
//            diagonal, or the hit is a chance hit (96 % of them).  Two independent loads per hit, requested a batch
//            ahead of the comparison;
//   verify   (same kernel, whenever 64 hits have passed): one lane per hit runs the pattern's Myers column steps (bits
//            along the pattern, tiled_step.h) over the m + 3k + 1 characters in front of the last end position the
//            hit allows, from the fresh column -- exact after m + k characters -- and appends every end position
//            with cost <= k in its range to the (pattern, position, cost) list.  (A first version wrote the hits to
//            a global list and verified them in a second kernel: 176 GB of traffic and 8 GB of memory at config 4
//            for nothing -- the hits of a wave are consumed where they are found, the text around them still in L2.)
//   the list holds ALL positions with cost <= k (some twice: a match seen through two pieces), which is what the
//   report rule needs: sort, drop duplicates, flag the reports per run (sort_kernels.hip), trace them
//   (trace_wave_kernel) -- the tail of the pattern-tiled search (host.hip: finish_pattern_list).
//
// Cost at config 4: 3.7 table hits per text position, 162 VALU per 64 hits -- the test 84, queueing 30,
// verification 26 (4 % of the hits pass), table look-ups 11, the passed queue 8 -- at 0.83 of the VALU issue rate; never
// to be priced against the 17 operations per (position, pattern) of the pattern-tiled scan: the kernel does not do
// that work.  The text is read once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "tiled_step.h"

namespace sassy_hip {

namespace {

// 16 text bytes -> 16 Dna codes ((c >> 1) & 3, src/profiles/dna.rs:19-40), 2 bits each, byte 0 lowest
__device__ __forceinline__ uint32_t pack_codes16(const uint4 v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t d[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    d[i] = __builtin_amdgcn_udot4((w[i] >> 1) & 0x03030303u, 0x40100401u, 0u, false);  // weights 1, 4, 16, 64
  return d[0] | (d[1] << 8) | (d[2] << 16) | (d[3] << 24);
}

// 2-bit copy of the text for the sub-piece test: dword x = characters [16 x, 16 x + 16)
__global__ __launch_bounds__(256) void pack_text_kernel(const uint4* __restrict__ text16, uint64_t n16,
                                                        uint32_t* __restrict__ packed) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    packed[i] = pack_codes16(text16[i]);
}

// two neighbouring table offsets, one load (one line access of the vector cache instead of two)
struct __attribute__((packed, aligned(4))) StartPair { uint32_t a, b; };

// The sub-piece test (common.h: SeedParams::sub): false = no alignment with <= k edits keeps this hit's piece intact.
// KT >= 0: k is the compile-time constant KT (the loops over sub-pieces and shifts unroll), KT < 0: any k.
//
// What bounds the seeded search is the latency and the number of the test's loads (a wave spent 60 % of its time
// waiting for them; the vector cache takes one line access per lane and load), so:
//   * the hit queue holds the INDEX of the hit's table entry (the pass over the text loads no entries);
//   * entries16[index] = (pattern << 3 | piece, the pattern's packed rows) in one 16-byte record (with care words: 32
//     bytes, the second half = 11 at the rows the test may compare), and ONE window of the packed text that starts
//     win_left characters in front of the seed's end, whatever the piece: two loads per hit (three with care words),
//     none of them waiting for another one;
//   * the test comes in two halves: test_issue requests the loads of the next 64 hits, test_finish compares the
//     batch before -- its loads were in flight while the batch before it was compared;
//   * rows: SeedParams::sub staged in LDS (8 dwords per piece).
// MODE 1 (narrow): every sub-piece starts within 32 characters of the window's start -- four dwords of text, the
// 64 bits a sub-piece is compared in start in dword 0 or 1; no care words; positions fit 32 bits.
// MODE 2 (wide): within 48 characters -- five dwords, 64 bits from dword 0, 1 or 2.  MODE 3: with care words.
// MODE 4: with care words, positions of any size.
// cand comes back as (position << kSeedPosShift) | pattern << 3 | piece, what the verification reads.
struct TestLoads {
  unsigned long long cand;
  uint4 entry;         // (pattern << 3 | piece, packed rows lo, hi, -)
  uint32_t care[2];
  uint32_t d[5];       // the text window's dwords
};
template <int MODE>
__device__ __forceinline__ TestLoads test_issue(const SeedParams& P, unsigned long long cand) {
  TestLoads L;
  L.cand = cand;
  const uint32_t idx = (uint32_t)cand & ((1u << kSeedPosShift) - 1u);
  const char* table = reinterpret_cast<const char*>(P.entries16);
  const uint32_t* src;
  if (MODE == 4) {
    const uint64_t i = cand >> kSeedPosShift;
    const uint64_t cl = i >= P.win_left ? i - P.win_left : 0u;
    src = P.packed_text + (cl >> 4);
  } else {
    const uint32_t i = (uint32_t)(cand >> kSeedPosShift);
    const uint32_t cl = i >= P.win_left ? i - P.win_left : 0u;
    src = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(P.packed_text) + ((cl >> 4) << 2));
  }
  if (MODE >= 3) {
    const uint4* e = reinterpret_cast<const uint4*>(table + ((uint64_t)idx << 5));
    L.entry = e[0];
    const uint2 c = *reinterpret_cast<const uint2*>(e + 1);
    L.care[0] = c.x; L.care[1] = c.y;
  } else {
    L.entry = *reinterpret_cast<const uint4*>(table + (idx << 4));
    L.care[0] = L.care[1] = 0xFFFFFFFFu;
  }
  L.d[0] = src[0]; L.d[1] = src[1]; L.d[2] = src[2]; L.d[3] = src[3];
  L.d[4] = MODE >= 2 ? src[4] : 0u;
  return L;
}
template <int KT, int MODE>
__device__ __forceinline__ bool test_finish(const SeedParams& P, const uint32_t* rows, const TestLoads& L, unsigned long long& cand) {
  bool inside;
  uint32_t sh;  // bits of the window's first character in d[0]
  if (MODE == 4) {
    const uint64_t i = L.cand >> kSeedPosShift;
    inside = i >= P.win_left;  // (beyond the text's end the packed copy holds zeros: a test that passes too often)
    sh = 2u * ((uint32_t)(i - P.win_left) & 15u);
  } else {
    const uint32_t i = (uint32_t)(L.cand >> kSeedPosShift);
    inside = i >= P.win_left;
    sh = 2u * ((i - P.win_left) & 15u);
  }
  const uint4 E = L.entry;
  cand = (L.cand & ~(unsigned long long)((1u << kSeedPosShift) - 1u)) | E.x;
  const uint4 r0 = *reinterpret_cast<const uint4*>(rows + 8u * (E.x & 7u));
  uint4 r1 = make_uint4(0, 0, 0, 0);
  if (KT < 0 || KT > 3) r1 = *reinterpret_cast<const uint4*>(rows + 8u * (E.x & 7u) + 4u);
  if ((r0.x & 0xFFu) == 0xFFu || !inside) return true;  // untested
  // the window's characters from its start on, 16 per dword
  uint32_t x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = (j < 3 || MODE >= 2) ? __builtin_amdgcn_alignbit(L.d[j + 1], L.d[j], sh) : 0u;
  const unsigned long long pp = ((unsigned long long)E.z << 32) | E.y;
  const unsigned long long care = ((unsigned long long)L.care[1] << 32) | L.care[0];
  const uint32_t ent[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  const uint32_t k = KT >= 0 ? (uint32_t)KT : P.k;
  uint32_t miss = 0xFFFFFFFFu;  // minimum over all (sub-piece, shift) of the differing bits: 0 = one of them is intact
#pragma unroll
  for (uint32_t u = 0; u < 8; ++u) {
    if (u <= k) {
      // row fields (host.hip): 2a | (32 - 2 len) << 8 | 2 (off & 15) << 16 | (off >> 4) << 24
      const uint32_t want = (uint32_t)(pp >> (ent[u] & 0xFFu));
      uint32_t mask = 0xFFFFFFFFu >> ((ent[u] >> 8) & 0xFFu);
      if (MODE >= 3) mask &= (uint32_t)(care >> (ent[u] & 0xFFu));
      // the window from the leftmost shift on; every further shift is two bits down
      const uint32_t which = ent[u] >> 24;
      uint32_t w0 = which ? x[1] : x[0], w1 = which ? x[2] : x[1];
      if (MODE >= 2) {
        w0 = which > 1u ? x[2] : w0;
        w1 = which > 1u ? x[3] : w1;
      }
      const unsigned long long w = (((unsigned long long)w1 << 32) | w0) >> ((ent[u] >> 16) & 0xFFu);
      const uint32_t wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
      if (KT >= 0) {
#pragma unroll
        for (uint32_t d = 0; d <= 2u * (uint32_t)(KT >= 0 ? KT : 0); ++d) {
          const uint32_t got = d == 0 ? wl : __builtin_amdgcn_alignbit(wh, wl, 2u * d);
          miss = min(miss, (got ^ want) & mask);
        }
      } else {
        for (uint32_t d = 0; d <= 2u * k; ++d) miss = min(miss, (__builtin_amdgcn_alignbit(wh, wl, 2u * d) ^ want) & mask);
      }
    }
  }
  return miss == 0u;
}

}  // namespace

// One lane per candidate.  EDGE: the candidate's window leaves the text (the first / last few dozen characters),
// or the buffer holds several texts with separators between them.
// (the launch parameters the verification reads, by value: it is called out of line -- its 64 unrolled column steps,
// inlined three times, kept 280 scalars of the seed pass spilled)
struct VerifyArgs {
  const uint8_t* text;
  uint64_t text_len;
  const void* peq;
  Candidate* out;
  uint32_t* out_count;
  uint64_t rem_packed;
  uint32_t mks;  // m | k << 8 | separators << 16  (64 bytes in all: a larger struct would travel through scratch memory)
  uint32_t out_cap, out_stop;
};
static_assert(sizeof(VerifyArgs) <= 64, "VerifyArgs must travel in registers");
template <int WORDS, bool EDGE>
__device__ __noinline__ void verify_candidate(const VerifyArgs P, unsigned long long cand) {
  typedef typename std::conditional<WORDS == 1, uint32_t, unsigned long long>::type Word;
  const uint32_t entry = (uint32_t)cand & ((1u << kSeedPosShift) - 1u);
  const uint32_t pat = entry >> 3, piece = entry & 7u;
  const int64_t i = (int64_t)(cand >> kSeedPosShift);
  // (the arguments arrive in vector registers: m, k and what follows from them are the same in every lane -- as scalars
  // the per-step tests "does this step report?" are scalar branches, not compares and exec masks)
  const uint32_t mks = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.mks);
  const int m = (int)(mks & 0xFFu), k = (int)((mks >> 8) & 0xFFu);
  const bool separators = (mks >> 16) != 0;
  const int T = m + 3 * k + 1;
  const int64_t e_hi = i + (int64_t)((P.rem_packed >> (8u * piece)) & 0xFFu) + k;  // last end position the seed allows
  const int64_t s0 = e_hi - T;                          // first character of the window
  // the pattern's match masks per Dna code
  Word e[4];
  if (WORDS == 1) {
    const uint4 v = reinterpret_cast<const uint4*>(P.peq)[pat];
    e[0] = (Word)v.x; e[1] = (Word)v.y; e[2] = (Word)v.z; e[3] = (Word)v.w;
  } else {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(P.peq) + 4ull * pat;
    e[0] = (Word)src[0]; e[1] = (Word)src[1]; e[2] = (Word)src[2]; e[3] = (Word)src[3];
  }
  // the window: kSeedWindowDwords dwords from the aligned address below s0, shifted so that byte 0 is character s0
  uint32_t win[kSeedWindowDwords];
  {
    int64_t a0 = s0 & ~(int64_t)3;
    const uint32_t off = (uint32_t)(s0 - a0);
    uint32_t raw[kSeedWindowDwords + 1];
#pragma unroll
    for (int x = 0; x <= kSeedWindowDwords; ++x) {
      const int64_t a = a0 + 4 * x;
      raw[x] = 0;
      if (4 * x < T + 4) {
        if (!EDGE) raw[x] = *reinterpret_cast<const uint32_t*>(P.text + a);
        else if (a >= 0 && a + 4 <= (int64_t)((P.text_len + 15) & ~15ull)) raw[x] = *reinterpret_cast<const uint32_t*>(P.text + a);
      }
    }
#pragma unroll
    for (int x = 0; x < kSeedWindowDwords; ++x) win[x] = __builtin_amdgcn_alignbyte(raw[x + 1], raw[x], off);
  }
  TiledState<Word> S;
  S.vp = m >= (int)(8 * sizeof(Word)) ? (Word)~(Word)0 : (Word)(((Word)1 << m) - 1);
  S.vn = 0;
  S.cost = m;
  const uint32_t top_shift = (uint32_t)(m - 1) & 31u;
  const int emit_from = T - (2 * k + 1);  // steps whose end position lies in the seed's range
  // The end positions this lane has to report stay in a register until the window is done: step emit_from + j is
  // nibble j (8 | cost; k <= 7, 2k + 1 <= 15 steps).  One counter update per wave and batch then -- an atomic per
  // step, with the saturation test's read of the counter in front of it, was two round trips to L2 per step with
  // all 64 lanes waiting (reads x barcodes, a true match in nearly every batch: 29 ms per 100 MB instead of 1.5).
  unsigned long long found = 0;
  bool done = false;
#pragma unroll
  for (int x = 0; x < kSeedWindowDwords; ++x) {
    if (4 * x < T && !done) {  // wave-uniform
      // The last row's cost falls by at most one per character: a lane whose cost cannot reach k by the last
      // step is done, and when that holds for every lane of the wave (nearly every candidate is a chance hit of
      // one piece: the cost hovers around m / 2) the rest of the window is skipped.
      if (4 * x >= 8 && __all(S.cost - (T - 4 * x) > k)) done = true;
      else {
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const int t = 4 * x + y;
          if (t < T) {
            // eq = e[code of the character]: two mux levels on the code's bits
            Word eq;
            if (WORDS == 1) {  // three v_bitop3 (s ? x1 : x0)
              const uint32_t s0 = __builtin_amdgcn_sbfe((int)win[x], 8 * y + 1, 1), s1 = __builtin_amdgcn_sbfe((int)win[x], 8 * y + 2, 1);
              const uint32_t lo = __builtin_amdgcn_bitop3_b32(s0, (uint32_t)e[0], (uint32_t)e[1], 0xAC);
              const uint32_t hi = __builtin_amdgcn_bitop3_b32(s0, (uint32_t)e[2], (uint32_t)e[3], 0xAC);
              eq = (Word)__builtin_amdgcn_bitop3_b32(s1, lo, hi, 0xAC);
            } else {
              const Word b0 = (Word)(long long)(int)__builtin_amdgcn_sbfe((int)win[x], 8 * y + 1, 1);  // (the builtin's type is unsigned)
              const Word b1 = (Word)(long long)(int)__builtin_amdgcn_sbfe((int)win[x], 8 * y + 2, 1);
              eq = (((e[0] & ~b0) | (e[1] & b0)) & ~b1) | (((e[2] & ~b0) | (e[3] & b0)) & b1);
            }
            if (EDGE) {
              const int64_t c = s0 + t;
              if (c < 0 || c >= (int64_t)P.text_len) eq = 0;  // outside the text: the fresh column stays fresh
              // multi-text buffers: the separator 'X' (and any text 'X': the empty IUPAC set) matches nothing
              if (separators && ((win[x] >> (8 * y + 3)) & 1u)) eq = 0;
            }
            tiled_step(S, eq, top_shift);
            if (t >= emit_from && S.cost <= k) {
              const int64_t pos = s0 + t + 1;
              if (!EDGE || (pos >= 1 && pos <= (int64_t)P.text_len))
                found |= (unsigned long long)(8u | (uint32_t)S.cost) << (4 * (t - emit_from));
            }
          }
        }
      }
    }
  }
  const unsigned long long who = __ballot(found != 0);
  if (who == 0) return;
  const uint32_t mine = (uint32_t)__popcll(found & 0x8888888888888888ull);
  uint32_t before = 0, total = 0;  // entries of the lanes in front of this one / of the wave
  const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  for (unsigned long long w = who; w; w &= w - 1) {
    const uint32_t l = (uint32_t)__builtin_ctzll(w);
    if (l == lane) before = total;
    total += (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)l);
  }
  const uint32_t leader = (uint32_t)__builtin_ctzll(who);
  uint32_t base = 0;
  if (lane == leader) {
    // (saturating counter, see TiledParams::cand_stop: beyond out_stop the update is taken back, so the counter
    // stays within a few waves' worth of out_stop and never wraps)
    base = atomicAdd(P.out_count, total);
    if (base > P.out_stop) atomicSub(P.out_count, total);
  }
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader) + before;
  for (unsigned long long f = found; f; ) {
    const uint32_t j = (uint32_t)__builtin_ctzll(f) >> 2;
    const uint32_t nib = (uint32_t)(f >> (4 * j)) & 15u;
    f &= ~(15ull << (4 * j));
    if (base < P.out_cap)
      P.out[base] = Candidate{(uint64_t)(e_hi - 2 * k + (int64_t)j), (int32_t)(nib & 7u), pat << kCandTextShift};
    ++base;
  }
}

// One wave walks a contiguous range of the text, 2 KiB per step: lane l takes the 32 characters [g, g + 32),
// g = step base + 32 l, plus the 16 in front of them (seeds that end in its characters start there).
// KT: the sub-piece test's k (>= 0: compile-time, -1: run-time, -2: no test); MODE: the test's layout (0: no test).
template <int WORDS, int KT, int MODE>
__global__ __launch_bounds__(256) void seed_search_kernel(const SeedParams P) {
  static_assert((KT < -1) == (MODE == 0), "a test needs a layout");
  constexpr bool TEST = MODE != 0;
  constexpr uint32_t kTurn = 4;
  // with a test the hit queue takes a position's hits of all lanes before any of them is tested
  constexpr uint32_t kRing = TEST ? 256 : 128;
  __shared__ unsigned long long queue_mem[kWavesPerGroup][kRing], pass_mem[kWavesPerGroup][128];
  // "does any pattern have a seed that ends like this?" -- one bit per min(len, 8)-gram and table (<= 2 x 8 KiB): with
  // few patterns nearly every position fails it and never reads the tables in global memory
  __shared__ uint32_t bits_lds[2 * 2048];
  for (uint32_t x = threadIdx.x; x < P.bits_off[1] + (P.len[1] ? (1u << (2 * (P.len[1] < 8 ? P.len[1] : 8))) / 32 : 0); x += blockDim.x)
    bits_lds[x] = P.seed_bits[x];
  __shared__ __attribute__((aligned(16))) uint32_t sub_rows[64];
  if (TEST && threadIdx.x < 64) sub_rows[threadIdx.x] = P.sub[threadIdx.x];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_in_group = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t wave = (uint64_t)blockIdx.x * kWavesPerGroup + wave_in_group;
  const uint64_t n_waves = (uint64_t)gridDim.x * kWavesPerGroup;
  const uint64_t steps = (P.text_len + 2047) / 2048;
  const uint64_t per_wave = (steps + n_waves - 1) / n_waves;
  const uint64_t s_lo = wave * per_wave;
  const uint64_t s_hi = s_lo + per_wave < steps ? s_lo + per_wave : steps;
  if (s_lo >= s_hi) return;
  unsigned long long* queue = queue_mem[wave_in_group];
  unsigned long long* passed = pass_mem[wave_in_group];
  // both queues are rings (of kRing and 128 entries): [head, head + count) modulo their size
  uint32_t queued = 0, n_passed = 0, q_head = 0, p_head = 0;  // wave-uniform
  uint64_t n_hits = 0, n_pass = 0;

  // verify the first `count` hits of `from`, one per lane (the window of a hit near the text's ends needs range checks)
  auto verify_from = [&](const unsigned long long* from, uint32_t head, uint32_t count, uint32_t ring = 128) __attribute__((always_inline)) {
    const bool have = lane < count;
    unsigned long long cand = 0;
    bool edge = false;
    if (have) {
      cand = from[(head + lane) & (ring - 1u)];
      const int64_t i = (int64_t)(cand >> kSeedPosShift);
      const int64_t e_hi = i + (int64_t)((P.rem_packed >> (8u * ((uint32_t)cand & 7u))) & 0xFFu) + (int64_t)P.k;
      const int64_t s0 = e_hi - ((int64_t)P.m + 3 * (int64_t)P.k + 1);
      edge = s0 < 4 || e_hi + 8 > (int64_t)P.text_len || P.separators != 0u;  // (separators: the checked copy of the loop)
    }
    VerifyArgs va;
    va.text = P.text; va.text_len = P.text_len; va.peq = P.peq; va.out = P.out; va.out_count = P.out_count;
    va.rem_packed = P.rem_packed; va.mks = P.m | (P.k << 8) | (P.separators ? 1u << 16 : 0u);
    va.out_cap = P.out_cap; va.out_stop = P.out_stop;
    if (__any(edge)) {
      if (have) verify_candidate<WORDS, true>(va, cand);
    } else {
      if (have) verify_candidate<WORDS, false>(va, cand);
    }
  };
  // the batch whose loads are in flight (test_issue) and its size; compared when the next one's loads have been issued
  TestLoads pend{};
  uint32_t pend_n = 0;  // wave-uniform
  auto passed_push = [&](bool ok, unsigned long long cand) __attribute__((always_inline)) {
    const unsigned long long m = __ballot(ok);
    if (ok)
      passed[(p_head + n_passed + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & 127u] = cand;
    n_passed += (uint32_t)__popcll(m);
    n_pass += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (n_passed >= 64) {
      verify_from(passed, p_head, 64);
      p_head = (p_head + 64u) & 127u;
      n_passed -= 64;
    }
  };
  auto verify_pending = [&]() __attribute__((always_inline)) {
    if (pend_n) {
      unsigned long long cand = 0;
      bool ok = false;
      if (lane < pend_n) ok = test_finish<KT, MODE>(P, sub_rows, pend, cand);
      passed_push(ok, cand);
    }
  };
  // The first `count` queued hits, one per lane: with the sub-piece test (WORDS == 1, P.sub) their loads are requested
  // and the batch before them is compared -- the few that pass collect in a second queue and are verified 64 at a
  // time --, else they are verified at once.
  auto verify = [&](uint32_t count) __attribute__((always_inline)) {
    if constexpr (!TEST) verify_from(queue, q_head, count, kRing);
    else {
      const TestLoads next = test_issue<MODE>(P, lane < count ? queue[(q_head + lane) & (kRing - 1u)] : 0ull);
      verify_pending();
      pend = next;
      pend_n = count;
    }
  };
  auto verify_drain = [&]() __attribute__((always_inline)) {
    if constexpr (TEST) {
      verify_pending();
      pend_n = 0;
    }
  };

  const uint32_t mask0 = P.len[0] ? (uint32_t)((1ull << (2 * P.len[0])) - 1) : 0u;
  const uint32_t mask1 = P.len[1] ? (uint32_t)((1ull << (2 * P.len[1])) - 1) : 0u;
  const uint32_t cut0 = P.len[0] > 8 ? 2 * (P.len[0] - 8) : 0u, cut1 = P.len[1] > 8 ? 2 * (P.len[1] - 8) : 0u;
  for (uint64_t s = s_lo; s < s_hi; ++s) {
    const uint64_t g = s * 2048 + 32ull * lane;
    // 48 characters as 96 bits of codes; bytes outside the text are never part of an accepted seed
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    if (g < P.text_len) {
      const uint4* src = reinterpret_cast<const uint4*>(P.text + g);
      if (g >= 16) c0 = pack_codes16(src[-1]);
      c1 = pack_codes16(src[0]);
      if (g + 16 < P.text_len) c2 = pack_codes16(src[1]);
    }
    const unsigned long long q01 = ((unsigned long long)c1 << 32) | c0, q12 = ((unsigned long long)c2 << 32) | c1;
    // table rows of position j: [first, last) of the entry lists (table 0: the longer pieces).  The loads are left
    // in flight: the caller subtracts when it gets to the position.
    auto look_up = [&](uint32_t j, uint32_t& a0, uint32_t& b0, uint32_t& a1, uint32_t& b1) __attribute__((always_inline)) {
      const unsigned long long q = j < 16 ? q01 : q12;
      const uint64_t end = g + j + 1;  // exclusive end of the seeds that end in character j
      const bool in_text = j < 32 && end <= P.text_len;
      a0 = b0 = a1 = b1 = 0;
      if (P.len[0] && in_text && end >= P.len[0]) {
        const uint32_t code = (uint32_t)(q >> (2u * (17u + (j & 15u) - P.len[0]))) & mask0;
        const uint32_t c8 = code >> cut0;  // the seed's last min(len, 8) characters
        if ((bits_lds[c8 >> 5] >> (c8 & 31u)) & 1u) {
          const StartPair ab = *reinterpret_cast<const StartPair*>(P.start[0] + code);
          a0 = ab.a;
          b0 = ab.b;
        }
      }
      if (P.len[1] && in_text && end >= P.len[1]) {
        const uint32_t code = (uint32_t)(q >> (2u * (17u + (j & 15u) - P.len[1]))) & mask1;
        const uint32_t c8 = code >> cut1;
        if ((bits_lds[P.bits_off[1] + (c8 >> 5)] >> (c8 & 31u)) & 1u) {
          const StartPair ab = *reinterpret_cast<const StartPair*>(P.start[1] + code);
          a1 = ab.a;
          b1 = ab.b;
        }
      }
    };
    uint32_t a0, b0, a1, b1;
    look_up(0, a0, b0, a1, b1);
#pragma unroll 1
    for (uint32_t j = 0; j < 32; ++j) {
      uint32_t xa0, xb0, xa1, xb1;  // the next position's rows are in flight while this one's hits are queued
      look_up(j + 1, xa0, xb0, xa1, xb1);
      const uint64_t end = g + j + 1;
      const uint32_t n0 = b0 - a0, n = n0 + (b1 - a1);
      if constexpr (TEST) {
        // The hits of this position: first all of them into the queue (one per lane and turn: the index of its table
        // entry, nothing to load), then the full batches out of it -- a tight loop in front of the one that carries
        // the batch in flight.
        const uint32_t rest = P.entries16_off1 + a1 - n0;  // hit r >= n0 is entry rest + r
        uint32_t r = 0;
        bool more = true;
        while (more) {
          for (;; ++r) {
            const bool active = r < n;
            const unsigned long long m = __ballot(active);
            more = m != 0;
            if (!more || queued > kRing - 64u) break;
            if (active) {
              const uint32_t slot = q_head + queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
              queue[slot & (kRing - 1u)] = ((unsigned long long)end << kSeedPosShift) | (r + (r < n0 ? a0 : rest));
            }
            queued += (uint32_t)__popcll(m);
          }
          __builtin_amdgcn_wave_barrier();
          while (queued >= 64) {
            verify(64);
            q_head = (q_head + 64u) & (kRing - 1u);
            queued -= 64;
            n_hits += 64;
          }
        }
      } else {
        // the hits of this position, kTurn per lane and turn: their entry loads are in flight together
        for (uint32_t r0 = 0; __ballot(r0 < n) != 0; r0 += kTurn) {
          uint32_t ent[kTurn];
#pragma unroll
          for (uint32_t x = 0; x < kTurn; ++x) {
            const uint32_t r = r0 + x;
            ent[x] = 0;
            if (r < n) ent[x] = r < n0 ? P.entries[0][a0 + r] : P.entries[1][a1 + (r - n0)];
          }
#pragma unroll
          for (uint32_t x = 0; x < kTurn; ++x) {
            const bool active = r0 + x < n;
            const unsigned long long m = __ballot(active);
            if (m == 0) break;
            if (active) {
              const uint32_t slot = q_head + queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
              queue[slot & (kRing - 1u)] = ((unsigned long long)end << kSeedPosShift) | ent[x];
            }
            queued += (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (queued >= 64) {
              verify(64);
              q_head = (q_head + 64u) & (kRing - 1u);
              queued -= 64;
              n_hits += 64;
            }
          }
        }
      }
      a0 = xa0; b0 = xb0; a1 = xa1; b1 = xb1;
    }
  }
  if (queued) verify(queued);
  verify_drain();
  n_hits += queued;
  if (n_passed) verify_from(passed, p_head, n_passed);
  if (P.hit_count && lane == 0 && n_hits) {
    atomicAdd(P.hit_count, (unsigned long long)n_hits);
    atomicAdd(P.hit_count + 1, (unsigned long long)n_pass);
  }
}

// ---------------------------------------------------------------- texts with other letters than ACGT (Iupac searcher)
// The seeded search reads Dna codes, so it is exact only where the m + k characters in front of an end position are
// plain.  Around every run of other letters ("dirty": N, R, Y, ..., non-letters) the end positions are computed by
// the pattern-tiled scan on a gathered copy of those neighbourhoods instead (host.hip: search_encoded_seeded):
//   dirty_scan_kernel      finds the runs: their first positions, their end positions, and the positions of dirty
//                          letters that are not full wildcards (a run of full wildcards -- N, non-letters -- longer
//                          than m + 1 is cut out of the gathered copy: every pattern's cost is constant inside);
//   gather_zones_kernel    copies the neighbourhoods ("zones") into one buffer, 'X' separators between them;
//   map_zone_list_kernel   turns the tiled scan's (pattern, position in the zone buffer, cost) records into text
//                          positions, keeps those a zone is responsible for and marks the last one in front of a
//                          cut-out stretch (kCandCont);
//   drop_excluded_kernel   flags the seeded search's records that lie where a zone is responsible.
namespace {
__device__ __forceinline__ bool dirty_byte(uint32_t c) {
  const uint32_t u = c & 0xDFu;
  return !(u == 'A' || u == 'C' || u == 'G' || u == 'T');
}
// dirty letters that do not match every base.  The Iupac profile looks a text byte up by its five low bits
// (src/profiles/iupac.rs:281-330): 15 = every base (N, and every byte whose low bits are no IUPAC letter's),
// anything else a proper subset -- also for bytes that are no letters ('-' reads as M).
__device__ __forceinline__ bool hard_byte(uint32_t c) {
  // bit i = the letter with low bits i stands for a proper subset: A B C D G H K M R S T U V W X Y
  constexpr uint32_t kProper = (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 7) | (1u << 8) | (1u << 11) | (1u << 13) |
                               (1u << 18) | (1u << 19) | (1u << 20) | (1u << 21) | (1u << 22) | (1u << 23) | (1u << 24) | (1u << 25);
  return (kProper >> (c & 31u)) & 1u;
}
}  // namespace

__global__ __launch_bounds__(256) void dirty_scan_kernel(const uint8_t* __restrict__ text, uint64_t n,
                                                         unsigned long long* __restrict__ starts,
                                                         unsigned long long* __restrict__ ends,
                                                         unsigned long long* __restrict__ hard, uint32_t cap,
                                                         uint32_t* __restrict__ counts) {
  const uint64_t n16 = (n + 15) / 16;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4*>(text)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t bad = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t sel = (w[d] >> 1) & 0x03030303u;
      bad |= (w[d] & 0xDFDFDFDFu) ^ __builtin_amdgcn_perm(0u, 0x47544341u, sel);  // (as acgt_check_kernel)
    }
    if (!bad) continue;  // sixteen plain characters (a run cannot start or end inside: their neighbours see to it)
    for (uint32_t j = 0; j < 16; ++j) {
      const uint64_t x = i * 16 + j;
      if (x >= n) break;
      const uint32_t c = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
      if (!dirty_byte(c)) continue;
      const bool prev_dirty = x > 0 && dirty_byte(text[x - 1]);
      const bool next_dirty = x + 1 < n && dirty_byte(text[x + 1]);
      if (!prev_dirty) { const uint32_t k = atomicAdd(counts + 0, 1u); if (k < cap) starts[k] = x; }
      if (!next_dirty) { const uint32_t k = atomicAdd(counts + 1, 1u); if (k < cap) ends[k] = x + 1; }
      if (hard_byte(c)) { const uint32_t k = atomicAdd(counts + 2, 1u); if (k < cap) hard[k] = x; }
    }
  }
}

// seg[4 g .. 4 g + 3] = source offset in the text, destination offset in the zone buffer, length, unused; the bytes
// the segments do not cover (the separators) are filled by the caller.
__global__ __launch_bounds__(256) void gather_zones_kernel(const uint8_t* __restrict__ text, uint8_t* __restrict__ dst,
                                                           const unsigned long long* __restrict__ seg, uint32_t n_seg) {
  for (uint32_t g = blockIdx.y; g < n_seg; g += gridDim.y) {
    const unsigned long long src = seg[4 * g], to = seg[4 * g + 1], len = seg[4 * g + 2];
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < len; x += (uint64_t)gridDim.x * blockDim.x)
      dst[to + x] = text[src + x];
  }
}

// zone[6 z .. 6 z + 5] = first byte of the zone in the zone buffer, text position of that byte, first and last end
// position (in the text) the zone is responsible for, 1 if a cut-out stretch follows the last one, unused.  Sorted by
// the first field.
constexpr uint32_t kMapTile = 2048;  // records per workgroup of map_zone_list_kernel
__global__ __launch_bounds__(256) void map_zone_list_kernel(const Candidate* __restrict__ in, uint32_t count,
                                                            const unsigned long long* __restrict__ zone, uint32_t n_zones,
                                                            Candidate* __restrict__ out, uint32_t* __restrict__ out_count,
                                                            uint32_t out_cap) {
  // A workgroup takes kMapTile records, 8 per thread, and updates the one counter ONCE (an update per wave -- 2 M of
  // them on one address -- was all of this kernel's 22 ms for the 10^8 records of a guide set on a text with N runs).
  // The list may have holes (position 0: the tiled scan took its slots in ranges).
  __shared__ uint32_t wave_sum[4];
  __shared__ uint32_t block_first;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long p8[8];
  uint32_t cf8[8];  // flags
  int cost8[8];
  uint32_t keep_bits = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t i = blockIdx.x * kMapTile + j * 256u + threadIdx.x;
    Candidate c{0, 0, 0};
    if (i < count) c = in[i];
    const bool have = i < count && c.pos != 0;
    // The zone whose bytes hold character c.pos - 1: the largest z with dst[z] <= c.pos - 1.  The records of a wave come
    // from one or two waves of the tiled scan -- one or two zones --, so the wave searches once, for its first record,
    // and a lane searches by itself only when its record lies elsewhere.
    uint32_t lo = 0;
    const unsigned long long active = __ballot(have);
    if (active) {
      const unsigned long long pos0 = __shfl(c.pos, __ffsll((long long)active) - 1, 64) - 1;
      uint32_t hi = n_zones;
      while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (zone[6 * mid] <= pos0) lo = mid; else hi = mid;
      }
    }
    bool keep = false;
    unsigned long long p = 0;
    uint32_t flags = c.flags;
    if (have) {
      const unsigned long long at = c.pos - 1;
      const bool here = zone[6 * lo] <= at && (lo + 1 >= n_zones || at < zone[6 * (lo + 1)]);
      if (!here) {
        lo = 0;
        uint32_t hi = n_zones;
        while (lo + 1 < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (zone[6 * mid] <= at) lo = mid; else hi = mid;
        }
      }
      if (zone[6 * lo] <= at) {  // (else: in the separator in front of the first zone)
        p = zone[6 * lo + 1] + (c.pos - zone[6 * lo]);  // end position in the text
        keep = p >= zone[6 * lo + 2] && p <= zone[6 * lo + 3];  // (else: context, or the separator behind the zone)
        if (keep && p == zone[6 * lo + 3] && zone[6 * lo + 4]) flags |= kCandCont;
      }
    }
    p8[j] = p;
    cost8[j] = c.cost;
    cf8[j] = flags;
    keep_bits |= (keep ? 1u : 0u) << j;
  }
  const uint32_t mine = (uint32_t)__popc(keep_bits);
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63) wave_sum[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    block_first = total ? atomicAdd(out_count, total) : 0u;
  }
  __syncthreads();
  uint32_t at = block_first + incl - mine;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) at += wave_sum[w];
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j)
    if ((keep_bits >> j) & 1u) {
      if (at < out_cap) out[at] = Candidate{p8[j], cost8[j], cf8[j]};
      ++at;
    }
}

// excl[2 x], excl[2 x + 1]: first and last end position of an interval where the zones are responsible (sorted,
// disjoint).  keep[i] = 0 for the records inside one.
__global__ __launch_bounds__(256) void drop_excluded_kernel(const Candidate* __restrict__ in, uint32_t count,
                                                            const unsigned long long* __restrict__ excl, uint32_t n_excl,
                                                            unsigned char* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const unsigned long long p = in[i].pos;
  uint32_t lo = 0, hi = n_excl;
  bool inside = false;
  if (n_excl && excl[0] <= p) {
    while (lo + 1 < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (excl[2 * mid] <= p) lo = mid; else hi = mid;
    }
    inside = p <= excl[2 * lo + 1];
  }
  keep[i] = inside ? 0 : 1;
}

hipError_t launch_dirty_scan(const uint8_t* d_text, uint64_t n, unsigned long long* d_starts, unsigned long long* d_ends,
                             unsigned long long* d_hard, uint32_t cap, uint32_t* d_counts, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint64_t n16 = (n + 15) / 16;
  hipLaunchKernelGGL(dirty_scan_kernel, dim3((uint32_t)std::min<uint64_t>(8192, (n16 + 255) / 256)), dim3(256), 0, stream,
                     d_text, n, d_starts, d_ends, d_hard, cap, d_counts);
  return hipGetLastError();
}
hipError_t launch_gather_zones(const uint8_t* d_text, uint8_t* d_dst, const unsigned long long* d_seg, uint32_t n_seg,
                               hipStream_t stream) {
  if (n_seg == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_zones_kernel, dim3(8, std::min<uint32_t>(n_seg, 4096)), dim3(256), 0, stream, d_text, d_dst, d_seg, n_seg);
  return hipGetLastError();
}
hipError_t launch_map_zone_list(const Candidate* d_in, uint32_t count, const unsigned long long* d_zone, uint32_t n_zones,
                                Candidate* d_out, uint32_t* d_out_count, uint32_t out_cap, hipStream_t stream) {
  if (count == 0 || n_zones == 0) return hipSuccess;
  hipLaunchKernelGGL(map_zone_list_kernel, dim3((count + kMapTile - 1) / kMapTile), dim3(256), 0, stream, d_in, count, d_zone, n_zones, d_out,
                     d_out_count, out_cap);
  return hipGetLastError();
}
hipError_t launch_drop_excluded(const Candidate* d_in, uint32_t count, const unsigned long long* d_excl, uint32_t n_excl,
                                unsigned char* d_keep, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(drop_excluded_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_in, count, d_excl, n_excl, d_keep);
  return hipGetLastError();
}

// packed[0 .. ceil(n / 16)) = the text's Dna codes (the caller pads the buffer with 4 more dwords)
hipError_t launch_pack_text(const uint8_t* d_text, uint64_t n, uint32_t* d_packed, hipStream_t stream) {
  const uint64_t n16 = (n + 15) / 16;
  if (n16 == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(8192, (n16 + 255) / 256);
  hipLaunchKernelGGL(pack_text_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(d_text), n16, d_packed);
  return hipGetLastError();
}

hipError_t launch_seed_search(const SeedParams& P, uint32_t grid, hipStream_t stream) {
  if (P.text_len == 0 || grid == 0) return hipSuccess;
  const dim3 g(grid), b(256);
  // (modes: see test_issue)
  const int mode = P.sub == nullptr ? 0 : P.pos64 ? 4 : P.pat_care ? 3 : P.win_dwords == 4 ? 1 : 2;
  if (P.m > 32) hipLaunchKernelGGL((seed_search_kernel<2, -2, 0>), g, b, 0, stream, P);
  else if (mode == 0) hipLaunchKernelGGL((seed_search_kernel<1, -2, 0>), g, b, 0, stream, P);
  else {
#define SASSY_SEED_LAUNCH(K, M) hipLaunchKernelGGL((seed_search_kernel<1, K, M>), g, b, 0, stream, P)
#define SASSY_SEED_MODES(K)                       \
  do {                                            \
    if (mode == 1) SASSY_SEED_LAUNCH(K, 1);       \
    else if (mode == 2) SASSY_SEED_LAUNCH(K, 2);  \
    else if (mode == 3) SASSY_SEED_LAUNCH(K, 3);  \
    else SASSY_SEED_LAUNCH(K, 4);                 \
  } while (0)
    if (P.k == 0) SASSY_SEED_MODES(0);
    else if (P.k == 1) SASSY_SEED_MODES(1);
    else if (P.k == 2) SASSY_SEED_MODES(2);
    else if (P.k == 3) SASSY_SEED_MODES(3);
    else SASSY_SEED_MODES(-1);
#undef SASSY_SEED_MODES
#undef SASSY_SEED_LAUNCH
  }
  return hipGetLastError();
}

}  // namespace sassy_hip
