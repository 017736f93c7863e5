// scan_kernel.hip -- K1: the text-tiled bit-parallel edit-distance scan for gfx950 (MI355X).
//
// What it computes (reference semantics, NOT reference code): the last DP row of the semi-global
// edit distance of one pattern against the text and, from it, the end
// positions the reference's search() / search_all() report (reference: src/search.rs:1286-1369).
// The recurrence per 64-column word and pattern row is Myers'99 in the text-tiled form the
// reference uses (reference: src/bitpacking.rs:63-85); everything around it is designed for the
// CDNA4 execution model:
//
//   * one wavefront lane owns ONE 64-bit word (64 text columns of the current pattern row) and
//     a private chunk of consecutive 64-byte text blocks; it walks the pattern rows of a block,
//     then moves to the next block of its chunk.  The per-row (+1/-1) carries between adjacent
//     blocks never leave the lane: 32 rows are packed into one 32-bit register pair (more rows:
//     one LDS slot pair per 32 rows).
//   * text reaches the lanes through an LDS tile: 8 coalesced 16 B/lane loads fetch 128 B
//     (one full cache line = 2 blocks) for each of the 64 lane chunks of a wave; the 16-byte
//     slots of a tile row are XOR-swizzled so that every lane then reads its own 64 bytes back
//     with four conflict-free ds_read_b128.
//   * the per-block equality masks ("profile", reference: Profile::encode_ref) are built by every
//     lane for its own block, all 64 lanes in parallel: bit b of the 4 bytes of a dword is
//     isolated with one v_and and gathered into a nibble with one v_dot4_u32_u8 (weights 1,2,4,8
//     / 16..128), giving the 64-bit bit-plane of that text bit; the slot masks are boolean
//     functions of the planes (Dna: 2 planes; Iupac: 5 planes through a bit-sliced letter->base-set
//     table; Ascii: 8 planes).  Masks go to LDS so that a pattern row fetches its Eq word with
//     one conflict-free ds_read_b64 at a wave-uniform slot offset.
//   * a block whose last row has no cell <= k (the normal case on random text) costs one cheap
//     popcount bound; only blocks that may hold a match run the exact 64-step minima scan and
//     append (end position, cost) records through an atomic counter.
//   * chunks are independent (fresh start + warm-up blocks).  The direction in
//     which a <=k plateau that straddles a chunk start was entered cannot be known locally; such
//     reports are flagged kCandCond and every chunk publishes its exit state so that the host
//     resolves them exactly (see DESIGN.md "seams").
//
// No MFMA: this is integer bit-twiddling bounded by VALU issue and HBM reads.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace sassy_hip {

__device__ __forceinline__ uint32_t lo32(uint64_t x) { return (uint32_t)x; }
__device__ __forceinline__ uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }

// v_bitop3_b32 with an explicit truth table: result bit = TT[(a << 2) | (b << 1) | c].
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}
__device__ __forceinline__ uint32_t mux(uint32_t s, uint32_t x0, uint32_t x1) {  // s ? x1 : x0
  return bitop3<0xAC>(s, x0, x1);
}

// One 64-column word of horizontal deltas (+1 mask, -1 mask), kept as explicit 32-bit halves so
// that every boolean of the step maps to one v_bitop3_b32 / v_and / v_or.
struct DpWord {
  uint32_t vpl, vph, vml, vmh;
};

// One pattern row on one 64-column word: Myers'99 in the text-tiled form (reference:
// src/bitpacking.rs:63-85).  eq = columns whose text char matches the row's
// pattern char; hp0/hm0 = vertical delta on the block's left edge in this row (0/1 each);
// the right-edge vertical delta is shifted into nhp/nhm (first row ends up in the top bit).
// (a << SH) + b on 64-bit register pairs: one full-rate VALU instruction on gfx950 (the compiler splits a
// 64-bit shift by one into three instructions when left to itself)
template <int SH>
__device__ __forceinline__ uint64_t lshl_add_u64(uint64_t a, uint64_t b) {
  uint64_t r;
  asm("v_lshl_add_u64 %0, %1, %3, %2" : "=&v"(r) : "v"(a), "v"(b), "n"(SH));
  return r;
}
// (a register pair, not arithmetic: `(hi << 32) | lo` left a v_and_or x, 1, 0 behind every carry bit)
typedef uint32_t u32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t pair64(uint32_t lo, uint32_t hi) {
  const u32pair v = {lo, hi};
  return __builtin_bit_cast(uint64_t, v);
}
// (the arithmetic form: list_rows_kernel, whose carry bits come out of one DPP register, is 5 us faster with it -- one wave per
// SIMD, every instruction of the chain counts, and the register pair costs it two copies per row)
__device__ __forceinline__ uint64_t pair64_or(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// zz: two registers holding 0, one per carry (each sits behind its carry bit in a 64-bit register pair; two, so
// that neither has to be copied into place row after row).
template <bool CARRY_PAIRS = true>
__device__ __forceinline__ void dp_row(DpWord& V, uint2 eq, uint32_t hp0, uint32_t hm0,
                                       uint32_t& nhp, uint32_t& nhm, uint2 zz = make_uint2(0u, 0u)) {
  auto mk = [](uint32_t lo, uint32_t hi) -> uint64_t { return CARRY_PAIRS ? pair64(lo, hi) : pair64_or(lo, hi); };
  const uint32_t vxl = eq.x | V.vml, vxh = eq.y | V.vmh;
  const uint32_t e2l = eq.x | hm0, e2h = eq.y;
  const uint64_t sum = lshl_add_u64<0>(mk(e2l & V.vpl, e2h & V.vph), mk(V.vpl, V.vph));
  const uint32_t hxl = bitop3<0xBE>(lo32(sum), V.vpl, e2l);  // (a ^ b) | c
  const uint32_t hxh = bitop3<0xBE>(hi32(sum), V.vph, e2h);
  const uint32_t Hpl = bitop3<0xF1>(V.vml, hxl, V.vpl), Hph = bitop3<0xF1>(V.vmh, hxh, V.vph);  // a | ~(b | c)
  const uint32_t Hml = V.vpl & hxl, Hmh = V.vph & hxh;
  nhp = __builtin_amdgcn_alignbit(nhp, Hph, 31);  // (nhp << 1) | (Hph >> 31)
  nhm = __builtin_amdgcn_alignbit(nhm, Hmh, 31);
  const uint64_t Hp2 = lshl_add_u64<1>(mk(Hpl, Hph), mk(hp0, zz.x));
  const uint64_t Hm2 = lshl_add_u64<1>(mk(Hml, Hmh), mk(hm0, zz.y));
  V.vpl = bitop3<0xF1>(lo32(Hm2), vxl, lo32(Hp2));
  V.vph = bitop3<0xF1>(hi32(Hm2), vxh, hi32(Hp2));
  V.vml = lo32(Hp2) & vxl;
  V.vmh = hi32(Hp2) & vxh;
}

// Lower bound on the minimum of the 65 cells of a row: cell b = ds + P_b - M_b with P_b / M_b the
// number of +1 / -1 deltas among the first b columns.  Inside byte q of the word every cell is
// >= ds + P_{8q} - M_{8q+8}.  Returns true when some cell MAY be <= k (never false for a live row).
__device__ __forceinline__ bool row_maybe_live(int ds, const DpWord& V, int k) {
  const uint32_t pl = V.vpl, ph = V.vph, ml = V.vml, mh = V.vmh;
  const int P8 = __popc(pl & 0xFFu), P16 = __popc(pl & 0xFFFFu), P24 = __popc(pl & 0xFFFFFFu);
  const int P32 = __popc(pl);
  const int P40 = P32 + __popc(ph & 0xFFu), P48 = P32 + __popc(ph & 0xFFFFu);
  const int P56 = P32 + __popc(ph & 0xFFFFFFu);
  const int M8 = __popc(ml & 0xFFu), M16 = __popc(ml & 0xFFFFu), M24 = __popc(ml & 0xFFFFFFu);
  const int M32 = __popc(ml);
  const int M40 = M32 + __popc(mh & 0xFFu), M48 = M32 + __popc(mh & 0xFFFFu);
  const int M56 = M32 + __popc(mh & 0xFFFFFFu), M64 = M32 + __popc(mh);
  int mn = min(-M8, P8 - M16);
  mn = min(mn, min(P16 - M24, P24 - M32));
  mn = min(mn, min(P32 - M40, P40 - M48));
  mn = min(mn, min(P48 - M56, P56 - M64));
  return ds + mn <= k;
}

// The exact answer behind row_maybe_live: does the block's last row hold a cell <= k?  Only the byte groups whose
// bound passes are walked, column by column.  With few pattern rows (m not much more than 8 + k) the bound passes for
// some lane of nearly every wave and block, and what used to follow -- scan_block's walk for the whole wave -- made the
// streaming DP five times slower for m = 12 than for m = 16 (5.5 against 1.0 ms per 3 GB); the same, less pronounced,
// where k / m is large.  Out of line: the streaming loop keeps its registers.
__device__ __noinline__ bool row_live_exact(uint32_t pl, uint32_t ph, uint32_t ml, uint32_t mh, int ds, int k) {
  // (the cell on the block's left edge -- the last column of the block in front -- counts: the report rule decides about
  // it when it sees this block's first column, and the byte-granular bound has always let such a block through)
  bool live = ds <= k;
  int s = ds;  // cost in front of the byte group
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint32_t pb = ((q < 4 ? pl : ph) >> (8 * (q & 3))) & 0xFFu, mb = ((q < 4 ? ml : mh) >> (8 * (q & 3))) & 0xFFu;
    if (__any(s - (int)__popc(mb) <= k)) {  // (wave-uniform branch; the lanes whose bound fails walk along idly)
      int c = s;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        c += (int)((pb >> t) & 1u) - (int)((mb >> t) & 1u);
        live |= c <= k;
      }
    }
    s += (int)__popc(pb) - (int)__popc(mb);
  }
  return live;
}

// Lane state, packed in one register: bit 0 = dec ("decreasing" of the report rule, reference:
// src/search.rs:1349-1359), bit 1 = amb (dec is not yet determined by anything this chunk has
// seen in its exact region).
constexpr uint32_t kStDec = 1u, kStAmb = 2u;

// What the rare-path code needs, passed by value so that the kernel argument struct never has
// to live in memory (taking its address would spill all of it to scratch).
struct EmitCtx {
  Candidate* cand;
  uint32_t* cand_count;
  uint64_t text_len;
  uint64_t global_offset;
  uint32_t cand_cap;
  uint32_t k;
  uint32_t flags;
  float alpha;        // overhang: extra cost floor(alpha * (pos - text_len)) past the text end
  uint32_t ov_steps;  // overhang: end positions up to text_len + ov_steps exist
  uint64_t text_begin; // buffer position of the text's column 0 (0 unless per-text mode)
  uint32_t tag;        // OR-ed into the flags of every report (per-text mode: the text index)
  // (72 bytes: a larger struct would no longer travel in registers but through scratch memory.  Window chunks
  // (kDescWindow: block b holds the text bytes [64 b + shift, 64 b + shift + 64)) keep their shift in flags >> 24.)
};
static_assert(sizeof(EmitCtx) <= 72, "EmitCtx must stay in registers (one more word and it goes through scratch)");
constexpr int kEmitShiftBit = 24;

// Exact walk over the 64 columns of a block whose last row may contain a cell <= k.
// Same decisions as the reference's find_minima_with_overhang with alpha = None
// (reference: src/search.rs:1286-1369), plus the seam bookkeeping.
//   b: block index inside the buffer; owned: the block belongs to this lane's chunk (reports are
//   made) or is warm-up (state only); x0: first local column whose <=k values are exact
//   (-1: all); last_warm: this is the block right before the chunk's first owned block.
// The walk only DECIDES: it returns the block's reports as a bit mask -- x / y: bit i-1 = end position base + i
// (i = 1 .. 64), z: bits 0..7 the lane state, bits 8..15 how many of the reports (in position order) are
// conditional (kCandCond: they are the first ones, `amb` only ever goes from true to false), bit 16 = a report
// at `base` itself.  emit_reports() then appends them for the whole wave with ONE atomic (a report each used to
// take its own: one address for the whole chip, ~10 ns apiece -- 743 000 reports cost the list kernel 7.8 ms).
constexpr uint32_t kRepBase = 1u << 16;
typedef uint32_t rep4 __attribute__((ext_vector_type(4)));  // (a native vector travels in registers; HIP's uint4 struct went through scratch)
__device__ __forceinline__ rep4 make_rep4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { rep4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
// By the WHOLE WAVE for one block: lane c looks at column c + 1.  (Until round 6 every lane walked its own block: a chain
// of 64 dependent, branchy steps -- ~20 us for one call, whatever the number of lanes that were in it -- and on sparse results
// (a few live blocks per wave: every search of the benchmarks) those calls were most of the chunk DP's time: 40 of the 75 us
// of config 3's list kernel, most of the 30 us a wave of the fused launch spends behind its stream.  Here the 64 costs come
// from two masked popcounts per lane, the comparisons leave as ballots, and the two sequential pieces of the rule become
// bit arithmetic on those 64-bit masks: `dec` after column c = "the last rise or fall at or before c was a fall" (a fill of
// the falls through the columns without either, in six doubling steps; dec_in in front of the first), `amb` is cleared by
// the first exact event, and the reports made up to that column are the conditional ones.)  All arguments wave-uniform.
// The rule itself, column by column (pos = base + bit, bit = 1 .. 64, while pos <= text end + overhang steps):
//   cost = ds + (+1 deltas) - (-1 deltas) up to the column (+ the overshoot cost behind the text end);
//   search_all: report pos when owned, pos > r0 and cost <= k;
//   else: rising / falling = cost above / below the previous column's; report the PREVIOUS position when dec, rising,
//   its cost <= k, owned and behind r0 (conditional while `amb`); dec = falling || (dec && !rising); an exact column
//   (pos > x0) that rises, falls, exceeds k (itself or the one before) or costs 0 settles the state: amb = false (owned)
//   / determined (warm-up); behind the last warm-up block amb = !determined; the end of the text ends a plateau.
__device__ __forceinline__ rep4 scan_block_cols(const EmitCtx& P, uint32_t flags, uint64_t text_len, uint64_t text_begin, uint64_t vp,
                                                uint64_t vm, int ds, uint64_t b, bool owned, bool last_warm, int64_t x0, uint32_t state) {
  const uint32_t lane = __lane_id();
  const int k = (int)P.k;
  const bool all = (flags & kScanAllMinima) != 0;
  const bool dec_in = (state & kStDec) != 0, amb_in = (state & kStAmb) != 0;
  const int64_t r0 = x0 + (int64_t)((state >> 8) & 0xFFFFu);
  const uint64_t base = b * 64 + (flags >> kEmitShiftBit);
  const uint64_t max_pos = text_len + P.ov_steps;
  if (base >= max_pos) return make_rep4(0u, 0u, state & 0xFFu, 0u);
  const uint32_t nbits = max_pos - base < 64 ? (uint32_t)(max_pos - base) : 64u;  // columns that exist (>= 1)
  const bool ov = P.ov_steps != 0;
  auto total_of = [&](int c, uint64_t pos) -> int {
    return (ov && pos > text_len) ? c + __float2int_rd(P.alpha * (float)(pos - text_len)) : c;
  };
  const int cost0 = total_of(ds, base);
  const uint64_t low = lane == 63u ? ~0ull : ((2ull << lane) - 1ull);
  const uint64_t pos = base + lane + 1u;
  const int cost = total_of(ds + (int)__popcll(vp & low) - (int)__popcll(vm & low), pos);
  const bool valid = lane < nbits;
  const int prevc = __builtin_amdgcn_update_dpp(cost0, cost, 0x138, 0xF, 0xF, false);  // wave_shr:1 (lane 0: the block's left edge)
  const uint64_t LEK = __ballot(valid && cost <= k);
  uint64_t rep = 0;
  uint32_t rep_base = 0, ncond = 0;
  bool dec = dec_in, amb = amb_in;
  if (all) {
    if (owned && cost0 <= k && base == text_begin && P.global_offset == 0 && (flags & kScanTextStart)) rep_base = kRepBase;
    rep = __ballot(valid && owned && (int64_t)pos > r0 && cost <= k);
  } else {
    const uint64_t VAL = nbits == 64u ? ~0ull : ((1ull << nbits) - 1ull);
    const uint64_t R = __ballot(valid && cost > prevc), F = __ballot(valid && cost < prevc);
    const uint64_t ZERO = __ballot(valid && cost == 0);
    const uint64_t EX = __ballot(valid && (int64_t)pos > x0);
    const uint64_t OWNP = __ballot(valid && owned && (int64_t)(pos - 1u) > r0);  // (the report is for the column in front)
    const uint64_t E = R | F;
    uint64_t D = F, Pr = ~E;  // D: dec behind column c -- a fall at j <= c and neither rise nor fall in (j, c]
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      D |= Pr & (D << sft);
      Pr &= Pr << sft;
    }
    if (dec_in) D |= E ? ((E & (0ull - E)) - 1ull) : ~0ull;  // ... or dec_in and neither in [0, c]
    const uint64_t LEKp = (LEK << 1) | (cost0 <= k ? 1ull : 0ull);
    const uint64_t Dp = (D << 1) | (dec_in ? 1ull : 0ull);
    const uint64_t REP = R & Dp & LEKp & OWNP;  // bit c: a report for end position base + c
    const uint64_t GTK = ~LEK & VAL, GTKp = ((~LEK << 1) | (cost0 > k ? 1ull : 0ull)) & VAL;
    const uint64_t EVX = (E | GTK | GTKp | ZERO) & EX;  // exact columns that settle the plateau state
    bool determined = x0 < 0;
    if (owned) {
      // (the column of the first exact event still reports with the old `amb`: the reports up to it are the conditional ones)
      if (amb_in) ncond = (uint32_t)__popcll(REP & (EVX ? (((EVX & (0ull - EVX)) << 1) - 1ull) : ~0ull));
      if (EVX) amb = false;
    } else if (EVX) {
      determined = true;
    }
    dec = ((D >> (nbits - 1u)) & 1ull) != 0;
    if (last_warm) amb = !determined;
    rep_base = (REP & 1ull) ? kRepBase : 0u;
    rep = REP >> 1;
    if (owned && (flags & kScanTextEnd) && base + nbits == max_pos && dec && ((LEK >> (nbits - 1u)) & 1ull) &&
        (int64_t)(base + nbits) > r0) {  // the end of the text ends a plateau (src/search.rs:1352-1366)
      rep |= 1ull << (nbits - 1u);
      if (amb) ncond = (uint32_t)__popcll(rep) + (rep_base ? 1u : 0u);
    }
  }
  return make_rep4((uint32_t)rep, (uint32_t)(rep >> 32), (dec ? kStDec : 0u) | (amb ? kStAmb : 0u) | (ncond << 8) | rep_base, 0u);
}
// scan_block for every lane with `live` set, one block after the other, each by the whole wave.  Called in wave-uniform
// control flow by all lanes; a lane without a live block gets {0, 0, its state, 0}.
__device__ __noinline__ rep4 scan_blocks(const EmitCtx P, bool live, uint64_t vp, uint64_t vm, int ds, uint64_t b, bool owned,
                                          bool last_warm, int64_t x0, uint32_t state) {
  rep4 out = make_rep4(0u, 0u, state & 0xFFu, 0u);
  unsigned long long todo = __ballot(live);
  const uint32_t lane = __lane_id();
  auto bc32 = [](uint32_t v, int L) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, L); };
  auto bc64 = [&](uint64_t v, int L) -> uint64_t { return pair64(bc32(lo32(v), L), bc32(hi32(v), L)); };
  while (todo) {
    const int L = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const uint32_t fl = bc32(P.flags, L);
    const uint32_t bits = bc32((owned ? 1u : 0u) | (last_warm ? 2u : 0u), L);
    const rep4 r = scan_block_cols(P, fl, bc64(P.text_len, L), bc64(P.text_begin, L), bc64(vp, L), bc64(vm, L), (int)bc32((uint32_t)ds, L),
                                   bc64(b, L), (bits & 1u) != 0, (bits & 2u) != 0, (int64_t)bc64((uint64_t)x0, L), bc32(state, L));
    if (lane == (uint32_t)L) out = r;
  }
  return out;
}

// Appends the reports scan_block() decided on, for all lanes of the wave at once: called in wave-uniform control
// flow (every lane of the wave, lanes without a report pass r = 0), one atomic on the report counter per call
// (and one on the TextStash counter with kScanStash: a slot per reporting block).  vp / vm / ds / b: the block's last
// row, as scan_block got them (a report's cost is re-derived from them).  Returns the lane's TextStash slot + 1
// (0: none, 0xFFFFFF: out of slots), which the caller -- who has the block's text -- fills.
__device__ __noinline__ uint32_t emit_reports(const EmitCtx P, rep4 r, uint64_t vp, uint64_t vm, int ds, uint64_t b) {
  uint64_t rep = ((uint64_t)r.y << 32) | r.x;
  const uint32_t rep_base = (r.z & kRepBase) ? 1u : 0u;
  const uint32_t cnt = (uint32_t)__popcll(rep) + rep_base;
  const uint32_t lane = __lane_id();
  uint32_t inc = cnt;  // inclusive prefix sum over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += t;
  }
  const uint32_t total = __shfl(inc, 63, 64);
  if (total == 0) return 0u;
  uint32_t first = 0;
  if (lane == 0) first = atomicAdd(P.cand_count, total);
  first = __shfl(first, 0, 64);
  uint32_t idx = first + inc - cnt;
  uint32_t slot_tag = 0;
  if (P.flags & kScanStash) {
    const uint64_t who = __ballot(cnt != 0);
    uint32_t sfirst = 0;
    if (lane == 0) sfirst = atomicAdd(P.cand_count + kCtlStashWord, (uint32_t)__popcll(who));
    sfirst = __shfl(sfirst, 0, 64);
    const uint32_t sl = sfirst + (uint32_t)__popcll(who & ((1ull << lane) - 1ull));
    if (cnt != 0) slot_tag = sl < 0xFFFFFEu ? sl + 1u : 0xFFFFFFu;
  }
  if (cnt == 0) return 0u;
  const uint32_t tagbits = ((slot_tag == 0xFFFFFFu ? 0u : slot_tag << kCandTextShift)) | P.tag;
  uint32_t n_cond = (r.z >> 8) & 0xFFu;
  const uint64_t base = b * 64 + (P.flags >> kEmitShiftBit);
  const bool ov = P.ov_steps != 0;
  auto put = [&](uint64_t pos, int raw) {
    const int cost = (ov && pos > P.text_len) ? raw + __float2int_rd(P.alpha * (float)(pos - P.text_len)) : raw;
    if (idx < P.cand_cap) {
      Candidate c;
      c.pos = P.global_offset + pos;
      c.cost = cost;
      c.flags = tagbits | (n_cond ? kCandCond : 0u);
      P.cand[idx] = c;
    }
    if (n_cond) --n_cond;
    ++idx;
  };
  if (rep_base) put(base, ds);
  while (rep) {
    const int i = __ffsll((long long)rep);  // 1-based bit = column offset
    rep &= rep - 1;
    const uint64_t low = i == 64 ? ~0ull : ((1ull << i) - 1ull);
    put(base + (uint64_t)i, ds + (int)__popcll(vp & low) - (int)__popcll(vm & low));
  }
  return slot_tag;
}

// 16 text bytes that straddle or lie past the end of the buffer (cold path): bytes past the end
// read as 'X' (reference: src/search.rs:202-207).
__device__ __noinline__ uint4 load_tail16(const uint8_t* text, uint64_t off, uint64_t text_len, uint32_t pad = 'X') {
  uint64_t lo = 0x0101010101010101ull * pad, hi = lo;  // 'X', or 'N' with overhang
  if (off < text_len) {
    const uint32_t valid = (uint32_t)min((uint64_t)16, text_len - off);
#pragma unroll 1
    for (uint32_t q = 0; q < valid; ++q) {
      const uint64_t ch = text[off + q];
      const uint32_t sh = 8u * (q & 7u);
      if (q < 8) lo = (lo & ~(0xFFull << sh)) | (ch << sh);
      else hi = (hi & ~(0xFFull << sh)) | (ch << sh);
    }
  }
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// 16 bytes of the text a list-mode launch scans, at logical offset `off`: the buffer itself, or --
// Rc strand without a reversed copy -- the rev_n bytes at `text` read backwards (one misaligned
// 16-byte load + four byte swaps; the list kernels read a few MB, the misalignment costs nothing there).
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
__device__ __noinline__ uint4 load_tail16_rev(const uint8_t* text, uint64_t off, uint64_t n) {
  uint32_t w[4] = {0x58585858u, 0x58585858u, 0x58585858u, 0x58585858u};  // past the end: 'X'
#pragma unroll 1
  for (uint32_t q = 0; q < 16 && off + q < n; ++q) {
    const uint32_t sh = 8u * (q & 3u);
    w[q >> 2] = (w[q >> 2] & ~(0xFFu << sh)) | ((uint32_t)text[n - 1 - (off + q)] << sh);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 load_text16(const ScanParams& P, uint64_t off) {
  if (P.rev_n == 0) {
    if (off + 16 <= P.text_len) {  // (any alignment: window chunks start at any byte)
      const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned*>(P.text + off);
      return make_uint4(v.x, v.y, v.z, v.w);
    }
    return load_tail16(P.text, off, P.text_len);
  }
  if (off + 16 <= P.rev_n) {
    const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned*>(P.text + (P.rev_n - off - 16));
    return make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x));
  }
  return load_tail16_rev(P.text, off, P.rev_n);
}

// The 64 text bytes of block `blk` for a lane of the list kernels (lanes with on = false get 'X').  The
// lanes of a wave read blocks that lie far apart in a multi-GB buffer: every load is a TLB miss plus an
// HBM miss, microseconds, and a list kernel has only a few waves per SIMD to hide them behind.  So the
// common case -- forward text, block inside the buffer, for every lane of the wave -- is four loads in
// ONE basic block: they are issued back to back and waited for once, where they are used (the caller
// fetches one block ahead).  A wave with a lane at the buffer's tail, and the Rc strand's backwards
// reads, take the general path chunk by chunk.
__device__ __forceinline__ void fetch_block(const ScanParams& P, bool on, uint64_t blk, uint32_t (&dst)[16], uint32_t shift = 0) {
  const uint64_t off = blk * 64 + shift;
  const bool plain = P.rev_n == 0 && off + 64 <= P.text_len && shift == 0;
  uint4 v0 = make_uint4(0x58585858u, 0x58585858u, 0x58585858u, 0x58585858u), v1 = v0, v2 = v0, v3 = v0;
  if (__all(!on || plain)) {  // wave-uniform
    if (on) {
      const uint4* p = reinterpret_cast<const uint4*>(P.text + off);
      v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3];
    }
  } else if (on) {
    v0 = load_text16(P, off);
    v1 = load_text16(P, off + 16);
    v2 = load_text16(P, off + 32);
    v3 = load_text16(P, off + 48);
  }
  dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w;
  dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
  dst[8] = v2.x; dst[9] = v2.y; dst[10] = v2.z; dst[11] = v2.w;
  dst[12] = v3.x; dst[13] = v3.y; dst[14] = v3.z; dst[15] = v3.w;
}

// ------------------------------------------------------------------ the profile, lane-parallel
// Bit-plane BIT of the lane's 64 text bytes: bit c of the result = bit BIT of byte c.
// Per dword: v_and isolates the bit of its 4 bytes, v_dot4_u32_u8 with weights 1,2,4,8 (even
// dword of a pair) / 16,32,64,128 (odd dword) gathers them; 8 chars land in bits BIT..BIT+7.
template <int BIT>
__device__ __forceinline__ uint2 bit_plane(const uint32_t (&x)[16]) {
  constexpr uint32_t kSel = 0x01010101u << BIT;
  uint32_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t a = __builtin_amdgcn_udot4(x[2 * i] & kSel, 0x08040201u, 0u, false);
    v[i] = __builtin_amdgcn_udot4(x[2 * i + 1] & kSel, 0x80402010u, a, false);
  }
  uint2 r;
  r.x = (v[0] >> BIT) | (v[1] << (8 - BIT)) | (v[2] << (16 - BIT)) | (v[3] << (24 - BIT));
  r.y = (v[4] >> BIT) | (v[5] << (8 - BIT)) | (v[6] << (16 - BIT)) | (v[7] << (24 - BIT));
  return r;
}

// IUPAC letter (c & 31) -> low nibble of its base set; non-letters act as N (15), X = 0
// (reference: src/profiles/iupac.rs:281-330).  A=1 C=2 T=4 G=8.
__host__ __device__ constexpr int iupac_nib(int i) {
  return i == 1 ? 1 : i == 3 ? 2 : i == 20 ? 4 : i == 21 ? 4 : i == 7 ? 8 : i == 14 ? 15
       : i == 18 ? 9 : i == 25 ? 6 : i == 19 ? 10 : i == 23 ? 5 : i == 11 ? 12 : i == 13 ? 3
       : i == 2 ? 14 : i == 4 ? 13 : i == 8 ? 7 : i == 22 ? 11 : i == 24 ? 0 : 15;
}
// Truth table (index = b2*4 + b1*2 + b0) of output bit O of the nibble table for letters
// 8*HI .. 8*HI+7.
template <int O, int HI>
struct IupacTT {
  static constexpr int value() {
    int tt = 0;
    for (int i = 0; i < 8; ++i) tt |= ((iupac_nib(HI * 8 + i) >> O) & 1) << i;
    return tt;
  }
};
// Bit-sliced table lookup: base-set bit O of all 32 text chars of a half word, from the five
// letter-index planes b0..b4: Shannon expansion on b4, b3 over four 3-input functions.
template <int O>
__device__ __forceinline__ uint32_t iupac_base_plane(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3,
                                                     uint32_t b4) {
  const uint32_t g0 = bitop3<IupacTT<O, 0>::value()>(b2, b1, b0);
  const uint32_t g1 = bitop3<IupacTT<O, 1>::value()>(b2, b1, b0);
  const uint32_t g2 = bitop3<IupacTT<O, 2>::value()>(b2, b1, b0);
  const uint32_t g3 = bitop3<IupacTT<O, 3>::value()>(b2, b1, b0);
  return mux(b4, mux(b3, g0, g1), mux(b3, g2, g3));
}

// Slot masks of one block from the lane's 64 text bytes.  m[s] = mask of slot s (lo, hi).
template <int PROFILE, int NS>
__device__ __forceinline__ void build_masks(const uint32_t (&x)[16], const ScanParams& P, uint2 (&m)[NS]) {
  if constexpr (PROFILE == PROFILE_DNA) {
    // code = (c >> 1) & 3: A=0 C=1 T=2 G=3 (reference: src/profiles/dna.rs:19-40)
    const uint2 p1 = bit_plane<1>(x), p2 = bit_plane<2>(x);
    m[0] = make_uint2(~(p1.x | p2.x), ~(p1.y | p2.y));
    m[1] = make_uint2(p1.x & ~p2.x, p1.y & ~p2.y);
    m[2] = make_uint2(~p1.x & p2.x, ~p1.y & p2.y);
    m[3] = make_uint2(p1.x & p2.x, p1.y & p2.y);
  } else if constexpr (PROFILE == PROFILE_IUPAC) {
    // mask[slot] = (base set of the text letter) intersects (base set of the slot's pattern
    // letter) (reference: src/profiles/iupac.rs:68-128)
    const uint2 b0 = bit_plane<0>(x), b1 = bit_plane<1>(x), b2 = bit_plane<2>(x), b3 = bit_plane<3>(x),
                b4 = bit_plane<4>(x);
    uint2 base[4];
    base[0] = make_uint2(iupac_base_plane<0>(b0.x, b1.x, b2.x, b3.x, b4.x), iupac_base_plane<0>(b0.y, b1.y, b2.y, b3.y, b4.y));
    base[1] = make_uint2(iupac_base_plane<1>(b0.x, b1.x, b2.x, b3.x, b4.x), iupac_base_plane<1>(b0.y, b1.y, b2.y, b3.y, b4.y));
    base[2] = make_uint2(iupac_base_plane<2>(b0.x, b1.x, b2.x, b3.x, b4.x), iupac_base_plane<2>(b0.y, b1.y, b2.y, b3.y, b4.y));
    base[3] = make_uint2(iupac_base_plane<3>(b0.x, b1.x, b2.x, b3.x, b4.x), iupac_base_plane<3>(b0.y, b1.y, b2.y, b3.y, b4.y));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint32_t sv = P.slot_val[s];  // wave-uniform; unused slots hold 0 -> empty mask
      uint2 r = make_uint2(0u, 0u);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const uint32_t sel = ((sv >> o) & 1u) ? 0xFFFFFFFFu : 0u;
        r.x |= base[o].x & sel;
        r.y |= base[o].y & sel;
      }
      m[s] = r;
    }
  } else if constexpr (PROFILE == (int)PROFILE_ASCII_BYTES) {
    // byte mode: the eight bit planes themselves (dp_word compares them with the row's pattern byte)
    static_assert(NS == 8, "byte mode keeps eight planes");
    m[0] = bit_plane<0>(x); m[1] = bit_plane<1>(x); m[2] = bit_plane<2>(x); m[3] = bit_plane<3>(x);
    m[4] = bit_plane<4>(x); m[5] = bit_plane<5>(x); m[6] = bit_plane<6>(x); m[7] = bit_plane<7>(x);
  } else {
    // Ascii: byte equality with the slot's pattern byte (reference: src/profiles/ascii.rs:75-90)
    uint2 pl[8];
    pl[0] = bit_plane<0>(x); pl[1] = bit_plane<1>(x); pl[2] = bit_plane<2>(x); pl[3] = bit_plane<3>(x);
    pl[4] = bit_plane<4>(x); pl[5] = bit_plane<5>(x); pl[6] = bit_plane<6>(x); pl[7] = bit_plane<7>(x);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint32_t sv = P.slot_val[s];
      uint2 r = (s < (int)P.nslots) ? make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu) : make_uint2(0u, 0u);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint32_t inv = ((sv >> b) & 1u) ? 0u : 0xFFFFFFFFu;  // XNOR with the slot's bit
        r.x &= pl[b].x ^ inv;
        r.y &= pl[b].y ^ inv;
      }
      m[s] = r;
    }
  }
}

// first block a chunk touches: its first owned block minus `back` = the warm-up plus the parity
// that makes it even (a staged pair of blocks is then one aligned 128-byte line; bpl is even),
// clipped at the buffer start (only the very first chunk of a text clips).
__device__ __forceinline__ uint64_t chunk_blk0(uint64_t first_owned, uint32_t bpl, uint32_t back, uint64_t chunk) {
  const uint64_t start = first_owned + chunk * (uint64_t)bpl;
  return start > back ? start - back : 0;
}

typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;  // scalar (s_load) reads

// Row -> LDS offset of its slot mask, from the packed row table: one byte per row holding
// slot*2, so that (byte << 8) = slot * 512.  Scalar extraction (s_bfe) + one VALU add.
__device__ __forceinline__ uint32_t row_mask_off(const uint32_t (&pk)[8], int r) {
  return ((pk[r >> 2] >> (8 * (r & 3))) & 0xFFu) << 8;
}

// Bounded rows (reference: src/search.rs:1129-1162 early-termination check, :941-975 check_lanes /
// min_in_lane, :1244-1249 reset_rows).  After `done` rows of a word, row `done` of the
// block (the one just finished) is DEAD for a lane when
//   (1) none of its 65 cells (left edge + 64 columns) can be <= k: the byte-granular popcount bound
//       row_maybe_live on the row's horizontal deltas, and
//   (2) no left-edge cell further down is <= k either: the left-edge cost L at this row minus every
//       -1 vertical delta below it is still > k.
// Then every cell below this row inside the block is > k (a cell <= k needs a neighbour <= k above it or
// on the left edge), so the remaining rows are skipped and their right-edge carries are reset to (+1, 0) --
// an over-estimate that only ever reaches cells whose true value is > k.  The lanes of a wave run in
// lockstep, so the rows stop when ALL lanes of the wave are dead (a wave vote; the reference votes over
// its 4 / 8 SIMD lanes).  Tests cost about two rows' worth of VALU: they start at the row count the wave
// learnt from its previous blocks (first_test) and repeat every 4 rows.
constexpr uint32_t kCutFirstRows = 8;  // a wave's first block tests from here (reference: CHECK_AT_LEAST_ROWS = 8)
struct CutCtx {
  int k;
  int ds_word;         // left-edge cost above this word's first row
  int minus_below;     // -1 vertical deltas on the left edge in all LATER words
  uint32_t first_test; // test when at least this many rows of the word are done (> 32: never)
  bool more_words;     // rows follow behind this word
  bool idle;           // the lane's block does not count (no chunk / outside its range): votes "dead"
};

// The rows of one 32-row word.  pk_in holds the profile slot of each row (one byte per row).
// SCALAR_PK: the word (and so its row table) is the same for the whole wave (scalar registers);
// false: every lane works on its own word (list_words_kernel).
// CUT: bounded rows (see CutCtx).  Returns 0 when all rows of the word were computed, else the number of
// rows (4 .. 32) after which the wave stopped (nhp / nhm then hold (+1, 0) for the skipped rows).
// BYTES (PROFILE_ASCII_BYTES): pk_in holds the rows' pattern bytes, my_masks the block's eight bit planes; a row's
// Eq word = AND over the bits b of (plane b XNOR bit b of the row's byte).
template <bool SCALAR_PK = true, bool CUT = false, bool BYTES = false>
__device__ __forceinline__ uint32_t dp_word(DpWord& V, const unsigned char* my_masks, uint32_t ohp, uint32_t ohm,
                                            const uint32_t (&pk_in)[8], uint32_t rows, uint32_t& nhp_out,
                                            uint32_t& nhm_out, const CutCtx* cut = nullptr) {
  // Opaque copies: keeps the 32 per-row offsets from being hoisted out of the block loop as 32
  // live scalars (SGPR spills cost VALU v_readlane ops); re-deriving them is one s_bfe each.
  uint32_t pk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    pk[i] = pk_in[i];
    if constexpr (SCALAR_PK) asm volatile("" : "+s"(pk[i]));
  }
  uint32_t nhp = 0, nhm = 0, done = 0;
  bool stopped = false;  // wave-uniform
  uint2 zz = make_uint2(0u, 0u);
  asm volatile("" : "+v"(zz.x), "+v"(zz.y));
  uint2 planes[BYTES ? 8 : 1];
  if constexpr (BYTES) {
#pragma unroll
    for (int b = 0; b < 8; ++b) planes[b] = *reinterpret_cast<const uint2*>(my_masks + b * 512);
  }
  auto eq_of_byte = [&](uint32_t pb) -> uint2 {  // pb: the row's pattern byte (bits above 7 ignored)
    uint2 e = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint32_t n = ~(uint32_t)__builtin_amdgcn_sbfe((int)pb, b, 1);  // all ones iff bit b of the byte is 0
      e.x = bitop3<0x60>(e.x, planes[BYTES ? b : 0].x, n);  // a & (b ^ c)
      e.y = bitop3<0x60>(e.y, planes[BYTES ? b : 0].y, n);
    }
    return e;
  };
  auto row_byte = [&](int r) -> uint32_t { return pk[r >> 2] >> (8 * (r & 3)); };
  uint2 eqn[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if constexpr (BYTES) eqn[u] = eq_of_byte(row_byte(u));
    else eqn[u] = *reinterpret_cast<const uint2*>(my_masks + row_mask_off(pk, u));
  }
  // (straight-line flow with early exits: nothing but the exit merges register values, so the Eq words of the next
  // group are loaded into fresh registers instead of being copied into place)
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (4u * g + 4u > rows) break;  // wave-uniform
    uint2 eqc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) eqc[u] = eqn[u];
    if (g < 7) {  // prefetch the next group's Eq words while this group computes
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (BYTES) eqn[u] = eq_of_byte(row_byte(4 * g + 4 + u));
        else eqn[u] = *reinterpret_cast<const uint2*>(my_masks + row_mask_off(pk, 4 * g + 4 + u));
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      dp_row(V, eqc[u], __builtin_amdgcn_ubfe(ohp, 31u - (4 * g + u), 1u), __builtin_amdgcn_ubfe(ohm, 31u - (4 * g + u), 1u), nhp, nhm, zz);
    done = 4u * g + 4u;
    if constexpr (CUT) {
      if (done >= cut->first_test && (done < rows || cut->more_words)) {  // wave-uniform
        const uint32_t top = 0xFFFFFFFFu << (28 - 4 * g);  // rows 0 .. done-1 of the word (compile-time: g is unrolled)
        // (HIP's __popc returns unsigned: keep the arithmetic signed, the bound is often negative)
        const int here = cut->ds_word + (int)__popc(ohp & top) - (int)__popc(ohm & top);
        const bool below_dead = here - (int)__popc(ohm & ~top) - cut->minus_below > cut->k;
        // two votes: the cheap left-edge condition first (a lane whose left edge is still <= k further down keeps
        // the whole wave going whatever this row looks like), the row's own cells only when it holds everywhere
        if (__all(cut->idle || below_dead)) stopped = __all(cut->idle || !row_maybe_live(here, V, cut->k));
        if (stopped) break;
      }
    }
  }
  if (!stopped) {
    // up to three leftover rows when the word's row count is not a multiple of 4: they all lie in
    // table word done/4
    if (done < rows) {
      const uint32_t q = done >> 2;
      const uint32_t pw = q == 0 ? pk[0] : q == 1 ? pk[1] : q == 2 ? pk[2] : q == 3 ? pk[3]
                        : q == 4 ? pk[4] : q == 5 ? pk[5] : q == 6 ? pk[6] : pk[7];
      for (uint32_t r = done; r < rows; ++r) {
        uint2 eq;
        if constexpr (BYTES) eq = eq_of_byte(pw >> (8 * (r & 3)));
        else eq = *reinterpret_cast<const uint2*>(my_masks + (((pw >> (8 * (r & 3))) & 0xFFu) << 8));
        dp_row(V, eq, (ohp >> (31 - r)) & 1u, (ohm >> (31 - r)) & 1u, nhp, nhm, zz);
      }
    }
    if (rows < 32) { nhp <<= (32 - rows); nhm <<= (32 - rows); }
    nhp_out = nhp;
    nhm_out = nhm;
    return 0;
  }
  // stopped after `done` (4 .. 32) rows: row r of the word sits at bit 31 - r; the skipped rows done .. rows-1
  // get the right-edge carry (+1, 0)
  const uint32_t word_rows_mask = rows == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> rows);
  const uint32_t done_mask = done == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> done);
  const uint32_t sh = 32u - done;  // 0 .. 28
  nhp_out = (sh ? (nhp << sh) : nhp) | (word_rows_mask & ~done_mask);
  nhm_out = sh ? (nhm << sh) : nhm;
  return done;
}

// The DP rows of one block for a lane whose per-row carries live in LDS (scan_kernel, list_kernel): all
// words top-down with the wave-wide row cut-off.  Returns true when the block ran to its last row (then
// V / ds describe that row), false when the rows were cut (no cell <= k in the block for any lane).
// first_rows (wave-uniform, in/out): the row count from which this wave tests; it follows the text --
// down to four rows before the last stop, up by four after a block that ran through.
// minus_total (per lane, in/out): -1 vertical deltas on the block's left edge over all words.
// AHEAD: the next word's carries are fetched while this word's rows run (carries in global memory: a load per word
// would otherwise be waited for 375 times per block at m = 12 000).
template <bool BYTES = false, bool AHEAD = false>
__device__ __forceinline__ bool dp_block(DpWord& V, int& ds, const unsigned char* my_masks, uint32_t* carry, uint32_t lane,
                                         const_u32_ptr row_tab, const uint32_t (&pkw0)[8], uint32_t nwords, uint32_t last_rows,
                                         uint32_t last_word_init, int k, bool idle, uint32_t& first_rows, int& minus_total,
                                         uint32_t m, bool counting, unsigned long long& cnt_rows) {
  V.vpl = V.vph = V.vml = V.vmh = 0;  // row 0 of the matrix is all 0: horizontal deltas 0
  ds = 0;                             // cost at the block's left edge, rows above the current word
  int seen_minus = 0, next_minus = 0;
  uint32_t stop_at = 0;               // != 0: the wave stopped after this many rows of the block
  uint32_t ahp = 0, ahm = 0;
  if constexpr (AHEAD) { ahp = carry[lane]; ahm = carry[64 + lane]; }
  for (uint32_t w = 0; w < nwords; ++w) {
    const bool last = w == nwords - 1;
    if (stop_at) {  // wave-uniform: the rows of this word are skipped, right-edge carry (+1, 0)
      carry[(w * 2 + 0) * 64 + lane] = last ? last_word_init : 0xFFFFFFFFu;
      carry[(w * 2 + 1) * 64 + lane] = 0;
      continue;
    }
    uint32_t ohp, ohm;
    if constexpr (AHEAD) {
      ohp = ahp; ohm = ahm;
      if (!last) { ahp = carry[(w * 2 + 2) * 64 + lane]; ahm = carry[(w * 2 + 3) * 64 + lane]; }
    } else {
      ohp = carry[(w * 2 + 0) * 64 + lane];
      ohm = carry[(w * 2 + 1) * 64 + lane];
    }
    seen_minus += (int)__popc(ohm);
    const uint32_t rows = last ? last_rows : 32u;
    uint32_t pkw[8];
    if (w == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) pkw[i] = pkw0[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) pkw[i] = row_tab[8 * w + i];
    }
    CutCtx cc;
    cc.k = k;
    cc.ds_word = ds;
    cc.minus_below = minus_total - seen_minus;
    cc.first_test = first_rows > 32u * w ? first_rows - 32u * w : 4u;
    cc.more_words = !last;
    cc.idle = idle;
    uint32_t nhp, nhm;
    const uint32_t cut_at = dp_word<true, true, BYTES>(V, my_masks, ohp, ohm, pkw, rows, nhp, nhm, &cc);
    ds += (int)__popc(ohp) - (int)__popc(ohm);
    carry[(w * 2 + 0) * 64 + lane] = nhp;
    carry[(w * 2 + 1) * 64 + lane] = nhm;
    next_minus += (int)__popc(nhm);
    if (cut_at) stop_at = 32u * w + cut_at;
  }
  minus_total = next_minus;
  if (counting) cnt_rows += stop_at ? stop_at : m;
  if (stop_at) {
    first_rows = stop_at;  // the next block tests there first
    return false;
  }
  if (first_rows < m) first_rows = first_rows + 4u < m ? first_rows + 4u : m;  // ran through: test later (never, on match-dense text)
  return true;
}

// SB = text blocks per lane chunk fetched by one staging step: 2 = one full 128-byte line per
// chunk (8 KiB tile), 1 = half lines (4 KiB tile, more waves fit in the LDS).
// GC: the per-row carries in global memory (P.carry_global; patterns whose carries do not fit one wave's LDS).
template <int PROFILE, int NS, int SB, bool GC = false>
__global__ __launch_bounds__(256) void scan_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kRowBytes = 64u * SB;          // tile row = the staged bytes of one lane chunk
  constexpr uint32_t kSlots = 4u * SB;              // 16-byte slots per row
  constexpr uint32_t kOwnersPerInstr = 64u / kSlots;  // tile rows filled by one 64-lane load
  constexpr int kStageInstr = 4 * SB;
  constexpr uint32_t kTile = 64u * kRowBytes;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* wbase = smem + (size_t)wave * P.lds_per_wave;
  unsigned char* tile = wbase;
  unsigned char* mask_bytes = wbase + kTile;                                 // [NS][64] u64
  uint32_t* carry = reinterpret_cast<uint32_t*>(wbase + kTile + NS * 512);  // [word][hp|hm][lane]
  if constexpr (GC) carry = P.carry_global + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * P.nwords * 128u;

  const uint64_t wave_chunk0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave) * kWave;  // (1 .. 4 waves per workgroup)
  if (wave_chunk0 >= P.n_chunks) return;  // wave-uniform
  const uint64_t chunk = wave_chunk0 + lane;

  const uint32_t bpl = P.bpl;
  const uint64_t first_owned = P.first_owned_block;
  const uint32_t back = P.wb + (uint32_t)((first_owned + P.wb) & 1u);  // warm-up + evenness (bpl is even)
  const uint64_t own_lo = first_owned + chunk * (uint64_t)bpl;
  uint64_t own_hi = own_lo + bpl;
  if (own_hi > P.n_blocks) own_hi = P.n_blocks;
  const bool has_chunk = chunk < P.n_chunks && own_lo < P.n_blocks;
  const uint64_t blk0 = chunk_blk0(first_owned, bpl, back, chunk);
  // The chunk's fresh start is the true DP boundary only at column 0 of the whole text.
  const bool exact_start = (blk0 == 0) && (P.flags & kScanTextStart);
  const int64_t x0 = exact_start ? -1 : (int64_t)(blk0 * 64 + P.m + P.k);

  const int k = (int)P.k;
  const uint32_t m = P.m;
  const uint32_t nwords = P.nwords;
  const uint32_t last_rows = m - 32 * (nwords - 1);
  const uint32_t last_word_init = last_rows == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> last_rows);
  const_u32_ptr row_tab = (const_u32_ptr)(P.row_tab);
  uint32_t pkw0[8];  // row table of word 0, resident in scalar registers
#pragma unroll
  for (int i = 0; i < 8; ++i) pkw0[i] = row_tab[i];

  // per-row carries between horizontally adjacent blocks, 32 rows per word, row r of a word at
  // bit 31-r, one LDS slot pair per word and lane.  Fresh start: every vertical delta on the left
  // edge is +1 (D[j][start] = j).
  // With overhang the chunk that starts at column 0 of the text gets the alpha left edge instead.
  const bool ov_seed = exact_start && (P.flags & kScanOverhang);
  for (uint32_t w = 0; w < nwords; ++w) {
    uint32_t hp0 = (w == nwords - 1) ? last_word_init : 0xFFFFFFFFu;
    if (ov_seed) hp0 = P.ov_tab[w];
    carry[(w * 2 + 0) * 64 + lane] = hp0;
    carry[(w * 2 + 1) * 64 + lane] = 0;
  }
  const uint32_t tail_pad = (P.flags & kScanOverhang) ? (uint32_t)'N' : (uint32_t)'X';

  // ---- staging geometry: instruction i of a stage loads, for tile row `owner`, the 16-byte
  // chunk that belongs into slot (lane % kSlots) of that row.  Slots are XOR-swizzled so that the
  // owners' ds_read_b128 of their own rows are bank-conflict free:
  //   SB = 2: slot = chunk ^ ((owner >> 1) & 7);  SB = 1: slot = chunk ^ ((owner >> 2) & 3).
  const uint64_t wave_blk0 = chunk_blk0(first_owned, bpl, back, wave_chunk0);
  const uint8_t* text_base = P.text + wave_blk0 * 64;
  uint32_t soff[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    const uint32_t owner = (uint32_t)i * kOwnersPerInstr + lane / kSlots;
    const uint32_t slot = lane % kSlots;
    const uint32_t j = slot ^ (SB == 2 ? ((owner >> 1) & 7u) : ((owner >> 2) & 3u));
    soff[i] = (uint32_t)((chunk_blk0(first_owned, bpl, back, wave_chunk0 + owner) - wave_blk0) * 64) + j * 16u;
  }
  // wave-uniform: can every staged byte of this wave be read without a bounds check?
  const uint64_t wave_last = chunk_blk0(first_owned, bpl, back, wave_chunk0 + 63) + P.n_iter + 2;
  const bool interior = wave_last * 64 <= P.text_len;
  // reading side: the lane's own row, logical chunks 4*sub + c
  const uint32_t fsw = SB == 2 ? ((lane >> 1) & 7u) : ((lane >> 2) & 3u);
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * kRowBytes + (((uint32_t)c ^ (fsw & 3u)) << 4);

  uint32_t st = kStDec;  // dec = true, amb = false
  EmitCtx ctx;
  ctx.cand = P.cand;
  ctx.cand_count = P.cand_count;
  ctx.text_len = P.text_len;
  ctx.global_offset = P.global_offset;
  ctx.cand_cap = P.cand_cap;
  ctx.k = P.k;
  ctx.flags = P.flags;
  ctx.alpha = P.alpha;
  ctx.ov_steps = (P.flags & kScanOverhang) ? P.ov_steps : 0u;
  ctx.text_begin = 0;
  ctx.tag = 0;

  unsigned long long cnt_rows = 0, cnt_blocks = 0, cnt_live = 0;
  const unsigned char* my_masks = mask_bytes + lane * 8;
  // bounded rows: where this wave starts testing (wave-uniform; beyond every pattern = never)
  uint32_t first_rows = (P.flags & kScanNoRowCut) ? 0x40000000u : kCutFirstRows;
  int minus_total = 0;                  // -1 deltas on the current block's left edge (none at a fresh start)

  for (uint32_t it = 0; it < P.n_iter; ++it) {
    const uint32_t sub = SB == 2 ? (it & 1u) : 0u;
    if (sub == 0) {
      // ---- stage SB blocks for each of the 64 lane chunks: kStageInstr x (64 lanes x 16 B) ----
      if (interior) {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) {
          const uint4 v = stream_load16<SASSY_NT_SCAN>(text_base + (uint64_t)it * 64 + soff[i]);
          *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) {
          const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + soff[i];
          uint4 v;
          if (off + 16 <= P.text_len) v = *reinterpret_cast<const uint4*>(P.text + off);
          else v = load_tail16(P.text, off, P.text_len, tail_pad);
          *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
        }
      }
    }

    // ---- the lane's own 64 text bytes -> profile masks -> LDS ----
    {
      const uint32_t hs = SB == 2 ? (((sub << 2) ^ (fsw & 4u)) << 4) : 0u;  // slot bit 2 = block of the pair
      uint32_t x[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs);
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
      }
      uint2 msk[NS];
      build_masks<PROFILE, NS>(x, P, msk);
#pragma unroll
      for (int s = 0; s < NS; ++s)
        *reinterpret_cast<uint2*>(mask_bytes + s * 512 + lane * 8) = msk[s];
    }

    // ---- the DP rows of this block (bounded: the wave stops at the first row below which no lane can
    // hold a cell <= k) ----
    const uint64_t b = blk0 + it;
    const bool active = has_chunk && b < own_hi;
    if ((it & 7u) == 7u && first_rows > 8u && first_rows <= m) first_rows -= 4u;  // now and then try an earlier row again
    DpWord V;
    int ds;  // cost at the block's left edge in the last row
    const bool ran_through = dp_block<PROFILE == (int)PROFILE_ASCII_BYTES, GC>(V, ds, my_masks, carry, lane, row_tab, pkw0, nwords, last_rows, last_word_init, k,
                                      !active, first_rows, minus_total, m, P.counters != nullptr && active, cnt_rows);

    // ---- last row of the block: anything <= k ? ----
    rep4 rr = make_rep4(0u, 0u, 0u, 0u);
    const uint64_t vp = ((uint64_t)V.vph << 32) | V.vpl, vm = ((uint64_t)V.vmh << 32) | V.vml;
    bool live = active && ran_through && row_maybe_live(ds, V, k);
    if (__any(live)) live = row_live_exact(V.vpl, V.vph, V.vml, V.vmh, ds, k) && live;
    if (__any(live)) rr = scan_blocks(ctx, live, vp, vm, ds, b, b >= own_lo, b + 1 == own_lo, x0, st);  // (the wave, block by block)
    if (active) {
      if (P.counters) cnt_blocks += 1;
      if (live) {
        if (P.counters) cnt_live += 1;
        st = rr.z & (kStDec | kStAmb);
      } else {
        st = kStDec;  // dec = true: a later <=k run can only be entered by a decrease; amb = false
      }
    }
    // (wave-uniform: the reports of all lanes go out with one atomic)
    if (__any((rr.x | rr.y | (rr.z & kRepBase)) != 0u)) (void)emit_reports(ctx, rr, vp, vm, ds, b);
  }

  if (chunk < P.n_chunks) {
    const uint32_t fin = (st & kStAmb) ? kStatePass : ((st & kStDec) ? kStateDecTrue : kStateDecFalse);
    P.chunk_state[chunk] = (uint8_t)fin;
    // the chunk that ends the buffer publishes what a following shard needs (control block tail)
    if (chunk + 1 == P.n_chunks) {
      uint32_t* tail = P.cand_count + kCtlTailWord;
      tail[0] = (uint32_t)own_lo; tail[1] = fin; tail[2] = 0u; tail[3] = 1u;
    }
  }
  if (P.counters) {
    atomicAdd(&P.counters[0], cnt_rows);
    atomicAdd(&P.counters[1], cnt_blocks);
    atomicAdd(&P.counters[3], cnt_live);
  }
}

// ====================================================================== K0: the prefilter
// Pigeonhole: cut the first n_pieces * piece_len pattern rows into n_pieces = k+1 disjoint pieces;
// an alignment with <= k edits leaves at least one piece untouched, so every cell <= k in the
// last DP row at column c implies an EXACT occurrence of some piece ending at a text position
// e in [c - (m+k), c].  This kernel streams over the text exactly like the DP kernel (same
// staging, same lane-parallel profile), but per block it only evaluates, for every piece, the
// bit-parallel exact-match word  E_p = AND_j (mask[slot(p,j)] << (piece_len-1-j))  with the bits
// shifted in from the previous block's masks, and records the blocks in which some piece ends.
// Blocks far from every recorded block cannot hold a cell <= k and never see the DP.
// (The piece test uses the same slot masks as the scan, so it is exact for every profile.)
// NPG: 0 = any number of pieces (row table in LDS), 1 / 2 = up to 4 / 8 pieces with the row table
// in scalar registers and all pieces advanced side by side (independent chains hide LDS latency).
template <int PROFILE, int NS, int SB, int NPG>
__global__ __launch_bounds__(256) void filter_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kRowBytes = 64u * SB;
  constexpr uint32_t kSlots = 4u * SB;
  constexpr uint32_t kOwnersPerInstr = 64u / kSlots;
  constexpr int kStageInstr = 4 * SB;
  constexpr uint32_t kTile = 64u * kRowBytes;
  constexpr uint32_t kTermTabBytes = 1024;  // LDS offset of the slot mask of every piece row
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t* term_off = reinterpret_cast<uint32_t*>(smem);
  unsigned char* wbase = smem + kTermTabBytes + (size_t)wave * P.lds_per_wave;
  unsigned char* tile = wbase;
  unsigned char* mask_bytes = wbase + kTile;  // two buffers of [NS][64] u64: this block / previous block

  const uint32_t n_terms = P.n_pieces * P.piece_len;
  {
    const_u32_ptr row_tab = (const_u32_ptr)(P.row_tab);
    for (uint32_t t = threadIdx.x; t < n_terms; t += blockDim.x)
      term_off[t] = ((row_tab[t >> 2] >> (8 * (t & 3))) & 0xFFu) << 8;
  }
  // previous-block masks of a chunk that starts at the buffer start: nothing matches before it
#pragma unroll
  for (int s = 0; s < 2 * NS; ++s) *reinterpret_cast<uint2*>(mask_bytes + s * 512 + lane * 8) = make_uint2(0u, 0u);
  __syncthreads();

  const uint64_t wave_chunk0 = ((uint64_t)blockIdx.x * kWavesPerGroup + wave) * kWave;
  if (wave_chunk0 >= P.n_chunks) return;  // wave-uniform
  const uint64_t chunk = wave_chunk0 + lane;
  const uint32_t bpl = P.bpl;
  const uint64_t first_owned = P.first_owned_block;
  const uint32_t back = 1u + (uint32_t)((first_owned + 1u) & 1u);  // previous block + evenness
  const uint64_t own_lo = first_owned + chunk * (uint64_t)bpl;
  uint64_t own_hi = own_lo + bpl;
  if (own_hi > P.n_blocks) own_hi = P.n_blocks;
  const bool has_chunk = chunk < P.n_chunks && own_lo < P.n_blocks;
  const uint64_t blk0 = chunk_blk0(first_owned, bpl, back, chunk);

  const uint64_t wave_blk0 = chunk_blk0(first_owned, bpl, back, wave_chunk0);
  const uint8_t* text_base = P.text + wave_blk0 * 64;
  uint32_t soff[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    const uint32_t owner = (uint32_t)i * kOwnersPerInstr + lane / kSlots;
    const uint32_t slot = lane % kSlots;
    const uint32_t j = slot ^ (SB == 2 ? ((owner >> 1) & 7u) : ((owner >> 2) & 3u));
    soff[i] = (uint32_t)((chunk_blk0(first_owned, bpl, back, wave_chunk0 + owner) - wave_blk0) * 64) + j * 16u;
  }
  const uint64_t wave_last = chunk_blk0(first_owned, bpl, back, wave_chunk0 + 63) + P.n_iter + 2;
  const bool interior = wave_last * 64 <= P.text_len;
  const uint32_t fsw = SB == 2 ? ((lane >> 1) & 7u) : ((lane >> 2) & 3u);
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * kRowBytes + (((uint32_t)c ^ (fsw & 3u)) << 4);

  const uint32_t q = P.piece_len;

  for (uint32_t it = 0; it < P.n_iter; ++it) {
    const uint32_t sub = SB == 2 ? (it & 1u) : 0u;
    if (sub == 0) {
      if (interior) {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) {
          const uint4 v = stream_load16<SASSY_NT_GENERIC>(text_base + (uint64_t)it * 64 + soff[i]);
          *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) {
          const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + soff[i];
          uint4 v;
          if (off + 16 <= P.text_len) v = *reinterpret_cast<const uint4*>(P.text + off);
          else v = load_tail16(P.text, off, P.text_len);
          *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
        }
      }
    }
    unsigned char* cur = mask_bytes + (it & 1u) * (NS * 512);
    const unsigned char* prv = mask_bytes + ((it & 1u) ^ 1u) * (NS * 512);
    {
      const uint32_t hs = SB == 2 ? (((sub << 2) ^ (fsw & 4u)) << 4) : 0u;
      uint32_t x[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs);
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
      }
      uint2 msk[NS];
      build_masks<PROFILE, NS>(x, P, msk);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<uint2*>(cur + s * 512 + lane * 8) = msk[s];
    }
    // ---- exact piece occurrences that end inside this block ----
    uint32_t hl = 0, hh = 0;
    const unsigned char* curl = cur + lane * 8;
    const unsigned char* prvl = prv + lane * 8;
    if constexpr (NPG > 0) {
      uint32_t sl[4 * NPG], sh[4 * NPG];
#pragma unroll
      for (int i = 0; i < 4 * NPG; ++i) { sl[i] = 0xFFFFFFFFu; sh[i] = 0xFFFFFFFFu; }
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        if ((uint32_t)j + 1u < q) {                      // wave-uniform
          const uint32_t sr = 32u - (q - 1u - (uint32_t)j);  // shift left by q-1-j as a funnel shift right
#pragma unroll
          for (int g = 0; g < NPG; ++g) {
            const uint32_t w = P.piece_tab[g][j];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
              const uint32_t off = ((w >> (8 * pp)) & 0xFFu) << 8;
              const uint2 c = *reinterpret_cast<const uint2*>(curl + off);
              const uint2 pv = *reinterpret_cast<const uint2*>(prvl + off);
              sl[4 * g + pp] &= __builtin_amdgcn_alignbit(c.x, pv.y, sr);
              sh[4 * g + pp] &= __builtin_amdgcn_alignbit(c.y, c.x, sr);
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < NPG; ++g) {
        const uint32_t w = P.piece_last[g];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const uint2 c = *reinterpret_cast<const uint2*>(curl + (((w >> (8 * pp)) & 0xFFu) << 8));
          hl |= sl[4 * g + pp] & c.x;
          hh |= sh[4 * g + pp] & c.y;
        }
      }
    } else {
      uint32_t t = 0;
      for (uint32_t p = 0; p < P.n_pieces; ++p) {
        uint32_t sl = 0xFFFFFFFFu, sh = 0xFFFFFFFFu;
        for (uint32_t j = 0; j + 1 < q; ++j, ++t) {
          const uint32_t off = term_off[t];
          const uint2 c = *reinterpret_cast<const uint2*>(curl + off);
          const uint2 pv = *reinterpret_cast<const uint2*>(prvl + off);
          const uint32_t sr = 32u - (q - 1u - j);
          sl &= __builtin_amdgcn_alignbit(c.x, pv.y, sr);
          sh &= __builtin_amdgcn_alignbit(c.y, c.x, sr);
        }
        const uint2 c = *reinterpret_cast<const uint2*>(curl + term_off[t]);  // last row of the piece: no shift
        ++t;
        hl |= sl & c.x;
        hh |= sh & c.y;
      }
    }
    const uint64_t b = blk0 + it;
    const bool evaluate = has_chunk && b >= own_lo && b < own_hi;
    if (evaluate && (hl | hh) != 0) {
      atomicOr(&P.hit_bitmap[b >> 6], 1ull << (b & 63));
    }
  }
}

// Rare path of filter_dna_kernel, kept out of line so that it does not shape the register allocation
// and scheduling of the streaming loop.  `bits`: end positions (bit i = text index 64 b + i) of exact
// occurrences of piece pp in block b.  The piece is known, so the blocks that can hold the END of a
// match around the occurrence are known exactly: with rem pattern rows behind the piece and <= k
// edits, a match that contains the occurrence ending at text position e ends in
// [e + rem - k, e + rem + k].  The blocks of those columns and of the one behind them are marked
// (not the block of the occurrence); K0b adds the warm-up in front.
// blocks [x, y] in which a match around the occurrences `bits` of a piece can end (see above)
__device__ __forceinline__ uint2 piece_end_blocks(uint64_t bits, uint64_t b, int64_t mirror_n_plus_q, int64_t rem, int64_t k,
                                                  uint64_t n_blocks) {
  int64_t e_lo = (int64_t)(b * 64) + __ffsll((long long)bits);        // first end position
  int64_t e_hi = (int64_t)(b * 64) + 64 - __clzll((long long)bits);   // last end position
  if (mirror_n_plus_q >= 0) {
    // a piece of the Rc strand's pattern, its string reversed: the occurrence T[f-q, f) is, in
    // the reversed text, that piece ending at reversed column n - f + q (bitmap: the Rc strand's)
    const int64_t r_lo = mirror_n_plus_q - e_hi, r_hi = mirror_n_plus_q - e_lo;
    e_lo = r_lo;
    e_hi = r_hi;
  }
  int64_t c_lo = e_lo + rem - k;
  // + 1: the report rule decides about an end position when it sees the next column
  const int64_t c_hi = e_hi + rem + k + 1;
  if (c_lo < 1) c_lo = 1;
  uint64_t blo = (uint64_t)(c_lo - 1) >> 6, bhi = (uint64_t)(c_hi - 1) >> 6;
  if (bhi >= n_blocks) bhi = n_blocks - 1;
  return make_uint2((uint32_t)blo, (uint32_t)bhi);
}
__device__ __noinline__ void mark_piece_ends(unsigned long long* bitmap, uint64_t bits, uint64_t b, int64_t mirror_n_plus_q,
                                             int64_t rem, int64_t k, uint64_t n_blocks) {
  const uint2 r = piece_end_blocks(bits, b, mirror_n_plus_q, rem, k, n_blocks);
  for (uint64_t x = r.x; x <= (uint64_t)r.y; ++x) atomicOr(&bitmap[x >> 6], 1ull << (x & 63));
}

// ====================================================================== K0 for Dna, linear streaming
// Same bit-plane evaluation as filter_dna_kernel, other data movement: a wave walks ONE contiguous text
// range, 128 consecutive blocks (8 KiB) per step, lane l taking blocks 2l and 2l+1 of the step.  The
// loads of a wave are then one contiguous 8 KiB read per step (the access pattern the HBM likes best,
// profiles/r01_stream_read.txt) instead of 64 streams bpl * 64 bytes apart, whose speed depends on how
// that stride and the lane count fall onto the channel mapping (host.hip: GeoTuner).  What a block
// needs from its predecessor -- the high halves of its two code planes -- comes from the neighbour
// lane by DPP (wave_shr:1; lane 0 gets the previous step's last block through the `old` operand), the
// second block of a lane from its own first.  Every block is evaluated exactly once; a wave primes its
// planes with one extra step in front of its range.
template <int NPG>
__global__ __launch_bounds__(256) void filter_dna_linear_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NP = 4 * NPG;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* tile = smem + (size_t)wave * 8192u;

  const uint64_t cover_lo = P.first_owned_block & ~1ull;  // rows are pairs of blocks = aligned 128-byte lines
  const uint64_t range = 128ull * P.lin_steps;
  const uint64_t w_lo = cover_lo + ((uint64_t)blockIdx.x * kWavesPerGroup + wave) * range;
  if (w_lo >= P.n_blocks) return;  // wave-uniform
  const uint64_t w_hi = w_lo + range;

  // staging: instruction i fetches the 128-byte rows of lanes 8i .. 8i+7 (1 KiB contiguous), the 16-byte
  // pieces of a row swizzled so that the owners' ds_read_b128 are conflict free (as in filter_dna_kernel)
  uint32_t soff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t owner = (uint32_t)i * 8u + lane / 8u;
    const uint32_t slot = lane % 8u;
    const uint32_t j = slot ^ ((owner >> 1) & 7u);
    soff[i] = owner * 128u + j * 16u;
  }
  const uint32_t fsw = (lane >> 1) & 7u;
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * 128u + (((uint32_t)c ^ (fsw & 3u)) << 4);

  const uint32_t q = P.piece_len;
  uint32_t nb0[NP], nb1[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    nb0[pp] = ~P.piece_bits[pp][0];
    nb1[pp] = ~P.piece_bits[pp][1];
  }
  // steps: one priming step in front of the range (unless the range starts the buffer), then lin_steps
  const bool prime = w_lo >= 128;
  const uint64_t base0 = prime ? w_lo - 128 : w_lo;
  const uint32_t n_steps = P.lin_steps + (prime ? 1u : 0u);
  const bool interior = (base0 + 128ull * n_steps) * 64 <= P.text_len;
  const uint8_t* text_base = P.text + base0 * 64;
  uint32_t carry0 = 0, carry1 = 0;  // plane high halves of the last block of the previous step

  uint4 nxt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    nxt[i] = make_uint4(0u, 0u, 0u, 0u);
    if (interior) nxt[i] = stream_load16<SASSY_NT_DNA>(text_base + soff[i]);
  }
  for (uint32_t st = 0; st < n_steps; ++st) {
    const uint64_t base = base0 + 128ull * st;
    if (base >= P.n_blocks) break;  // wave-uniform
    if (interior) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = nxt[i];
      if (st + 1 < n_steps) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          nxt[i] = stream_load16<SASSY_NT_DNA>(text_base + (uint64_t)(st + 1) * 8192 + soff[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint64_t off = base * 64 + soff[i];
        uint4 v;
        if (off + 16 <= P.text_len) v = *reinterpret_cast<const uint4*>(P.text + off);
        else v = load_tail16(P.text, off, P.text_len);
        *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
      }
    }
    // the planes of the lane's two blocks
    uint2 ta0, ta1, tb0, tb1;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const uint32_t hs = ((((uint32_t)sub << 2) ^ (fsw & 4u)) << 4);
      uint32_t x[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs);
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
      }
      if (sub == 0) { ta0 = bit_plane<1>(x); ta1 = bit_plane<2>(x); }
      else { tb0 = bit_plane<1>(x); tb1 = bit_plane<2>(x); }
    }
    // predecessor of block 2l: block 2l-1 = the neighbour lane's second block (lane 0: the previous step's)
    const uint32_t pa0 = (uint32_t)__builtin_amdgcn_update_dpp((int)carry0, (int)tb0.y, 0x138, 0xF, 0xF, false);  // wave_shr:1
    const uint32_t pa1 = (uint32_t)__builtin_amdgcn_update_dpp((int)carry1, (int)tb1.y, 0x138, 0xF, 0xF, false);
    carry0 = (uint32_t)__builtin_amdgcn_readlane((int)tb0.y, 63);
    carry1 = (uint32_t)__builtin_amdgcn_readlane((int)tb1.y, 63);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const uint2 t0 = sub == 0 ? ta0 : tb0, t1 = sub == 0 ? ta1 : tb1;
      const uint32_t prev0 = sub == 0 ? pa0 : ta0.y, prev1 = sub == 0 ? pa1 : ta1.y;
      uint32_t al[NP], ah[NP];
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) { al[pp] = 0xFFFFFFFFu; ah[pp] = 0xFFFFFFFFu; }
#pragma unroll
      for (int d = 0; d < 12; ++d) {
        if ((uint32_t)d < q) {  // wave-uniform
          uint32_t s0l, s0h, s1l, s1h;
          if (d == 0) {
            s0l = t0.x; s0h = t0.y; s1l = t1.x; s1h = t1.y;
          } else {
            s0l = __builtin_amdgcn_alignbit(t0.x, prev0, 32 - d);
            s0h = __builtin_amdgcn_alignbit(t0.y, t0.x, 32 - d);
            s1l = __builtin_amdgcn_alignbit(t1.x, prev1, 32 - d);
            s1h = __builtin_amdgcn_alignbit(t1.y, t1.x, 32 - d);
          }
          const uint32_t j = q - 1u - (uint32_t)d;
#pragma unroll
          for (int pp = 0; pp < NP; ++pp) {
            const uint32_t n0 = 0u - ((nb0[pp] >> j) & 1u);
            const uint32_t n1 = 0u - ((nb1[pp] >> j) & 1u);
            al[pp] = bitop3<0x60>(al[pp], s0l, n0);  // a & (b ^ c)
            ah[pp] = bitop3<0x60>(ah[pp], s0h, n0);
            al[pp] = bitop3<0x60>(al[pp], s1l, n1);
            ah[pp] = bitop3<0x60>(ah[pp], s1h, n1);
          }
        }
      }
      uint32_t hit = 0;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) hit |= al[pp] | ah[pp];
      const uint64_t b = base + 2ull * lane + (uint64_t)sub;
      const bool evaluate = b >= w_lo && b < w_hi && b >= P.first_owned_block && b < P.n_blocks;
      if (evaluate && hit != 0) {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          const uint64_t bits = ((uint64_t)ah[pp] << 32) | al[pp];
          if (bits != 0) {
            const bool mirror = (P.piece_mirror >> pp) & 1u;
            mark_piece_ends(mirror ? P.hit_bitmap_rc : P.hit_bitmap, bits, b,
                            mirror ? (int64_t)P.text_len + (int64_t)q : (int64_t)-1, (int64_t)P.piece_rem[pp], (int64_t)P.k,
                            P.n_blocks);
          }
        }
      }
    }
  }
}

// ====================================================================== K0 for many Dna patterns
// search_encoded_patterns with thousands of equal-length patterns (CRISPR guides): the text bytes,
// the two code bit planes and their q shifted copies are the same for every pattern, only the
// scalar piece bits differ.  So one pass over the text evaluates a whole batch of patterns: per
// block the planes and shifts are built once (kept in registers), then every pattern costs
// (k+1) * q * 4 v_bitop3 -- about 90 VALU for 20-mers at k = 2 against 800 for a full DP pass --
// and marks, in its own bitmap, the blocks a match around a piece occurrence can end in (as
// filter_dna_kernel does).  The chunk list / DP / rank / traceback stages then run per pattern.
// Q: piece length (compile time: the shifts, the bit positions of the piece rows and the unrolling
// depend on it); up to 8 pieces per pattern.
template <int SB, int Q>
__global__ __launch_bounds__(256) void filter_dna_multi_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kRowBytes = 64u * SB;
  constexpr uint32_t kSlots = 4u * SB;
  constexpr uint32_t kOwnersPerInstr = 64u / kSlots;
  constexpr int kStageInstr = 4 * SB;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* tile = smem + (size_t)wave * P.lds_per_wave;

  const uint64_t wave_chunk0 = ((uint64_t)blockIdx.x * kWavesPerGroup + wave) * kWave;
  if (wave_chunk0 >= P.n_chunks) return;  // wave-uniform
  const uint64_t chunk = wave_chunk0 + lane;
  const uint32_t bpl = P.bpl;
  const uint64_t first_owned = P.first_owned_block;
  const uint32_t back = 1u + (uint32_t)((first_owned + 1u) & 1u);
  const uint64_t own_lo = first_owned + chunk * (uint64_t)bpl;
  uint64_t own_hi = own_lo + bpl;
  if (own_hi > P.n_blocks) own_hi = P.n_blocks;
  const bool has_chunk = chunk < P.n_chunks && own_lo < P.n_blocks;
  const uint64_t blk0 = chunk_blk0(first_owned, bpl, back, chunk);
  const uint64_t wave_blk0 = chunk_blk0(first_owned, bpl, back, wave_chunk0);
  const uint8_t* text_base = P.text + wave_blk0 * 64;
  uint32_t soff[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    const uint32_t owner = (uint32_t)i * kOwnersPerInstr + lane / kSlots;
    const uint32_t slot = lane % kSlots;
    const uint32_t j = slot ^ (SB == 2 ? ((owner >> 1) & 7u) : ((owner >> 2) & 3u));
    soff[i] = (uint32_t)((chunk_blk0(first_owned, bpl, back, wave_chunk0 + owner) - wave_blk0) * 64) + j * 16u;
  }
  const uint64_t wave_last = chunk_blk0(first_owned, bpl, back, wave_chunk0 + 63) + P.n_iter + 2;
  const bool interior = wave_last * 64 <= P.text_len;
  const uint32_t fsw = SB == 2 ? ((lane >> 1) & 7u) : ((lane >> 2) & 3u);
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * kRowBytes + (((uint32_t)c ^ (fsw & 3u)) << 4);

  const uint32_t n_pieces = P.n_pieces;
  const_u32_ptr bits = (const_u32_ptr)(P.multi_bits);
  uint32_t prev0 = 0, prev1 = 0;

  // Two blocks per lane and iteration (the staged pair): the scalar work per piece row -- turning a
  // pattern bit into a 0 / ~0 word -- is shared by both, which keeps the loop VALU-bound.
  for (uint32_t it = 0; it < P.n_iter; it += 2) {
#pragma unroll
    for (int i = 0; i < kStageInstr; ++i) {
      const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + soff[i];
      uint4 v;
      if (interior) v = stream_load16<SASSY_NT_DNA>(text_base + (uint64_t)it * 64 + soff[i]);
      else if (off + 16 <= P.text_len) v = *reinterpret_cast<const uint4*>(P.text + off);
      else v = load_tail16(P.text, off, P.text_len);
      *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
    }
    // planes of the two blocks, shifted by d = 0 .. Q-1 with the previous block's bits shifted in
    uint32_t sl[2][2][Q + 1], sh[2][2][Q + 1];  // [block][plane][d]; d = Q only for the long pieces
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const uint32_t hs = (((uint32_t)blk << 2) ^ (fsw & 4u)) << 4;
      uint32_t x[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs);
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
      }
      const uint2 t0 = bit_plane<1>(x), t1 = bit_plane<2>(x);
      sl[blk][0][0] = t0.x; sh[blk][0][0] = t0.y; sl[blk][1][0] = t1.x; sh[blk][1][0] = t1.y;
#pragma unroll
      for (int d = 1; d <= Q; ++d) {
        sl[blk][0][d] = __builtin_amdgcn_alignbit(t0.x, prev0, 32 - d);
        sh[blk][0][d] = __builtin_amdgcn_alignbit(t0.y, t0.x, 32 - d);
        sl[blk][1][d] = __builtin_amdgcn_alignbit(t1.x, prev1, 32 - d);
        sh[blk][1][d] = __builtin_amdgcn_alignbit(t1.y, t1.x, 32 - d);
      }
      prev0 = t0.y;
      prev1 = t1.y;
    }
    const uint64_t b0 = blk0 + it;
    const bool eval0 = has_chunk && b0 >= own_lo && b0 < own_hi;
    const bool eval1 = has_chunk && it + 1 < P.n_iter && b0 + 1 >= own_lo && b0 + 1 < own_hi;

    // the 16 piece words of a pattern arrive with one scalar load, fetched one pattern ahead
    uint32_t wn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wn[i] = bits[i];
    for (uint32_t p = 0; p < P.multi_n; ++p) {
      uint32_t w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) w[i] = wn[i];
      const uint32_t pn = p + 1 < P.multi_n ? p + 1 : p;
#pragma unroll
      for (int i = 0; i < 16; ++i) wn[i] = bits[16 * pn + i];
      uint32_t marks = 0;
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) {
        if ((uint32_t)pp < n_pieces) {  // wave-uniform
          const int32_t nb0 = (int32_t)~w[2 * pp], nb1 = (int32_t)~w[2 * pp + 1];
          uint32_t al0 = 0xFFFFFFFFu, ah0 = 0xFFFFFFFFu, al1 = 0xFFFFFFFFu, ah1 = 0xFFFFFFFFu;
          // the piece words hold the row at distance d from the piece's end at bit d
          const bool is_long = (P.multi_long >> pp) & 1u;  // wave-uniform
#pragma unroll
          for (int d = 0; d <= Q; ++d) {
            if (d < Q || is_long) {  // wave-uniform
              const uint32_t n0s = (uint32_t)__builtin_amdgcn_sbfe(nb0, d, 1);  // 0 or ~0 (uniform)
              const uint32_t n1s = (uint32_t)__builtin_amdgcn_sbfe(nb1, d, 1);
              // into vector registers once: v_bitop3 with a scalar source operand issues at half rate
              uint32_t n0, n1;
              asm volatile("v_mov_b32 %0, %1" : "=v"(n0) : "s"(n0s));
              asm volatile("v_mov_b32 %0, %1" : "=v"(n1) : "s"(n1s));
              al0 = bitop3<0x60>(al0, sl[0][0][d], n0);
              ah0 = bitop3<0x60>(ah0, sh[0][0][d], n0);
              al1 = bitop3<0x60>(al1, sl[1][0][d], n0);
              ah1 = bitop3<0x60>(ah1, sh[1][0][d], n0);
              al0 = bitop3<0x60>(al0, sl[0][1][d], n1);
              ah0 = bitop3<0x60>(ah0, sh[0][1][d], n1);
              al1 = bitop3<0x60>(al1, sl[1][1][d], n1);
              ah1 = bitop3<0x60>(ah1, sh[1][1][d], n1);
            }
          }
          // Rare per lane, but some lane of the wave hits almost every time: keep this path short.
          // The blocks a match around the occurrence can end in (as in filter_dna_kernel), relative
          // to the lane's block pair, collected over the pieces and marked once per pattern.
          const int rem = (int)P.piece_rem[pp], kk = (int)P.k;
          if (eval0 && (al0 | ah0) != 0) {
            const uint64_t hb = ((uint64_t)ah0 << 32) | al0;
            const int r_lo = (__ffsll((long long)hb) + rem - kk - 1) >> 6;          // -1 .. 1
            const int r_hi = (64 - __clzll((long long)hb) + rem + kk) >> 6;         //  0 .. 2
            marks |= ((2u << (r_hi + 1)) - 1u) & ~((1u << (r_lo + 1)) - 1u);        // bit r+1: block b0 + r
          }
          if (eval1 && (al1 | ah1) != 0) {
            const uint64_t hb = ((uint64_t)ah1 << 32) | al1;
            const int r_lo = (__ffsll((long long)hb) + rem - kk - 1) >> 6;
            const int r_hi = (64 - __clzll((long long)hb) + rem + kk) >> 6;
            marks |= (((2u << (r_hi + 1)) - 1u) & ~((1u << (r_lo + 1)) - 1u)) << 1;  // relative to b0 + 1
          }
        }
      }
      if (marks) {
        unsigned long long* bm = P.hit_bitmap + (uint64_t)p * P.multi_stride;
        while (marks) {
          const int t = __ffs((int)marks) - 1;
          marks &= marks - 1u;
          const int64_t blk = (int64_t)b0 - 1 + t;
          if (blk >= 0 && (uint64_t)blk < P.n_blocks) atomicOr(&bm[blk >> 6], 1ull << (blk & 63));
        }
      }
    }
  }
}

// ====================================================================== K0 with a q-gram table
// Many pieces (k+1 > 8), or Iupac patterns: instead of evaluating every piece, every text position
// looks its q-gram up in a bit table of all 4^Q q-grams that some piece accepts (built on the host;
// ambiguous pattern letters are expanded).  The 2-bit code of a text byte is (c >> 1) & 3 -- exact
// for A C G T U in either case, which is also the Dna profile's own definition for every byte
// (src/profiles/dna.rs:19-40).  Under the Iupac profile other text bytes (N, R, ... or non-letters)
// match more than their code says, so a block that holds one, and the block after it, are simply
// recorded as hits (check_text).  The rolling code h lives in a register across the lane's
// consecutive blocks; the table sits in LDS (4^Q / 8 bytes, shared by the workgroup): byte
// h & (2^(2Q-3) - 1), bit h >> (2Q-3).  Cost per block: 64 x (6 VALU + one ds_read_u8), whatever
// the number of pieces.
template <int Q>
__global__ __launch_bounds__(256) void filter_table_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kTableBytes = 1u << (2 * Q - 3);
  constexpr uint32_t kRowBytes = 64u;   // SB = 1: more waves per CU, the kernel is VALU / LDS bound
  constexpr uint32_t kSlots = 4u;
  constexpr uint32_t kOwnersPerInstr = 16u;
  constexpr int kStageInstr = 4;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* table = smem;
  unsigned char* tile = smem + kTableBytes + (size_t)wave * 4096u;
  {
    const uint4* src = reinterpret_cast<const uint4*>(P.qgram_table);
    uint4* dst = reinterpret_cast<uint4*>(table);
    for (uint32_t x = threadIdx.x; x < kTableBytes / 16; x += blockDim.x) dst[x] = src[x];
  }
  __syncthreads();

  const uint64_t wave_chunk0 = ((uint64_t)blockIdx.x * kWavesPerGroup + wave) * kWave;
  if (wave_chunk0 >= P.n_chunks) return;  // wave-uniform
  const uint64_t chunk = wave_chunk0 + lane;
  const uint32_t bpl = P.bpl;
  const uint64_t first_owned = P.first_owned_block;
  const uint32_t back = 1u + (uint32_t)((first_owned + 1u) & 1u);  // previous block + evenness
  const uint64_t own_lo = first_owned + chunk * (uint64_t)bpl;
  uint64_t own_hi = own_lo + bpl;
  if (own_hi > P.n_blocks) own_hi = P.n_blocks;
  const bool has_chunk = chunk < P.n_chunks && own_lo < P.n_blocks;
  const uint64_t blk0 = chunk_blk0(first_owned, bpl, back, chunk);

  const uint64_t wave_blk0 = chunk_blk0(first_owned, bpl, back, wave_chunk0);
  const uint8_t* text_base = P.text + wave_blk0 * 64;
  uint32_t soff[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    const uint32_t owner = (uint32_t)i * kOwnersPerInstr + lane / kSlots;
    const uint32_t slot = lane % kSlots;
    const uint32_t j = slot ^ ((owner >> 2) & 3u);
    soff[i] = (uint32_t)((chunk_blk0(first_owned, bpl, back, wave_chunk0 + owner) - wave_blk0) * 64) + j * 16u;
  }
  const uint64_t wave_last = chunk_blk0(first_owned, bpl, back, wave_chunk0 + 63) + P.n_iter + 2;
  const bool interior = wave_last * 64 <= P.text_len;
  const uint32_t fsw = (lane >> 2) & 3u;
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * kRowBytes + (((uint32_t)c ^ fsw) << 4);
  const bool check_text = P.profile == PROFILE_IUPAC;  // wave-uniform

  uint32_t h = 0;          // rolling code of the last 16 text chars
  uint32_t bad_prev = 0;   // the previous block held a byte that is not A C G T U
  uint4 nxt[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    nxt[i] = make_uint4(0u, 0u, 0u, 0u);
    if (interior) nxt[i] = stream_load16<SASSY_NT_TABLE>(text_base + soff[i]);
  }

  for (uint32_t it = 0; it < P.n_iter; ++it) {
    if (interior) {
#pragma unroll
      for (int i = 0; i < kStageInstr; ++i) *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = nxt[i];
      if (it + 1 < P.n_iter) {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i)
          nxt[i] = stream_load16<SASSY_NT_TABLE>(text_base + (uint64_t)(it + 1) * 64 + soff[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < kStageInstr; ++i) {
        const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + soff[i];
        uint4 v;
        if (off + 16 <= P.text_len) v = *reinterpret_cast<const uint4*>(P.text + off);
        else v = load_tail16(P.text, off, P.text_len);
        *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
      }
    }
    uint32_t x[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c]);
      x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < 64; ++c) {
      const uint32_t code = __builtin_amdgcn_ubfe(x[c >> 2], 8 * (c & 3) + 1, 2);
      h = (h << 2) | code;
      const uint32_t byte = table[__builtin_amdgcn_ubfe(h, 0, 2 * Q - 3)];
      acc |= byte >> __builtin_amdgcn_ubfe(h, 2 * Q - 3, 3);
    }
    uint32_t bad = 0;
    if (check_text) {
      // bytes other than A C G T U (either case): compare with the letter their code stands for
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        const uint32_t sel = (x[d] >> 1) & 0x03030303u;
        const uint32_t up = x[d] & 0xDFDFDFDFu;
        const uint32_t e1 = __builtin_amdgcn_perm(0u, 0x47544341u, sel);  // 'A' 'C' 'T' 'G' by code
        const uint32_t e2 = __builtin_amdgcn_perm(0u, 0x47554341u, sel);  // 'A' 'C' 'U' 'G'
        bad |= (up ^ e1) & (up ^ e2);
      }
    }
    const uint64_t b = blk0 + it;
    const bool evaluate = has_chunk && b >= own_lo && b < own_hi;
    if (evaluate && ((acc & 1u) | bad | bad_prev) != 0)
      atomicOr(&P.hit_bitmap[b >> 6], 1ull << (b & 63));
    bad_prev = bad;
  }
}

// ====================================================================== K1-list: DP over a chunk list
// Same DP, same report rule, same seam bookkeeping as scan_kernel, but every lane takes its chunk
// (first block, end block, flags) from a descriptor list built from the prefilter's hit bitmap.
// The chunks are few and short, so each lane simply reads its own 64 bytes per block.
// The DP of one chunk per lane (descriptor d; lanes without one idle along): shared by list_kernel and by
// the fused filter (filter_dna_kernel<.., FUSED>), which runs it on the chunks its own wave has found.
// di: index of the lane's chunk in P.chunk_state (kNoStateSlot: the exit state is not recorded).
constexpr uint32_t kNoStateSlot = 0xFFFFFFFFu;
// WIN: the chunks are windows (kDescWindow): blocks at any byte offset, warm-up inside the owned blocks.
template <int PROFILE, int NS, bool WIN = false, bool GC = false>
__device__ __forceinline__ void list_lanes(const ScanParams& P, unsigned char* mask_bytes, uint32_t* carry, uint32_t lane,
                                           bool has_chunk, const ChunkDesc d, uint32_t di) {
  const uint64_t own_lo = d.own_lo, own_hi = d.own_hi;
  const bool clear_before = (d.flags & kDescClearBefore) != 0;
  // a chunk whose left neighbour block holds no cell <= k starts fresh at its own first block;
  // a continuation chunk (split of a long run) needs the warm-up blocks in front of it
  // per-text mode: the chunk is a whole text that starts at block own_lo
  const bool whole_text = (d.flags & kDescWholeText) != 0;
  const uint32_t shift = WIN ? (d.pad_ & 63u) : 0u;
  uint64_t blk0 = own_lo;
  if (!WIN && !clear_before && !whole_text) blk0 = own_lo > P.wb ? own_lo - P.wb : 0;
  const bool at_text_start = whole_text || (blk0 == 0 && shift == 0 && (P.flags & kScanTextStart));
  const bool exact_start = clear_before || at_text_start;
  const int64_t x0 = exact_start ? -1 : (int64_t)(blk0 * 64 + shift + P.m + P.k);
  // (windows: the chunk reports the end positions behind column start + (pad_ >> 8), scan_block's r0)
  const uint32_t rskip = WIN ? (uint32_t)std::max<int64_t>(0, (int64_t)(blk0 * 64 + shift + (d.pad_ >> 8)) - x0) & 0xFFFFu : 0u;
  // (windows may begin in the halo: the host drops the end positions in front of the first owned block)
  const uint32_t my_iters = has_chunk ? (uint32_t)(own_hi - blk0) : 0u;

  const int k = (int)P.k;
  const uint32_t m = P.m;
  const uint32_t nwords = P.nwords;
  const uint32_t last_rows = m - 32 * (nwords - 1);
  const uint32_t last_word_init = last_rows == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> last_rows);
  const_u32_ptr row_tab = (const_u32_ptr)(P.row_tab);
  uint32_t pkw0[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pkw0[i] = row_tab[i];
  // fresh start: every vertical delta on the left edge is +1; with overhang a chunk that starts at
  // column 0 of its text gets the alpha left edge instead
  const bool ov_seed = at_text_start && (P.flags & kScanOverhang);
  for (uint32_t w = 0; w < nwords; ++w) {
    uint32_t hp0 = (w == nwords - 1) ? last_word_init : 0xFFFFFFFFu;
    if (ov_seed) hp0 = P.ov_tab[w];
    carry[(w * 2 + 0) * 64 + lane] = hp0;
    carry[(w * 2 + 1) * 64 + lane] = 0;
  }
  // (a window has no warm-up block that would settle the plateau state: it is open until an exact column does)
  uint32_t st = (WIN && !exact_start) ? (kStDec | kStAmb) : kStDec;
  EmitCtx ctx;
  ctx.cand = P.cand;
  ctx.cand_count = P.cand_count;
  ctx.text_len = P.text_len;
  ctx.global_offset = P.global_offset;
  ctx.cand_cap = P.cand_cap;
  ctx.k = P.k;
  ctx.flags = P.flags;
  ctx.alpha = P.alpha;
  ctx.ov_steps = (P.flags & kScanOverhang) ? P.ov_steps : 0u;
  ctx.text_begin = 0;
  ctx.tag = 0;
  ctx.flags |= shift << kEmitShiftBit;
  if (WIN && P.stash != nullptr && !whole_text) ctx.flags |= kScanStash;
  if (whole_text) {  // this lane's text: its own column 0, its own end, its index on every report
    ctx.text_begin = own_lo * 64;
    ctx.text_len = P.texts_start[d.pad_] + P.texts_len[d.pad_];
    ctx.tag = d.pad_ << kCandTextShift;
  }
  unsigned long long cnt_rows = 0, cnt_blocks = 0, cnt_live = 0;
  const unsigned char* my_masks = mask_bytes + lane * 8;
  uint32_t first_rows = (P.flags & kScanNoRowCut) ? 0x40000000u : kCutFirstRows;
  int minus_total = 0;

  // the lane's next block is fetched while this one is computed (two or three waves per SIMD do not
  // hide a load that is issued and consumed in the same iteration)
  // Two copies of the block loop, chosen once per wave.  FAST: every block of every lane's chunk lies
  // whole inside a forward buffer, so the four 16-byte loads of the next block are unconditional,
  // straight-line code (idle lanes re-read their chunk's first block) and nothing waits for them until
  // the next iteration uses them: one TLB + HBM miss latency per block, hidden behind the DP of the
  // current block.  Otherwise (buffer tail, Rc strand read backwards, tiny texts) the general fetch.
  const bool lane_plain = !has_chunk || (P.rev_n == 0 && own_hi * 64 + shift <= P.text_len);
  const bool fast_wave = __all(lane_plain) && P.text_len >= 64;
  auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    auto fetch = [&](uint32_t step, uint32_t (&dst)[16]) {
      const bool on = step < my_iters;
      if constexpr (FAST) {
        const uint64_t blk = on ? blk0 + step : (has_chunk ? blk0 : 0);
        typedef typename std::conditional<WIN, u32x4_unaligned, uint4>::type vec16;
        const vec16* p = reinterpret_cast<const vec16*>(P.text + blk * 64 + (on || has_chunk ? shift : 0u));
        const vec16 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w;
        dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
        dst[8] = v2.x; dst[9] = v2.y; dst[10] = v2.z; dst[11] = v2.w;
        dst[12] = v3.x; dst[13] = v3.y; dst[14] = v3.z; dst[15] = v3.w;
      } else {
        fetch_block(P, on, blk0 + step, dst, shift);
      }
    };
    uint32_t xn[16];
    fetch(0, xn);
    for (uint32_t it = 0; __any(it < my_iters); ++it) {
      const bool active = it < my_iters;
      const uint64_t b = blk0 + it;
      uint32_t x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = xn[c];
      fetch(it + 1, xn);
      {
        uint2 msk[NS];
        build_masks<PROFILE, NS>(x, P, msk);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<uint2*>(mask_bytes + s * 512 + lane * 8) = msk[s];
      }
      if ((it & 7u) == 7u && first_rows > 8u && first_rows <= m) first_rows -= 4u;
      DpWord V;
      int ds;
      const bool ran_through = dp_block<PROFILE == (int)PROFILE_ASCII_BYTES, GC>(V, ds, my_masks, carry, lane, row_tab, pkw0, nwords, last_rows, last_word_init, k,
                                        !active, first_rows, minus_total, m, P.counters != nullptr && active, cnt_rows);
      rep4 rr = make_rep4(0u, 0u, 0u, 0u);
      const uint64_t vp = ((uint64_t)V.vph << 32) | V.vpl, vm = ((uint64_t)V.vmh << 32) | V.vml;
      bool live = active && ran_through && row_maybe_live(ds, V, k);
      if (__any(live)) live = row_live_exact(V.vpl, V.vph, V.vml, V.vmh, ds, k) && live;
      if (__any(live)) rr = scan_blocks(ctx, live, vp, vm, ds, b, b >= own_lo, b + 1 == own_lo, x0, st | (rskip << 8));
      if (active) {
        if (P.counters) cnt_blocks += 1;
        if (live) {
          if (P.counters) cnt_live += 1;
          st = rr.z & (kStDec | kStAmb);
        } else if (!WIN || (int64_t)((b + 1) * 64 + shift) > x0) {
          st = kStDec;  // no cell <= k in the block (a window's block that ends inside its warm-up says nothing)
        }
      }
      // (wave-uniform: the reports of all lanes go out with one atomic)
      if (__any((rr.x | rr.y | (rr.z & kRepBase)) != 0u)) {
        const uint32_t slot1 = emit_reports(ctx, rr, vp, vm, ds, b);
        if constexpr (WIN) {  // the block reported: its text goes with the reports (TextStash)
          if (slot1 != 0 && slot1 != 0xFFFFFFu && slot1 <= P.stash_cap) {
            TextStash* ts = P.stash + (slot1 - 1u);
            ts->base = b * 64 + shift;
            uint4* tt = reinterpret_cast<uint4*>(ts->text);
            tt[0] = make_uint4(x[0], x[1], x[2], x[3]);
            tt[1] = make_uint4(x[4], x[5], x[6], x[7]);
            tt[2] = make_uint4(x[8], x[9], x[10], x[11]);
            tt[3] = make_uint4(x[12], x[13], x[14], x[15]);
          }
        }
      }
    }
  };
  if (fast_wave) run(std::true_type{});
  else run(std::false_type{});
  if (has_chunk) {
    const uint32_t fin = (st & kStAmb) ? kStatePass : ((st & kStDec) ? kStateDecTrue : kStateDecFalse);
    if (di != kNoStateSlot) P.chunk_state[di] = (uint8_t)fin;
    // the chunk that reaches the end of the buffer publishes what a following shard needs
    if (WIN ? (own_hi * 64 + shift >= P.text_len) : (own_hi == P.n_blocks)) {
      uint32_t* tail = P.cand_count + kCtlTailWord;
      tail[0] = (uint32_t)own_lo; tail[1] = fin; tail[2] = d.flags; tail[3] = 1u;
    }
  }
  if (P.counters) {
    atomicAdd(&P.counters[0], cnt_rows);
    atomicAdd(&P.counters[1], cnt_blocks);
    atomicAdd(&P.counters[3], cnt_live);
  }
}

// GC: the per-row carries in global memory (P.carry_global), and a fixed number of waves that take the chunk list 64 at a time.
template <int PROFILE, int NS, bool GC = false>
__global__ __launch_bounds__(256) void list_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* wbase = smem + (size_t)wave * P.lds_per_wave;
  unsigned char* mask_bytes = wbase;                                        // [NS][64] u64
  uint32_t* carry = reinterpret_cast<uint32_t*>(wbase + NS * 512);          // [word][hp|hm][lane]
  const uint32_t gwave = blockIdx.x * (blockDim.x >> 6) + wave;            // (1 .. 4 waves per workgroup)
  if constexpr (GC) carry = P.carry_global + (size_t)gwave * P.nwords * 128u;

  uint32_t n_desc = *P.desc_count;
  if (n_desc > P.desc_cap) n_desc = P.desc_cap;
  if (n_desc <= P.list_words_max) return;  // few chunks of a multi-word pattern: list_words_kernel runs them
  const uint32_t wave_stride = GC ? gridDim.x * (blockDim.x >> 6) * kWave : 0u;
  for (uint32_t wave_first = gwave * kWave; wave_first < n_desc; wave_first += wave_stride) {  // (wave-uniform)
    const uint32_t di = wave_first + lane;
    const bool has_chunk = di < n_desc;
    ChunkDesc d;
    d.own_lo = d.own_hi = d.flags = d.pad_ = 0;
    if (has_chunk) d = P.desc[di];
    list_lanes<PROFILE, NS, false, GC>(P, mask_bytes, carry, lane, has_chunk, d, di);
    if constexpr (!GC) break;
  }
}

// Rare paths of the fused filter_dna_kernel, out of line like mark_piece_ends (inlined they cost the streaming
// loop 40 VGPRs = one wave per SIMD).  Everything by value in registers: a struct argument would go through
// scratch memory, and a kernel that owns a scratch segment starts its waves slower (0.55 -> 0.65 ms per launch).
//
// The end positions (columns) a match around the occurrences `bits` of a piece can have: with rem pattern rows
// behind the piece and <= k edits, [e + rem - k, e + rem + k] for an occurrence that ends at e, + 1 column (the
// report rule decides about a position when it sees the next one); relative to col_base (a lane's marks stay
// within a few thousand columns of its own blocks: 32 bits).
__device__ __noinline__ uint2 piece_end_cols(uint64_t bits, uint64_t b, int64_t rem, int64_t k, int64_t max_col, int64_t col_base) {
  const int64_t e_lo = (int64_t)(b * 64) + __ffsll((long long)bits);        // first end position
  const int64_t e_hi = (int64_t)(b * 64) + 64 - __clzll((long long)bits);   // last end position
  int64_t c_lo = e_lo + rem - k, c_hi = e_hi + rem + k + 1;
  if (c_lo < 1) c_lo = 1;
  if (c_hi > max_col) c_hi = max_col;
  return make_uint2((uint32_t)(c_lo - col_base), (uint32_t)(c_hi - col_base));
}
// The run of end positions a lane is collecting (columns relative to col_base): x = first (kRunNone: none),
// y = last, z = one past the last column the lane's queued windows cover, w = first column of the last queued one
// (bit 31 of w: kRunPressure -- the wave's queue is filling up; bit 30: kRunCont, see below).
constexpr uint32_t kRunNone = 0xFFFFFFFFu;
constexpr uint32_t kRunPressure = 0x80000000u;
constexpr uint32_t kRunMergeGap = 32;  // runs this close share a window
// Queues the window chunk for the columns [x, y]: the DP starts fresh at column `start` and is exact from start + mk on
// (mk = m + k), so the window [start, y] with start <= x - 1 - mk reports every end position of the run, plateau
// state included (unless the plateau reaches back beyond x - 1 without an exact cell of cost 0: such a report is
// conditional, scan_block).  Whole 64-column blocks ending at y: {first block, skip << 14 | blocks << 6 | byte shift},
// skip = x - 2 - start: the columns whose end positions the window does not report (warm-up, margin, rounding).
// A long run -- a run of N under Iupac, poly-A against poly-A, a microsatellite -- is no work for one lane: it leaves
// the lane in windows of kFuseSplitBlocks blocks of end positions each, as it grows (fuse_add_range), every window
// with its own warm-up; neighbouring windows share up to 63 exact columns, and what both report is dropped where the
// reports are ranked.
constexpr uint32_t kFuseSplitBlocks = 8;
// A window that continues a run -- behind a split, or at the very beginning of the lane's range, where the run may be
// the neighbour lane's going on -- begins inside cells <= k: how its plateau was entered lies in front of it.  It gets
// kFuseMargin more columns on the left: whatever rises, falls, reaches 0 or exceeds k in them settles the state before
// the window's own columns begin (their reports are the previous window's: copies, dropped).  What stays open -- a
// plateau of one cost > 0 that is flat for more than the margin -- is a conditional report, the classic chain's.
constexpr uint32_t kFuseMargin = 64;
constexpr uint32_t kFuseMaxBlocks = kFuseSplitBlocks + 16;  // (no window is longer: split length + the widest single range + warm-up + margin)
constexpr uint32_t kRunCont = 0x40000000u;  // (bit 30 of w) the pending run continues a window that was split off
// press_at: a lane whose entry gets this index or a later one raises kRunPressure -- the wave then runs the chunk DP
// over its queue at the next block pair instead of at the end of its text range (the queue never overflows: between
// two looks at the flag every lane adds at most two entries).
__device__ __forceinline__ rep4 fuse_emit(uint2* queue, uint32_t* qcount, uint32_t* fuse_word, uint32_t cap, uint32_t press_at,
                                          uint32_t mk, int64_t col_base, rep4 st, uint32_t x, uint32_t y) {
  // (65536: the lane's first own column, see col_base)
  const uint32_t margin = ((st.w & kRunCont) || x < 65536u + 2u * 64u) ? kFuseMargin : 0u;
  uint32_t nv = (y - x + margin + mk + 2u + 63u) / 64u;
  if (nv > kFuseMaxBlocks) atomicOr(fuse_word, kFuseOverflow);  // (cannot happen: runs are split before they get there)
  const int64_t end = col_base + (int64_t)y;
  int64_t start = end - 64 * (int64_t)nv;
  if (start < 0) {  // the buffer starts inside the window: whole blocks from byte 0
    start = 0;
    nv = (uint32_t)((end + 63) / 64);
  }
  // the window reports the end positions from x on: the exact columns in front of them (the margin, and what the
  // rounding to whole blocks adds) only settle the plateau state
  // (x - 2: the end position x - 1 is decided when column x is seen -- behind a split it is the last one of the window
  // in front, which could not decide it)
  const int64_t skip64 = (col_base + (int64_t)x - 2) - start;  // (>= mk - 1, or the window begins the buffer)
  const uint32_t skip = skip64 > 0 ? (uint32_t)skip64 : 0u;
  const uint32_t idx = atomicAdd(qcount, 1u);
  if (idx < cap) queue[idx] = make_uint2((uint32_t)(start >> 6), (skip << 14) | (nv << 6) | (uint32_t)(start & 63));
  else atomicOr(fuse_word, kFuseOverflow);
  st.z = y + 1u;
  st.w = x | ((idx >= press_at || (st.w & kRunPressure)) ? kRunPressure : 0u);  // (kRunCont: set by the caller that splits)
  return st;
}
// adds the columns [lo, hi] (lo = kRunNone: nothing to add, only queue the pending run)
__device__ __noinline__ rep4 fuse_add_range(uint2* queue, uint32_t* qcount, uint32_t* fuse_word, uint32_t cap, uint32_t press_at,
                                            uint32_t mk, int64_t col_base, uint32_t first_col, rep4 st, uint32_t lo, uint32_t hi) {
  if (lo == kRunNone) {
    if (st.x != kRunNone) {
      st = fuse_emit(queue, qcount, fuse_word, cap, press_at, mk, col_base, st, st.x, st.y);
      st.x = kRunNone;
    }
    return st;
  }
  if (lo < first_col) lo = first_col;  // the halo's end positions are not ours
  if (lo < (st.w & ~(kRunPressure | kRunCont))) {
    // columns in front of a window that is already queued (possible only when an occurrence's marks reach further
    // than a block: long patterns): the classic chain takes the search
    atomicOr(fuse_word, kFuseOverflow);
    return st;
  }
  if (lo < st.z) lo = st.z;  // covered by a queued window
  if (lo > hi) return st;
  if (st.x != kRunNone && lo <= st.y + kRunMergeGap) {
    st.x = min(st.x, lo);
    st.y = max(st.y, hi);
    if (st.y - st.x >= 64u * kFuseSplitBlocks + 64u) {  // a long run: its first kFuseSplitBlocks blocks leave now
      const uint32_t e = st.x + 64u * kFuseSplitBlocks - 1u;
      st = fuse_emit(queue, qcount, fuse_word, cap, press_at, mk, col_base, st, st.x, e);
      st.x = e + 1u;
      st.w |= kRunCont;
    }
    return st;
  }
  if (st.x != kRunNone) st = fuse_emit(queue, qcount, fuse_word, cap, press_at, mk, col_base, st, st.x, st.y);
  st.x = lo;
  st.y = hi;
  return st;
}

// ====================================================================== K0 for Dna: bit planes only
// The Dna code of a text byte is two bits ((c >> 1) & 3), so "text char i equals pattern char p"
// is  (T0 ^ ~P0) & (T1 ^ ~P1)  on the two code bit planes T0, T1 of the block with P0, P1 the
// replicated code bits of p -- wave-uniform values.  A piece occurrence ending at text bit i is the
// AND over its rows j of that term taken at bit i - (q-1-j).  The planes are shifted once per
// distance d = q-1-j (funnel shift with the previous block's planes, which this lane computed in
// its previous iteration and keeps in registers) and shared by all pieces; every term is then two
// v_bitop3_b32 (acc & (plane ^ scalar)) per 32-bit half.  No slot masks, no LDS besides the
// staging tile: per 64-byte block about 130 VALU ops for the planes + 4 * q for the shifts +
// 4 * (k+1) * q for the terms, which leaves the kernel bound by the HBM stream.
// Q: piece length (compile time: the shifts and the bit positions of the piece rows are immediates, the
// per-term scalars -- all-ones or zero -- come from one s_bfe_i32 each, issued next to the VALU work; as run-time
// values the compiler hoisted all 2 * Q * pieces of them out of the loop and spilled them: 80 v_readlane per block).
// NPG: 1 / 2 = up to 4 / 8 pieces (missing pieces repeat piece 0).
//
// FUSED: the whole scan in this one launch.  A lane that finds a piece occurrence knows the blocks a match
// around it can end in; instead of marking them in a global bitmap for a chunk builder and a list kernel to
// pick up (two more launches, each a latency chain of its own), it merges them into runs and queues the runs
// in its wave's LDS; when the wave has streamed its text range it runs the chunk DP (list_lanes: the list
// kernel's code) over what it queued, 64 chunks at a time, and appends the reports.  A run [lo, hi] becomes the
// chunk [lo - wb, hi] with a fresh start when the wb blocks in front of it are known to hold no cell <= k --
// unmarked by this lane and out of reach of the neighbour lanes' occurrences -- and otherwise the chunk [lo, hi]
// that warms up on the wb blocks in front (the list kernel's continuation chunk; a report whose plateau entry
// stays ambiguous carries kCandCond and sends the search to the classic chain).  Runs are per lane: the marks of
// an occurrence in a lane's first or last blocks can fall into the neighbour's range, and when the neighbour
// marks the same block both chunks report its end positions -- identical records, dropped where the reports
// are ranked (trace_wave_kernel).
//
// CHECK (fused only): an Iupac search whose PATTERN holds plain A C G T only, on a text that holds nothing else
// either.  There the profile's "sets intersect" is equality of the Dna codes: filter and chunk DP are the Dna ones.
// Every staged byte is checked on its way into the tile (one v_sad_u8 per dword against the letter its code stands
// for; U counts as other); one other byte anywhere sets the fuse flag and the classic Iupac chain takes the search
// (the host does not try again on that text).
// (CHECK) do these 16 text bytes hold anything but plain bases?  The letter a byte's code stands for
// (v_perm selectors 0 / 2 / 4 / 6 = twice the code -> A C T G) against the byte without its case bit, summed by v_sad_u8.
__device__ __forceinline__ uint32_t other_letters_sum16(const uint4 v, uint32_t diff) {
  const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t want = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, w4[d] & 0x06060606u);
    diff = __builtin_amdgcn_sad_u8(w4[d] & 0xDFDFDFDFu, want, diff);
  }
  return diff;  // (sums of absolute differences: 0 stays 0 only while every byte is a plain base)
}
__device__ __forceinline__ bool other_letters16(const uint4 v) { return other_letters_sum16(v, 0u) != 0u; }
//
// PAIR (fused only): the PAIRED filter for shapes whose k+1 pigeonhole pieces would be 5 or 6 rows -- the reference's own
// benchmark shape m = 23, k = 3 (benches/perf.rs:46-48), m = 32 with k = 4, 5 -- where an exact piece ends in every fourth
// to tenth block and the chunk DP behind the filter costs more than the streaming DP over every block.  The pattern is
// cut into S = ceil((k+1)/2) SUPER-PIECES of 2 Q rows instead (NPG = S here): of k edits one super-piece holds at most
// floor(k / S) = 1, so one of its two halves ("sub-pieces" A = first Q rows, B = next Q rows) occurs exactly and the
// other one with at most one edit RIGHT NEXT TO IT in the text.  Stage 1 is the bit-plane test of the 2 S sub-pieces
// (A-type pieces are detected Q + 2 columns late, so that the text their B lies in is in the lane's registers); stage 2
// takes every occurrence, in the lane that found it, from the planes it already holds: the Q + 1 characters behind A /
// in front of B against the sibling sub-piece with offsets -1, 0, +1 (one substitution, one skipped pattern row, one
// extra text character -- mismatch masks, first mismatch, what lies beyond it).  An occurrence whose sibling fails is
// dropped; 4 % of them survive on random text, and a window chunk is queued in every hundredth block instead of every
// fourth.  Exact: a match with <= k edits always has such a pair (pigeonhole on the super-pieces), and a pair whose
// detection column lies behind the last block of the buffer (A in the last Q + 2 columns) belongs to a match that ends
// in the last k + 2 columns: those are always searched.
// DPNS (CHECK only): slot masks the chunk DP builds -- 4 for a pattern of plain bases, 8 when the rows BEHIND the filter's
// pieces hold ambiguity letters (a CRISPR guide: 20 bases + NGG; the pieces themselves are plain, so the filter is the
// same): masks 8 x 512 B + carries of at most four pattern words fill the tile up to the saved segment state.
template <int Q, int NPG, bool FUSED, bool CHECK = false, bool PAIR = false, int DPNS = 4>
// (CHECK: four waves per SIMD are asked for -- left to itself the compiler settles for three, 0.59 instead of 0.52 ms)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CHECK || PAIR) ? 4 : 1))) void filter_dna_kernel(const ScanParams P) {
  static_assert(!CHECK || FUSED, "the text check exists in the fused launch only");
  static_assert(!PAIR || (FUSED && 2 * Q + 2 <= 31), "the paired filter exists in the fused launch only; its look-back stays inside one plane half");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SB = 2;
  constexpr uint32_t kRowBytes = 64u * SB;
  constexpr uint32_t kSlots = 4u * SB;
  constexpr uint32_t kOwnersPerInstr = 64u / kSlots;
  constexpr int kStageInstr = 4 * SB;
  constexpr int NP = PAIR ? 2 * NPG : 4 * NPG;
  // PAIR: A-type sub-pieces (even index) are detected DL columns late: the Q + 1 characters behind them then lie in front
  // of the detection column, like the characters in front of a B (one v_alignbit by the column index fetches either)
  constexpr int DL = Q + 2;
  constexpr int ND = PAIR ? Q + DL : Q;      // shift distances the piece rows are taken at
  constexpr uint32_t kTile = 64u * kRowBytes;
  typedef const ScanParams __attribute__((address_space(4)))* kparams_ptr;

  // FUSED: the wave streams its text range in SEGMENTS.  A segment ends when the range is done or when the wave's chunk
  // queue is filling up (kRunPressure); then the wave runs the chunk DP over what it queued -- the DP's masks and carries
  // take the place of the text tile -- and streams on.  What a segment needs from the one before: the next iteration,
  // the previous block's plane halves, the run the lane is collecting.  Everything else is derived again, inside the
  // loop, from a pointer to the launch parameters that is opaque to the optimiser, so that nothing of the chunk DP is
  // hoisted in front of the streaming loop and held there.
  // (The per-lane part of it waits in LDS while the DP runs -- six words per lane in the 2 KiB of the tile the DP's
  // masks and carries, at most eight pattern words, leave free -- and the lane index comes from v_mbcnt, the wave index
  // from a scalar: kept in vector registers across the DP they were spilled to scratch memory, and a kernel that owns a
  // scratch segment starts its waves slower.)
  constexpr uint32_t kSegState = 6144u;            // tile offset of the saved state: [7][64] u32
  uint32_t seg_it = 0;                             // wave-uniform, even
  bool first_segment = true;
  const uint32_t wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (;;) {
  kparams_ptr Pk = (kparams_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  uint32_t wave = __builtin_amdgcn_readfirstlane(wave_s);
  if constexpr (FUSED) asm volatile("" : "+s"(Pk), "+v"(lane), "+s"(wave));
  unsigned char* tile = smem + (size_t)wave * Pk->lds_per_wave;
  // FUSED: the wave's chunk queue behind the tile: fuse_queue_cap entries {first block, blocks << 6 | byte shift}, then
  // the count (16 bytes)
  uint2* queue = reinterpret_cast<uint2*>(tile + kTile);
  uint32_t* qcount = reinterpret_cast<uint32_t*>(tile + kTile + (size_t)Pk->fuse_queue_cap * 8u);
  const uint64_t group = (uint64_t)blockIdx.x + Pk->group_offset;
  const uint32_t bpl = Pk->bpl;
  const uint64_t first_owned = Pk->first_owned_block;
  const uint32_t back = 1u + (uint32_t)((first_owned + 1u) & 1u);  // previous block + evenness
  const uint32_t fsw = (lane >> 1) & 7u;
  uint32_t rc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rc[c] = lane * kRowBytes + (((uint32_t)c ^ (fsw & 3u)) << 4);
  // ~P0 / ~P1 of every piece row as bit masks (bit j = row j of the piece), wave-uniform
  uint32_t nb0[NP], nb1[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    nb0[pp] = ~Pk->piece_bits[pp][0];
    nb1[pp] = ~Pk->piece_bits[pp][1];
  }
  uint32_t u_bpl = bpl;
  uint64_t u_first = first_owned + group * 256ull * bpl;
  const uint32_t lc0 = wave * kWave;  // the wave's first lane chunk inside the unit
  if (u_first + (uint64_t)lc0 * u_bpl >= Pk->n_blocks) break;  // wave-uniform: nothing for this wave
  unsigned long long probe_t0 = 0;
  if constexpr (FUSED) {
    if (first_segment) {
      if (lane == 0) *qcount = 0;
    }
    if (Pk->fused & 2u) probe_t0 = wall_clock64();
  }
  const uint32_t n_iter = Pk->n_iter - bpl + u_bpl;
  const uint64_t own_lo = u_first + (uint64_t)(lc0 + lane) * u_bpl;
  uint64_t own_hi = own_lo + u_bpl;
  if (own_hi > Pk->n_blocks) own_hi = Pk->n_blocks;
  const bool has_chunk = own_lo < Pk->n_blocks;
  const int64_t col_base = (int64_t)(own_lo * 64) - 65536;  // FUSED: origin of the lane's relative columns
  const uint64_t blk0 = chunk_blk0(u_first, u_bpl, back, lc0 + lane);

  const uint64_t wave_blk0 = chunk_blk0(u_first, u_bpl, back, lc0);
  const uint8_t* text_base = Pk->text + wave_blk0 * 64;
  uint32_t soff[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    const uint32_t owner = (uint32_t)i * kOwnersPerInstr + lane / kSlots;
    const uint32_t slot = lane % kSlots;
    const uint32_t j = slot ^ ((owner >> 1) & 7u);
    soff[i] = (uint32_t)((chunk_blk0(u_first, u_bpl, back, lc0 + owner) - wave_blk0) * 64) + j * 16u;
  }
  const uint64_t wave_last = chunk_blk0(u_first, u_bpl, back, lc0 + 63) + n_iter + 2;
  const bool interior = wave_last * 64 <= Pk->text_len;
  uint32_t prev0 = 0, prev1 = 0;  // high halves of the previous block's planes

  // FUSED: the run of match-end columns this lane is collecting, and the end of the last window it queued
  rep4 run = make_rep4(kRunNone, 0u, 0u, 0u);
  if constexpr (FUSED) {
    if (!first_segment) {
      const uint32_t* sv = reinterpret_cast<const uint32_t*>(tile + kSegState) + lane;
      prev0 = sv[0]; prev1 = sv[64];
      run = make_rep4(sv[128], sv[192], sv[256], sv[320]);
    }
  }
  const uint32_t press_at = Pk->fuse_press;

  // software pipeline: the loads of the next staging step are in flight while this one is processed
  uint4 nxt[kStageInstr];
#pragma unroll
  for (int i = 0; i < kStageInstr; ++i) {
    nxt[i] = make_uint4(0u, 0u, 0u, 0u);
    if (interior && seg_it < n_iter) nxt[i] = stream_load16<SASSY_NT_DNA>(text_base + (uint64_t)seg_it * 64 + soff[i]);
  }
  uint32_t npure = 0;  // (CHECK) how many blocks of nothing but N the lane has just walked over
  if constexpr (CHECK) {
    if (!first_segment) npure = (reinterpret_cast<const uint32_t*>(tile + kSegState) + lane)[384];
  }

  uint32_t it = seg_it;
  for (; it < n_iter; ++it) {
    const uint32_t sub = it & 1u;
    if (sub == 0) {
      if (interior) {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = nxt[i];
        if (it + SB < n_iter) {
#pragma unroll
          for (int i = 0; i < kStageInstr; ++i)
            nxt[i] = stream_load16<SASSY_NT_DNA>(text_base + (uint64_t)(it + SB) * 64 + soff[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < kStageInstr; ++i) {
          const uint64_t off = wave_blk0 * 64 + (uint64_t)it * 64 + soff[i];
          uint4 v;
          if (off + 16 <= Pk->text_len) v = *reinterpret_cast<const uint4*>(Pk->text + off);
          else v = load_tail16(Pk->text, off, Pk->text_len);  // (bytes behind the end of the text read as 'X')
          *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
        }
      }
    }
    uint2 t0, t1;
    uint32_t dsum = 0;  // (CHECK) != 0: the block holds other letters
    {
      const uint32_t hs = (((sub << 2) ^ (fsw & 4u)) << 4);
      uint32_t x[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs);
        x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
      }
      t0 = bit_plane<1>(x);  // code bit 0
      t1 = bit_plane<2>(x);  // code bit 1
      // (CHECK) Other letters are handled where they lie, by the lane that owns the block: does it hold anything but
      // plain bases?  (four chains of sums of absolute differences against the letter each byte's code stands for; the
      // rare path below sorts out which 16-byte pieces they are in.)  The staging lanes used to check what they loaded
      // and tell the owners through LDS: a vote, a call and a word that lives across the block pair -- 0.54 -> 0.65 ms.
      if constexpr (CHECK) {
        uint32_t d4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const uint32_t want = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, x[c] & 0x06060606u);
          d4[c & 3] = __builtin_amdgcn_sad_u8(x[c] & 0xDFDFDFDFu, want, d4[c & 3]);
        }
        dsum = (d4[0] | d4[1]) | (d4[2] | d4[3]);
      }
    }
    // the piece rows' bits, opaque to the optimiser inside the loop: what is derived from them below (one scalar
    // per term) is computed here, per iteration, on the scalar unit, instead of living in 2 * Q * pieces registers
    uint32_t b0[NP], b1[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      b0[pp] = nb0[pp];
      b1[pp] = nb1[pp];
      asm volatile("" : "+s"(b0[pp]), "+s"(b1[pp]));
    }
    uint32_t al[NP], ah[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) { al[pp] = 0xFFFFFFFFu; ah[pp] = 0xFFFFFFFFu; }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      if (PAIR && d >= Q && d < DL) continue;  // (no sub-piece row sits at these distances)
      if constexpr (PAIR) __builtin_amdgcn_sched_barrier(0);
      uint32_t s0l, s0h, s1l, s1h;
      if (d == 0) {
        s0l = t0.x; s0h = t0.y; s1l = t1.x; s1h = t1.y;
      } else {
        s0l = __builtin_amdgcn_alignbit(t0.x, prev0, 32 - d);
        s0h = __builtin_amdgcn_alignbit(t0.y, t0.x, 32 - d);
        s1l = __builtin_amdgcn_alignbit(t1.x, prev1, 32 - d);
        s1h = __builtin_amdgcn_alignbit(t1.y, t1.x, 32 - d);
      }
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        // the piece row whose char sits d bits left of the detection column (PAIR: B-type pieces -- odd -- end there,
        // A-type pieces ended DL columns earlier)
        int j = Q - 1 - d;
        if constexpr (PAIR) {
          if ((pp & 1) == 0) j += DL;
          if (j < 0 || j >= Q) continue;
        }
        const uint32_t n0 = (uint32_t)__builtin_amdgcn_sbfe((int)b0[pp], j, 1);  // all ones iff bit j is set
        const uint32_t n1 = (uint32_t)__builtin_amdgcn_sbfe((int)b1[pp], j, 1);
        al[pp] = bitop3<0x60>(al[pp], s0l, n0);  // a & (b ^ c)
        ah[pp] = bitop3<0x60>(ah[pp], s0h, n0);
        al[pp] = bitop3<0x60>(al[pp], s1l, n1);
        ah[pp] = bitop3<0x60>(ah[pp], s1h, n1);
      }
    }
    if constexpr (PAIR) {
      // Stage 2: the sibling sub-piece of an occurrence, with at most one edit, right next to it -- for the FIRST occurrence
      // of every sub-piece in the block (a second one of the same sub-piece in the same block stays: a window too many in
      // one lane and block of five hundred).  No loop, no branch: all sub-pieces are tested at once, one 8-bit field
      // of a word each.
      //   w0 / w1: the code planes of the Q + 1 characters that follow an A (read forwards) / precede a B (read
      //   backwards), bit j of the field = the character at distance j; y0 / y1: the sibling's rows in that reading
      //   order (the host packs them); x0 / xm / xp: where sibling row j differs from the character at distance j / j - 1
      //   / j + 1.  With f = the first set bit of x0: one substitution <=> x0 has no other bit; row f skipped <=> xm is
      //   clear beyond f; an extra character in front of row f <=> xp is clear from f on (an earlier edit position can only
      //   do better where x0 is clear anyway).  Field arithmetic: bit 7 of every field is the guard the per-field
      //   negations, decrements and "is it zero" sums carry into.
      constexpr uint32_t F1 = 0x01010101u, F7 = 0x7F7F7F7Fu, F8 = 0x80808080u;
      constexpr uint32_t YM4 = ((1u << Q) - 1u) * F1;
      constexpr int NW = (NP + 3) / 4;
      uint32_t w0[NW], w1[NW];
#pragma unroll
      for (int wd = 0; wd < NW; ++wd) w0[wd] = w1[wd] = 0u;
      // (the scheduler is kept from weaving stage 2 into stage 1 and the sub-pieces into one another: left to itself it
      // holds every shifted plane and every sub-piece's intermediate values at once -- 165 VGPRs, three waves per SIMD)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        if (pp) __builtin_amdgcn_sched_barrier(0);
        const bool low = al[pp] != 0u;
        const uint32_t w = low ? al[pp] : ah[pp];
        uint32_t sh;  // column inside the half (no occurrence: -1 -- the field is computed and never used)
        asm("v_ffbl_b32 %0, %1" : "=v"(sh) : "v"(w));
        // the 32 columns in front of the detection column: bit 31 = the column right in front of it
        const uint64_t s0 = low ? pair64(prev0, t0.x) : pair64(t0.x, t0.y);
        const uint64_t s1 = low ? pair64(prev1, t1.x) : pair64(t1.x, t1.y);
        uint32_t e0 = (uint32_t)(s0 >> (sh & 31u)), e1 = (uint32_t)(s1 >> (sh & 31u));
        if (pp & 1) {
          e0 = __builtin_amdgcn_ubfe(__builtin_bitreverse32(e0), Q - 1, Q + 1);
          e1 = __builtin_amdgcn_ubfe(__builtin_bitreverse32(e1), Q - 1, Q + 1);
        } else {
          e0 = __builtin_amdgcn_ubfe(e0, 31 - Q, Q + 1);
          e1 = __builtin_amdgcn_ubfe(e1, 31 - Q, Q + 1);
        }
        w0[pp >> 2] |= e0 << (8 * (pp & 3));
        w1[pp >> 2] |= e1 << (8 * (pp & 3));
      }
      uint32_t fail[NW];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int wd = 0; wd < NW; ++wd) {
        const uint32_t y0 = Pk->pair_y[2 * wd], y1 = Pk->pair_y[2 * wd + 1];
        const uint32_t x0 = ((w0[wd] ^ y0) | (w1[wd] ^ y1)) & YM4;
        const uint32_t xp = (((w0[wd] >> 1) ^ y0) | ((w1[wd] >> 1) ^ y1)) & YM4;
        const uint32_t xm = (((w0[wd] << 1) ^ y0) | ((w1[wd] << 1) ^ y1)) & YM4;
        const uint32_t first = x0 & ((x0 ^ F7) + F1);
        const uint32_t below = (first | F8) - F1;       // (x0 == 0: all seven bits)
        const uint32_t sub_f = x0 ^ first, del_f = xm & ~(below | first), ins_f = xp & ~below;
        fail[wd] = (sub_f + F7) & (del_f + F7) & (ins_f + F7);  // bit 7 of a field: none of the three ways fits
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        // a failed sibling: the sub-piece's first occurrence -- the lowest set bit of its 64 columns -- leaves the mask
        const uint32_t keep = ~(uint32_t)__builtin_amdgcn_sbfe((int)fail[pp >> 2], 8 * (pp & 3) + 7, 1);
        const uint64_t less = lshl_add_u64<0>(pair64(al[pp], ah[pp]), ~0ull);  // mask - 1
        al[pp] &= lo32(less) | keep;
        ah[pp] &= hi32(less) | keep;
      }
    }
    prev0 = t0.y;
    prev1 = t1.y;
    uint32_t hit = 0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) hit |= al[pp] | ah[pp];
    // (CHECK: this block's dirty pieces; a lane that counts blocks of N looks at every block -- the rare path keeps the count)
    if constexpr (CHECK) hit |= dsum | npure;
    const uint64_t b = blk0 + it;
    const bool evaluate = has_chunk && b >= own_lo && b < own_hi;
    if (evaluate && hit != 0) {
      if constexpr (FUSED) {
        // (CHECK) The inside of a run of N: a block of 64 N whose last wb blocks were N as well.  Every alignment that
        // ends in it runs over N only -- cost 0, one plateau, nothing to report under the report rule and nothing to
        // learn (a cell of cost 0 settles the plateau state wherever the DP begins).  Such blocks are not queued at
        // all: a run of N costs the windows around its two ends, whatever its length (a genome's centromere gaps are
        // megabases).  Lists of ALL end positions <= k have every one of them: no short cut there.
        bool n_inside = false, n_leaving = false;
        uint32_t dp = 0;  // (CHECK) the block's 16-byte pieces that hold other letters
        if constexpr (CHECK) {
          const bool was_long = npure >= Pk->wb;
          bool pure = false;
          if (dsum != 0u) {  // (the block, once more, from the tile: which pieces, and is it nothing but N?)
            const uint32_t hs2 = (((sub << 2) ^ (fsw & 4u)) << 4);
            uint32_t acc = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const uint4 v = *reinterpret_cast<const uint4*>(tile + rc[c] + hs2);
              if (other_letters16(v)) dp |= 1u << c;
              acc |= ((v.x & 0xDFDFDFDFu) ^ 0x4E4E4E4Eu) | ((v.y & 0xDFDFDFDFu) ^ 0x4E4E4E4Eu) |
                     ((v.z & 0xDFDFDFDFu) ^ 0x4E4E4E4Eu) | ((v.w & 0xDFDFDFDFu) ^ 0x4E4E4E4Eu);
            }
            pure = acc == 0u && !(Pk->flags & kScanAllMinima) && b + 1 < Pk->n_blocks;
          }
          n_inside = pure && was_long;
          n_leaving = !pure && was_long;
          npure = pure ? (npure < 0xFFFFu ? npure + 1u : npure) : 0u;
        }
        uint32_t lo = kRunNone, hi = 0;
        if ((n_inside && b + 1 == own_hi) || n_leaving) {
          // leaving: the first block behind the run -- the plateau of cost 0 ends at its first column at the earliest: the
          // window reports from there on (it begins, with warm-up and margin, inside the run: settled at once).
          // The lane's range ends inside the run: should the run end there too, the plateau's last position is the next
          // lane's first, which knows nothing of the run -- the window around that border is queued here.
          const uint64_t bb = n_leaving ? b : b + 1;
          int64_t c_hi = (int64_t)(bb * 64) + (int64_t)Pk->m + (int64_t)Pk->k + 1;
          if (c_hi > (int64_t)(Pk->n_blocks * 64)) c_hi = (int64_t)(Pk->n_blocks * 64);
          lo = (uint32_t)((int64_t)(bb * 64) - col_base);
          hi = (uint32_t)(c_hi - col_base);
        }
        if (!n_inside) {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          const uint64_t bits = ((uint64_t)ah[pp] << 32) | al[pp];
          if (bits != 0) {
            // (PAIR: an A-type piece's rem counts from its detection column -- it may be -1)
            const uint2 r = piece_end_cols(bits, b, (int64_t)(int32_t)Pk->piece_rem[pp], (int64_t)Pk->k, (int64_t)(Pk->n_blocks * 64), col_base);
            lo = min(lo, r.x);
            hi = max(hi, r.y);
          }
        }
        if constexpr (CHECK) {
          if (dp != 0) {
            // a match that touches text byte t ends in [t + 1, t + m + k] (+ 1 column: the report rule decides about a
            // position when it sees the next one); t = the pieces' first .. last byte
            const int64_t t_first = (int64_t)(b * 64) + 16 * (__ffs((int)dp) - 1);
            const int64_t t_last = (int64_t)(b * 64) + 16 * (31 - __clz((int)dp)) + 15;
            int64_t c_hi = t_last + (int64_t)Pk->m + (int64_t)Pk->k + 1;
            if (c_hi > (int64_t)(Pk->n_blocks * 64)) c_hi = (int64_t)(Pk->n_blocks * 64);
            lo = min(lo, (uint32_t)(t_first + 1 - col_base));
            hi = max(hi, (uint32_t)(c_hi - col_base));
          }
        }
        }
        const int64_t fc = (int64_t)(Pk->dp_first_owned * 64) - col_base;
        if (lo != kRunNone)
          run = fuse_add_range(queue, qcount, Pk->cand_count + kCtlFuseWord, Pk->fuse_queue_cap, press_at, Pk->m + Pk->k, col_base,
                               fc > 0 ? (uint32_t)fc : 0u, run, lo, hi);
      } else {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          const uint64_t bits = ((uint64_t)ah[pp] << 32) | al[pp];
          if (bits != 0) {
            const bool mirror = (Pk->piece_mirror >> pp) & 1u;
            mark_piece_ends(mirror ? Pk->hit_bitmap_rc : Pk->hit_bitmap, bits, b,
                            mirror ? (int64_t)Pk->text_len + (int64_t)Q : (int64_t)-1, (int64_t)Pk->piece_rem[pp], (int64_t)Pk->k,
                            Pk->n_blocks);
          }
        }
      }
    }
    if constexpr (FUSED) {
      // the queue is filling up: the block pair is done with the tile -- the chunk DP may have it
      if (sub == 1u && __any((run.w & kRunPressure) != 0u)) { ++it; break; }
    }
  }
  if constexpr (!FUSED) break;

  if constexpr (FUSED) {
    // (one wave: its LDS operations complete in order; the fence keeps the compiler from moving the read up)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t nq = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t*>(qcount));
    const bool streamed = it >= n_iter;
    bool last = false;
    if (streamed && nq + 64u <= Pk->fuse_queue_cap) {  // the runs the lanes still hold (one entry each at most), then the last pass
      if constexpr (PAIR) {
        // An A-type sub-piece in the buffer's last DL columns would be detected behind the last block.  The match
        // around it ends in the last k + 1 + DL - Q columns (>= Q rows follow the piece): the lane that owns the last
        // block always has them searched.
        if (has_chunk && own_hi == Pk->n_blocks && own_lo < own_hi) {
          int64_t c_lo = (int64_t)Pk->text_len - (int64_t)Pk->k - (DL - Q - 1), c_hi = (int64_t)Pk->text_len + 1;
          if (c_lo < 1) c_lo = 1;
          if (c_hi > (int64_t)(Pk->n_blocks * 64)) c_hi = (int64_t)(Pk->n_blocks * 64);
          const int64_t fc = (int64_t)(Pk->dp_first_owned * 64) - col_base;
          run = fuse_add_range(queue, qcount, Pk->cand_count + kCtlFuseWord, Pk->fuse_queue_cap, 0xFFFFFFFFu, Pk->m + Pk->k, col_base,
                               fc > 0 ? (uint32_t)fc : 0u, run, (uint32_t)(c_lo - col_base), (uint32_t)(c_hi - col_base));
        }
      }
      if (run.x != kRunNone)
        run = fuse_add_range(queue, qcount, Pk->cand_count + kCtlFuseWord, Pk->fuse_queue_cap, 0xFFFFFFFFu, Pk->m + Pk->k, col_base, 0u, run,
                             kRunNone, 0u);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      nq = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile uint32_t*>(qcount));
      last = true;
    }
    // what the next segment starts from
    seg_it = it;
    first_segment = false;
    if (!last) {
      uint32_t* sv = reinterpret_cast<uint32_t*>(tile + kSegState) + lane;
      sv[0] = prev0; sv[64] = prev1;
      sv[128] = run.x; sv[192] = run.y; sv[256] = run.z; sv[320] = run.w & ~kRunPressure;
      if constexpr (CHECK) sv[384] = npure;
    }
    unsigned long long probe_t1 = 0;
    if (Pk->fused & 2u) probe_t1 = wall_clock64();  // SASSY_HIP_FUSED_PROBE: 100 MHz ticks spent streaming / in the chunk DP
    if (nq > Pk->fuse_queue_cap) {  // (cannot happen -- see press_at; the classic chain would take the search)
      if (lane == 0) atomicOr(&Pk->cand_count[kCtlFuseWord], kFuseOverflow);
      nq = 0;
    }
    if ((Pk->fused & 4u) == 0u && nq != 0) {  // (probe bit 4: no chunk DP at all -- timing only, no reports)
    // What the DP needs of the launch parameters is read HERE, through a pointer the optimiser cannot see
    // through: read from P they would be loaded at the top of the kernel and held (or spilled to VGPR lanes and
    // read back inside the streaming loop) for the whole life of the wave.
    kparams_ptr kp = (kparams_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    ScanParams L;
    L.text = kp->text;
    L.text_len = kp->text_len;
    L.n_blocks = kp->n_blocks;
    L.global_offset = kp->global_offset;
    L.wb = kp->wb;
    L.m = kp->m;
    L.k = kp->k;
    L.nwords = kp->nwords;
    L.flags = kp->flags;
    L.cand_cap = kp->cand_cap;
    L.row_tab = kp->row_tab;
    L.cand = kp->cand;
    L.cand_count = kp->cand_count;
    L.counters = kp->counters;
    L.dp_first_owned = kp->dp_first_owned;
    L.stash = kp->stash;
    L.stash_cap = kp->stash_cap;
    L.rev_n = 0;
    L.alpha = 0.0f;
    L.ov_steps = 0;
    L.ov_tab = nullptr;
    L.chunk_state = nullptr;
    L.texts_start = nullptr;
    L.texts_len = nullptr;
    if constexpr (CHECK) {  // (the Iupac masks of the pattern's letters: four slots -- A C T G -- or up to eight)
      L.nslots = DPNS == 4 ? 4u : kp->nslots;
#pragma unroll
      for (int sl = 0; sl < DPNS; ++sl) L.slot_val[sl] = kp->slot_val[sl];
    }
    // (the same for everything the DP derives from the thread index -- LDS addresses per lane, word, slot:
    // computed from an opaque copy, they cannot be hoisted in front of the streaming loop and held in VGPRs there)
    uint32_t dlane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t dwave = __builtin_amdgcn_readfirstlane(wave_s);
    asm volatile("" : "+v"(dlane), "+s"(dwave));
    unsigned char* dtile = smem + (size_t)dwave * kp->lds_per_wave;
    const uint2* dqueue = reinterpret_cast<const uint2*>(dtile + kTile);
    const uint32_t n_run = (last || nq < 64u) ? nq : (nq & ~63u);
    if (dlane == 0) atomicAdd(&L.cand_count[1], n_run);  // statistics: chunks
    // the DP's LDS -- slot masks, per-row carries -- takes the place of the text tile
    unsigned char* mask_bytes = dtile;
    uint32_t* carry = reinterpret_cast<uint32_t*>(dtile + DPNS * 512);
    // between segments only full batches of 64 chunks run; what is left over waits for the next pass
    for (uint32_t base = 0; base < n_run; base += 64u) {
      const bool has = base + dlane < n_run;
      uint2 e = make_uint2(0u, 0u);
      if (has) e = dqueue[base + dlane];
      ChunkDesc dsc;
      dsc.own_lo = e.x;
      dsc.own_hi = e.x + ((e.y >> 6) & 0xFFu);
      dsc.flags = kDescWindow;
      dsc.pad_ = (e.y & 63u) | ((e.y >> 14) << 8);  // byte shift | columns not to report << 8
      list_lanes<CHECK ? (int)PROFILE_IUPAC : (int)PROFILE_DNA, DPNS, true>(L, mask_bytes, carry, dlane, has, dsc, kNoStateSlot);
    }
    if ((kp->fused & 2u) && dlane == 0) {
      unsigned long long* pc = reinterpret_cast<unsigned long long*>(L.cand_count + 4);  // the control block's counters
      atomicAdd(&pc[0], probe_t1 - probe_t0);
      atomicAdd(&pc[1], wall_clock64() - probe_t1);
      atomicAdd(&pc[2], (unsigned long long)nq);
      atomicAdd(&pc[3], 1ull);
    }
    }
    // the queue keeps what did not fill a batch; the tile is the next segment's: the DP's LDS traffic is complete before
    // its first staging store (one wave, in-order LDS); the fence keeps the compiler from reordering across it
    {
      const uint32_t left = (last || (Pk->fused & 4u) || nq < 64u) ? 0u : (nq & 63u);
      uint2 keep_e = make_uint2(0u, 0u);
      if (lane < left) keep_e = queue[(nq & ~63u) + lane];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < left) queue[lane] = keep_e;
      if (lane == 0) *qcount = left;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (last) break;
  }
  }  // segments
}

// ====================================================================== K1-list, few long chunks
// The lane-per-chunk list kernel is bound by the instruction latency of ONE wave when the chunks
// are few and the pattern has several 32-row words (a q-gram counting prefilter leaves a few
// thousand chunks of ~10 blocks x 7 words for config 3: 0.4 ms with most SIMDs idle).  Here a chunk
// is worked on by a group of G = 2^j >= nwords lanes, lane w of the group owning pattern word w, as
// a software pipeline over the blocks: in step s lane w processes block s - w with the horizontal
// deltas lane w-1 produced for that block one step earlier (ds_bpermute within the group) and the
// vertical deltas it kept in registers from block s - w - 1.  A chunk of n blocks takes
// n + nwords - 1 word steps instead of n * nwords, and a wave holds 64 / G chunks, so the same
// work spreads over G times as many SIMDs.  Same row code, report rule and seam bookkeeping
// (scan_block, by the lane of the last word) as list_kernel; which of the two kernels takes a
// launch is decided on the device from the number of chunks (P.list_words_max).
template <int PROFILE, int NS>
__global__ __launch_bounds__(256) void list_words_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  unsigned char* mask_bytes = smem + (size_t)wave * (NS * 512u);  // [NS][64] u64, private per lane

  uint32_t n_desc = *P.desc_count;
  if (n_desc > P.desc_cap) n_desc = P.desc_cap;
  if (n_desc > P.list_words_max) return;  // many chunks: list_kernel (one lane each) runs them
  const uint32_t glog = P.list_group_log;
  const uint32_t G = 1u << glog;
  const uint32_t per_wave = 64u >> glog;
  const uint32_t wave_first = (blockIdx.x * kWavesPerGroup + wave) * per_wave;
  if (wave_first >= n_desc) return;  // wave-uniform
  const uint32_t w = lane & (G - 1u);
  const uint32_t di = wave_first + (lane >> glog);
  const uint32_t nwords = P.nwords;
  const bool has_chunk = di < n_desc && w < nwords;
  ChunkDesc d;
  d.own_lo = d.own_hi = d.flags = d.pad_ = 0;
  if (di < n_desc) d = P.desc[di];
  const uint64_t own_lo = d.own_lo, own_hi = d.own_hi;
  const bool clear_before = (d.flags & kDescClearBefore) != 0;
  uint64_t blk0 = own_lo;
  if (!clear_before) blk0 = own_lo > P.wb ? own_lo - P.wb : 0;
  const bool at_text_start = blk0 == 0 && (P.flags & kScanTextStart);
  const bool exact_start = clear_before || at_text_start;
  const int64_t x0 = exact_start ? -1 : (int64_t)(blk0 * 64 + P.m + P.k);
  const uint32_t my_iters = has_chunk ? (uint32_t)(own_hi - blk0) : 0u;

  const int k = (int)P.k;
  const uint32_t m = P.m;
  const bool last_word = w + 1 == nwords;
  const uint32_t last_rows = m - 32 * (nwords - 1);
  const uint32_t rows = last_word ? last_rows : 32u;
  uint32_t pkw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pkw[i] = has_chunk ? P.row_tab[8 * w + i] : 0u;
  // fresh start: every vertical delta on the left edge is +1
  uint32_t ohp = (last_word && last_rows != 32) ? ~(0xFFFFFFFFu >> last_rows) : 0xFFFFFFFFu;
  uint32_t ohm = 0;
  uint32_t st = kStDec;
  EmitCtx ctx;
  ctx.cand = P.cand;
  ctx.cand_count = P.cand_count;
  ctx.text_len = P.text_len;
  ctx.global_offset = P.global_offset;
  ctx.cand_cap = P.cand_cap;
  ctx.k = P.k;
  ctx.flags = P.flags;
  ctx.alpha = 0.0f;
  ctx.ov_steps = 0u;
  ctx.text_begin = 0;
  ctx.tag = 0;
  const unsigned char* my_masks = mask_bytes + lane * 8;  // (coop: set every step)
  DpWord Vout;
  Vout.vpl = Vout.vph = Vout.vml = Vout.vmh = 0;
  int ds_out = 0;

  // Eight lanes per chunk (5 .. 8 pattern words), Dna / Iupac, forward text inside the buffer: the lanes of a group
  // BUILD THE MASKS OF A BLOCK TOGETHER, once -- lane q its bytes 8q .. 8q+7, one byte of every slot's mask each, into
  // a ring of eight blocks per group (entry = group * 8 + step % 8 in place of the lane's own entry) -- instead of
  // every word's lane building the whole block again when the block reaches it (the masks were a third of a step).
  constexpr bool kCoopProfile = PROFILE == (int)PROFILE_DNA || PROFILE == (int)PROFILE_IUPAC;
  const bool chunk_here = di < n_desc;
  const uint32_t chunk_iters = chunk_here ? (uint32_t)(own_hi - blk0) : 0u;  // (also for the group's lane without a word)
  const bool coop = kCoopProfile && glog == 3u && P.rev_n == 0 && __all(!chunk_here || own_hi * 64 <= P.text_len);
  const uint32_t grp8 = (lane >> 3) * 8u;
  auto fetch8 = [&](uint32_t step) -> uint2 {
    uint2 v = make_uint2(0x58585858u, 0x58585858u);
    if (step < chunk_iters) v = *reinterpret_cast<const uint2*>(P.text + (blk0 + step) * 64 + 8u * w);
    return v;
  };
  auto plane8 = [&](const uint2 v, uint32_t sel, int bit) -> uint32_t {  // bit `bit` of the eight bytes
    const uint32_t a = __builtin_amdgcn_udot4(v.x & sel, 0x08040201u, 0u, false);
    return (__builtin_amdgcn_udot4(v.y & sel, 0x80402010u, a, false) >> bit) & 0xFFu;
  };

  // the text of the lane's next block is fetched one step ahead (a lone wave per SIMD cannot hide
  // the load latency behind other waves)
  auto fetch = [&](uint32_t step, uint32_t (&dst)[16]) {
    const bool on = has_chunk && step >= w && step - w < my_iters;
    const uint64_t blk = blk0 + (uint64_t)(step - w);
    fetch_block(P, on, blk, dst);
  };
  uint32_t xn[16];
  uint2 x8n = make_uint2(0u, 0u);
  if (coop) x8n = fetch8(0);
  else fetch(0, xn);
  for (uint32_t s = 0; __any(s < my_iters + nwords - 1u && my_iters != 0); ++s) {
    const bool active = has_chunk && s >= w && s - w < my_iters;
    const uint64_t b = blk0 + (uint64_t)(s - w);
    if (coop) {
      if constexpr (kCoopProfile) {
        const uint2 v = x8n;
        x8n = fetch8(s + 1);
        uint32_t mk[NS];
        if constexpr (PROFILE == (int)PROFILE_DNA) {
          const uint32_t p1 = plane8(v, 0x02020202u, 1), p2 = plane8(v, 0x04040404u, 2);
          mk[0] = ~(p1 | p2) & 0xFFu; mk[1] = p1 & ~p2; mk[2] = ~p1 & p2; mk[3] = p1 & p2;
        } else {
          const uint32_t b0 = plane8(v, 0x01010101u, 0), b1 = plane8(v, 0x02020202u, 1), b2 = plane8(v, 0x04040404u, 2),
                         b3 = plane8(v, 0x08080808u, 3), b4 = plane8(v, 0x10101010u, 4);
          const uint32_t base[4] = {iupac_base_plane<0>(b0, b1, b2, b3, b4), iupac_base_plane<1>(b0, b1, b2, b3, b4),
                                    iupac_base_plane<2>(b0, b1, b2, b3, b4), iupac_base_plane<3>(b0, b1, b2, b3, b4)};
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            const uint32_t sv = P.slot_val[q];  // wave-uniform; unused slots hold 0 -> empty mask
            uint32_t r = 0;
#pragma unroll
            for (int o = 0; o < 4; ++o) r |= base[o] & (((sv >> o) & 1u) ? 0xFFu : 0u);
            mk[q] = r;
          }
        }
        if (s < chunk_iters) {
          unsigned char* dst = mask_bytes + (grp8 + (s & 7u)) * 8u + w;
#pragma unroll
          for (int q = 0; q < NS; ++q) dst[q * 512] = (unsigned char)mk[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        my_masks = mask_bytes + (grp8 + ((s - w) & 7u)) * 8u;  // the block this lane's word works on now
      }
    } else {
    uint32_t x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = xn[c];
    fetch(s + 1, xn);
    {
      uint2 msk[NS];
      build_masks<PROFILE, NS>(x, P, msk);
#pragma unroll
      for (int q = 0; q < NS; ++q) *reinterpret_cast<uint2*>(mask_bytes + q * 512 + lane * 8) = msk[q];
    }
    }
    // the row above this word: what the lane of word w-1 left for this block one step ago
    DpWord V;
    V.vpl = __shfl_up(Vout.vpl, 1, 64);
    V.vph = __shfl_up(Vout.vph, 1, 64);
    V.vml = __shfl_up(Vout.vml, 1, 64);
    V.vmh = __shfl_up(Vout.vmh, 1, 64);
    int ds = __shfl_up(ds_out, 1, 64);
    if (w == 0) { V.vpl = V.vph = V.vml = V.vmh = 0; ds = 0; }
    ds += __popc(ohp) - __popc(ohm);
    uint32_t nhp, nhm;
    dp_word<false, false, PROFILE == (int)PROFILE_ASCII_BYTES>(V, my_masks, ohp, ohm, pkw, rows, nhp, nhm);
    rep4 rr = make_rep4(0u, 0u, 0u, 0u);
    const uint64_t vp = ((uint64_t)V.vph << 32) | V.vpl, vm = ((uint64_t)V.vmh << 32) | V.vml;
    const bool live = active && last_word && row_maybe_live(ds, V, k);
    if (__any(live)) rr = scan_blocks(ctx, live, vp, vm, ds, b, b >= own_lo, b + 1 == own_lo, x0, st);
    if (active) {
      ohp = nhp;
      ohm = nhm;
      Vout = V;
      ds_out = ds;
      if (last_word) st = live ? (rr.z & (kStDec | kStAmb)) : kStDec;
    }
    if (__any((rr.x | rr.y | (rr.z & kRepBase)) != 0u)) (void)emit_reports(ctx, rr, vp, vm, ds, b);
  }
  if (has_chunk && last_word) {
    const uint32_t fin = (st & kStAmb) ? kStatePass : ((st & kStDec) ? kStateDecTrue : kStateDecFalse);
    P.chunk_state[di] = (uint8_t)fin;
    if (own_hi == P.n_blocks) {
      uint32_t* tail = P.cand_count + kCtlTailWord;
      tail[0] = (uint32_t)own_lo; tail[1] = fin; tail[2] = d.flags; tail[3] = 1u;
    }
  }
}

// ====================================================================== K1-list, one lane per BLOCK
// list_words_kernel pipelines a chunk over its pattern words: (blocks + words - 1) steps of 32 dependent rows each -- for
// config 3 (m = 200, k = 20: ten blocks x seven words) a chain of ~500 rows x 30 instructions on one wave per SIMD, 89 us
// for a few thousand chunks that are 5 us of arithmetic.  The dependencies of the text-tiled recurrence allow a finer
// wavefront: cell (block b, row r) needs (b, r - 1) -- the lane's own registers -- and the two carry bits of (b - 1, r).
// Here lane i of a group of G lanes owns BLOCK i of a chunk and computes row t - i in step t: the carry bits of its row
// arrive from lane i - 1 by DPP (computed one step earlier), the horizontal deltas V stay in the lane, the Eq word of
// (its block, its row) comes from the block's slot masks, which the lane built once.  A chunk of n blocks takes
// m + n - 1 row steps (209 instead of ~500), every lane is busy m of them, and the 64 / G chunks of a wave fill it.
// Chunks longer than G blocks run in passes of G blocks; the carries between two passes -- a byte per row -- wait in
// LDS (read at step r by the pass's first lane, written at step r + G - 1 by its last: in place).
// The lane that finishes a block (row m - 1) decides about its reports exactly as list_kernel does (scan_block with the
// plateau state handed on from the block to its left, one step earlier), the wave appends them with one atomic.
// Row slots per row: bytes in LDS (row_tab is a byte array indexed by the row), padded on both sides so that the
// prefetches of lanes that have not started / are done read slot 0.
constexpr uint32_t kRowsPad = 64;
template <int PROFILE, int NS>
__global__ __launch_bounds__(256) void list_rows_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t n_desc = *P.desc_count;
  if (n_desc > P.desc_cap) n_desc = P.desc_cap;
  if (n_desc > P.list_words_max) return;  // many chunks: list_kernel (one lane each) runs them
  // groups of G lanes (any G from 4 to 64: the carries travel by wave_shr:1, which knows no rows), 64 / G chunks per wave
  const uint32_t G = P.list_group;
  const uint32_t per_wave = 64u / G;
  if (blockIdx.x * kWavesPerGroup * per_wave >= n_desc) return;  // (the whole workgroup)
  const uint32_t m = P.m;
  const uint32_t slots_bytes = (kRowsPad + m + kRowsPad + 15u) & ~15u;
  unsigned char* rowslot = smem + kRowsPad;  // rowslot[r] = 2 * slot of row r, r in [-kRowsPad, m + kRowsPad)
  {
    const uint8_t* rt = reinterpret_cast<const uint8_t*>(P.row_tab);
    for (uint32_t x = threadIdx.x; x < slots_bytes; x += blockDim.x) {
      const int r = (int)x - (int)kRowsPad;
      smem[x] = (r >= 0 && r < (int)m) ? rt[r] : (unsigned char)0;
    }
  }
  __syncthreads();
  const uint32_t carry_bytes = (m + 15u) & ~15u;
  unsigned char* wbase = smem + slots_bytes + (size_t)wave * (NS * 512u + per_wave * carry_bytes);
  unsigned char* mask_bytes = wbase;  // [NS][64] u64
  const uint32_t gi = lane / G, li = lane - gi * G;
  const bool in_group = gi < per_wave;  // (64 mod G lanes at the wave's end have no group)
  unsigned char* carry = wbase + NS * 512u + (in_group ? gi : 0u) * carry_bytes;  // the group's carries between two passes
  const uint32_t wave_first = (blockIdx.x * kWavesPerGroup + wave) * per_wave;
  if (wave_first >= n_desc) return;  // wave-uniform (no workgroup barrier below)
  const uint32_t di = wave_first + gi;
  const bool chunk_here = in_group && di < n_desc;
  ChunkDesc d;
  d.own_lo = d.own_hi = d.flags = d.pad_ = 0;
  if (chunk_here) d = P.desc[di];
  const uint64_t own_lo = d.own_lo, own_hi = d.own_hi;
  const bool clear_before = (d.flags & kDescClearBefore) != 0;
  uint64_t blk0 = own_lo;
  if (!clear_before) blk0 = own_lo > P.wb ? own_lo - P.wb : 0;
  const bool at_text_start = blk0 == 0 && (P.flags & kScanTextStart);
  const bool exact_start = clear_before || at_text_start;
  const int64_t x0 = exact_start ? -1 : (int64_t)(blk0 * 64 + P.m + P.k);
  const uint32_t nb = chunk_here ? (uint32_t)(own_hi - blk0) : 0u;
  const int k = (int)P.k;
  EmitCtx ctx;
  ctx.cand = P.cand;
  ctx.cand_count = P.cand_count;
  ctx.text_len = P.text_len;
  ctx.global_offset = P.global_offset;
  ctx.cand_cap = P.cand_cap;
  ctx.k = P.k;
  ctx.flags = P.flags;
  ctx.alpha = 0.0f;
  ctx.ov_steps = 0u;
  ctx.text_begin = 0;
  ctx.tag = 0;
  const unsigned char* my_masks = mask_bytes + lane * 8;
  const uint32_t first_lane = li == 0u ? 1u : 0u;
  const uint32_t keep_dpp = li == 0u ? 0u : 0xFFFFFFFFu;  // a group's first lane takes nothing from the lane on its left
  uint32_t st_pass = kStDec;   // plateau state in front of the pass's first block (a chunk starts with dec = true)
  int ds_pass = (int)m;        // cost on the pass's left edge in the last row (a fresh start: D[m][start] = m)
  uint32_t st_last = kStDec;   // ... behind the chunk's last block (the lane that owns it)
  for (uint32_t pass = 0; __any(pass * G < nb); ++pass) {
    const uint32_t bi = pass * G + li;  // the lane's block inside its chunk
    const bool has_blk = bi < nb;
    const uint64_t b = blk0 + bi;
    {
      uint32_t x[16];
      fetch_block(P, has_blk, b, x);
      uint2 msk[NS];
      build_masks<PROFILE, NS>(x, P, msk);
#pragma unroll
      for (int q = 0; q < NS; ++q) *reinterpret_cast<uint2*>(mask_bytes + q * 512 + lane * 8) = msk[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // blocks of this pass in the wave's longest group (wave-uniform): m + that - 1 steps
    uint32_t span = has_blk ? li + 1u : 0u;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) span = max(span, (uint32_t)__shfl_xor((int)span, dd, 64));
    span = (uint32_t)__builtin_amdgcn_readfirstlane((int)span);
    const uint32_t nsteps = m + span - 1u;
    const bool more = (pass + 1u) * G < nb;  // the chunk goes on behind this pass: its last lane leaves the carries
    const bool any_more = __any(more);
    const bool later_pass = pass != 0u;      // wave-uniform
    DpWord V;
    V.vpl = V.vph = V.vml = V.vmh = 0;
    uint32_t cout = 0;           // carry bits of the row the lane computed last: bit 0 = +1, bit 1 = -1
    uint32_t stv = st_pass;      // plateau state behind the lane's block (valid once the block is done)
    int dsr = ds_pass;           // last-row cost on the lane's RIGHT edge (valid once the block is done)
    int ds_blk = 0;              // ... on its left edge
    bool live = false;           // the block's last row may hold a cell <= k
    uint2 zz = make_uint2(0u, 0u);
    asm volatile("" : "+v"(zz.x), "+v"(zz.y));
    // Eq word of (my block, row r), two steps ahead: the row's slot byte, then the mask
    const unsigned char* rp = rowslot - (int)li;  // rp[0] = slot byte of row t - li
    uint32_t slot_n = rp[1];                       // row 1 - li (for step 1)
    uint2 eq_n = *reinterpret_cast<const uint2*>(my_masks + ((uint32_t)rp[0] << 8));  // row -li (step 0)
    // one row step.  CHECK: the lane may be outside its rows (ramp-up / tail); TAIL: blocks are being finished
    // PLAIN: a chunk's only pass, no carries through LDS (the steady rows of nearly every chunk); eq_use / eq_load: the two
    // Eq registers change roles step by step (two steps per iteration of the steady loop: no copies)
    auto step = [&](uint32_t t, auto check_tag, auto tail_tag, auto plain_tag, uint2& eq_use, uint2& eq_load) {
      constexpr bool CHECK = decltype(check_tag)::value, TAIL = decltype(tail_tag)::value, PLAIN = decltype(plain_tag)::value;
      // the carry bits the lane on the left produced one step ago (a group's first lane: a fresh start -- every
      // left-edge delta is +1 -- or, in a later pass, what the previous pass's last lane left for this row)
      uint32_t cin = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cout, 0x138, 0xF, 0xF, false);  // wave_shr:1
      cin = (cin & keep_dpp) | ((!PLAIN && later_pass) ? 0u : first_lane);
      const int r = (int)t - (int)li;
      bool active = has_blk;
      if constexpr (CHECK) active = has_blk && (uint32_t)r < m;
      if constexpr (!PLAIN) {
        if (later_pass) {  // wave-uniform
          if (li == 0u && active) cin = carry[r];
        }
      }
      const uint2 eq = eq_use;
      eq_load = *reinterpret_cast<const uint2*>(my_masks + (slot_n << 8));
      slot_n = rp[2];
      ++rp;
      if (active) {
        uint32_t nhp = 0, nhm = 0;
        dp_row<false>(V, eq, cin & 1u, cin >> 1, nhp, nhm, zz);
        cout = nhp | (nhm << 1);
        if constexpr (!PLAIN) {
          if (any_more) {  // wave-uniform
            if (more && li == G - 1u) carry[r] = (unsigned char)cout;
          }
        }
      }
      if constexpr (TAIL) {
        // a block is done: its last row's cost on the left edge comes from the block on its left (done one step ago)
        const int ds_left = __builtin_amdgcn_update_dpp(0, dsr, 0x138, 0xF, 0xF, false);
        if (has_blk && r == (int)m - 1) {
          ds_blk = li == 0u ? ds_pass : ds_left;
          dsr = ds_blk + (int)__popc(V.vpl) + (int)__popc(V.vph) - (int)__popc(V.vml) - (int)__popc(V.vmh);
          live = row_maybe_live(ds_blk, V, k);
        }
      }
    };
    uint32_t t = 0;
    // ramp-up until every lane with a block is inside its rows (t = span - 1) or the first block is done (t = m - 1)
    const uint32_t t_steady = min(span > 0u ? span - 1u : 0u, m - 1u), t_tail = m - 1u;
    uint2 eq_o = make_uint2(0u, 0u);  // (the other Eq register: even steps use eq_n and load eq_o, odd steps the other way)
    auto step1 = [&](uint32_t tt, auto check_tag, auto tail_tag, auto plain_tag) {  // (tt is wave-uniform)
      if (tt & 1u) step(tt, check_tag, tail_tag, plain_tag, eq_o, eq_n);
      else step(tt, check_tag, tail_tag, plain_tag, eq_n, eq_o);
    };
    for (; t < t_steady; ++t) step1(t, std::true_type{}, std::false_type{}, std::false_type{});   // ramp-up
    if (has_blk) {                                                             // every lane with a block is inside its rows
      uint32_t t2 = t;
      if ((t2 & 1u) && t2 < t_tail) { step1(t2, std::false_type{}, std::false_type{}, std::false_type{}); ++t2; }  // (t2 even from here)
      if (!later_pass && !any_more) {
        for (; t2 + 1u < t_tail; t2 += 2u) {
          step(t2, std::false_type{}, std::false_type{}, std::true_type{}, eq_n, eq_o);
          step(t2 + 1u, std::false_type{}, std::false_type{}, std::true_type{}, eq_o, eq_n);
        }
      } else {
        for (; t2 + 1u < t_tail; t2 += 2u) {
          step(t2, std::false_type{}, std::false_type{}, std::false_type{}, eq_n, eq_o);
          step(t2 + 1u, std::false_type{}, std::false_type{}, std::false_type{}, eq_o, eq_n);
        }
      }
      for (; t2 < t_tail; ++t2) step1(t2, std::false_type{}, std::false_type{}, std::false_type{});
    } else {
      rp += t_tail - t;
    }
    for (t = t_tail; t < nsteps; ++t) step1(t, std::true_type{}, std::true_type{}, std::false_type{});  // blocks are finished, one per step
    // ---- the pass's reports.  Every lane still holds its block's last row (V, ds_blk).  The report rule of a block
    // needs the plateau state behind the block on its left; that block is usually not live (state: dec = true, settled), so
    // all live lanes decide at once on that assumption, and a lane whose left neighbour turned out otherwise decides again
    // (a plateau that runs across a block border: a second round, rarely more) -- two calls of scan_block per pass
    // instead of one per live block and step, and ONE atomic for the pass's reports.
    rep4 rr = make_rep4(0u, 0u, 0u, 0u);
    const uint64_t vp = ((uint64_t)V.vph << 32) | V.vpl, vm = ((uint64_t)V.vmh << 32) | V.vml;
    uint32_t st_used = 0xFFFFFFFFu;  // the state the lane's decision stands on (none yet)
    stv = kStDec;
    for (;;) {
      const uint32_t st_dpp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)stv, 0x138, 0xF, 0xF, false);
      const uint32_t st_in = li == 0u ? st_pass : st_dpp;
      const bool again = live && st_in != st_used;
      if (!__any(again)) break;
      const rep4 r2 = scan_blocks(ctx, again, vp, vm, ds_blk, b, b >= own_lo, b + 1 == own_lo, x0, st_in);
      if (again) {
        rr = r2;
        stv = rr.z & (kStDec | kStAmb);
        st_used = st_in;
      }
    }
    if (has_blk && bi + 1u == nb) st_last = stv;
    if (__any((rr.x | rr.y | (rr.z & kRepBase)) != 0u)) (void)emit_reports(ctx, rr, vp, vm, ds_blk, b);
    // the next pass's first lane continues behind this pass's last one
    const int last_lane = (int)((in_group ? gi : 0u) * G + G - 1u);
    st_pass = (uint32_t)__shfl((int)stv, last_lane, 64);
    ds_pass = __shfl(dsr, last_lane, 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (chunk_here && nb != 0u && (nb - 1u) % G == li) {  // the lane that owned the chunk's last block
    const uint32_t fin = (st_last & kStAmb) ? kStatePass : ((st_last & kStDec) ? kStateDecTrue : kStateDecFalse);
    P.chunk_state[di] = (uint8_t)fin;
    if (own_hi == P.n_blocks) {
      uint32_t* tail = P.cand_count + kCtlTailWord;
      tail[0] = (uint32_t)own_lo; tail[1] = fin; tail[2] = d.flags; tail[3] = 1u;
    }
  }
}

// ------------------------------------------------------------------ launcher
template <int PROFILE, int NS, int SB>
static hipError_t launch_sb(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  static DeviceOnce attr_set;  // LDS beyond the 64 KiB default needs an explicit opt-in
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_kernel<PROFILE, NS, SB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.done();
  }
  hipLaunchKernelGGL((scan_kernel<PROFILE, NS, SB>), dim3(grid), dim3(64u * (P.waves_per_group ? P.waves_per_group : 4u)), smem, stream, P);
  return hipGetLastError();
}
template <int PROFILE, int NS>
static hipError_t launch_one(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  if (P.carry_global) {  // (long patterns: half-line staging, the carries in global memory)
    hipLaunchKernelGGL((scan_kernel<PROFILE, NS, 1, true>), dim3(grid), dim3(64u * (P.waves_per_group ? P.waves_per_group : 4u)), smem, stream, P);
    return hipGetLastError();
  }
  return P.stage_blocks == 1 ? launch_sb<PROFILE, NS, 1>(P, grid, smem, stream)
                             : launch_sb<PROFILE, NS, 2>(P, grid, smem, stream);
}

template <int PROFILE, int NS, int SB, int NPG>
static hipError_t launch_filter_npg(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_kernel<PROFILE, NS, SB, NPG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.done();
  }
  hipLaunchKernelGGL((filter_kernel<PROFILE, NS, SB, NPG>), dim3(grid), dim3(256), smem, stream, P);
  return hipGetLastError();
}
template <int PROFILE, int NS, int SB>
static hipError_t launch_filter_sb(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  if (P.piece_groups == 1) return launch_filter_npg<PROFILE, NS, SB, 1>(P, grid, smem, stream);
  if (P.piece_groups == 2) return launch_filter_npg<PROFILE, NS, SB, 2>(P, grid, smem, stream);
  return launch_filter_npg<PROFILE, NS, SB, 0>(P, grid, smem, stream);
}
template <int PROFILE, int NS>
static hipError_t launch_filter_one(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  return P.stage_blocks == 1 ? launch_filter_sb<PROFILE, NS, 1>(P, grid, smem, stream)
                             : launch_filter_sb<PROFILE, NS, 2>(P, grid, smem, stream);
}
template <int PROFILE, int NS>
static hipError_t launch_list_one(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&list_kernel<PROFILE, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&list_words_kernel<PROFILE, NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&list_rows_kernel<PROFILE, NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set.done();
  }
  if (P.list_words_max) {  // few chunks of a multi-word pattern: a lane per block (list_rows) or per pattern word (list_words)
    const uint32_t per_group = P.list_rows ? 4u * (64u / P.list_group) : 256u >> P.list_group_log;
    const uint32_t wgrid = (std::min(P.list_words_max, P.desc_cap) + per_group - 1) / per_group;
    if (P.list_rows) {
      const uint32_t per_wave = 64u / P.list_group;
      const size_t lds = ((size_t)(2 * kRowsPad + P.m + 15u) & ~(size_t)15) +
                         (size_t)kWavesPerGroup * (NS * 512u + per_wave * (size_t)((P.m + 15u) & ~15u));
      hipLaunchKernelGGL((list_rows_kernel<PROFILE, NS>), dim3(wgrid), dim3(256), lds, stream, P);
    } else {
      hipLaunchKernelGGL((list_words_kernel<PROFILE, NS>), dim3(wgrid), dim3(256), (size_t)kWavesPerGroup * NS * 512u,
                         stream, P);
    }
  }
  // (grid = 0: the host leaves the lane-per-chunk kernel out -- the launch above takes every chunk list it expects, and a
  // list beyond list_words_max sends the search through here once more with list_words_max = 0)
  if (grid && P.carry_global)
    hipLaunchKernelGGL((list_kernel<PROFILE, NS, true>), dim3(grid), dim3(64u * (P.waves_per_group ? P.waves_per_group : 4u)), smem, stream, P);
  else if (grid)
    hipLaunchKernelGGL((list_kernel<PROFILE, NS>), dim3(grid), dim3(64u * (P.waves_per_group ? P.waves_per_group : 4u)), smem, stream, P);
  return hipGetLastError();
}

#ifndef SASSY_SCAN_PROFILE
#error "compile with -DSASSY_SCAN_PROFILE=<0|1|2> (one translation unit per profile)"
#endif

#if SASSY_SCAN_PROFILE == 1
// Dna always has exactly the four slots A, C, T, G.
hipError_t launch_scan_dna(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  return launch_one<PROFILE_DNA, 4>(P, grid, smem, stream);
}
template <int Q, int NPG>
static hipError_t launch_filter_planes_q(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = (size_t)kWavesPerGroup * P.lds_per_wave;
  const LaunchEvents ev = g_launch_events;
  g_launch_events = LaunchEvents{};
  // (timed launches: the dispatch itself carries the events -- its own begin and end, what rocprofv3 reports --
  // instead of two markers around it, which add a few microseconds of queue latency)
  if (P.fused && ev.start)
    hipExtLaunchKernelGGL((filter_dna_kernel<Q, NPG, true>), dim3(grid), dim3(256), smem, stream, ev.start, ev.stop, 0, P);
  else if (P.fused) hipLaunchKernelGGL((filter_dna_kernel<Q, NPG, true>), dim3(grid), dim3(256), smem, stream, P);
  else hipLaunchKernelGGL((filter_dna_kernel<Q, NPG, false>), dim3(grid), dim3(256), smem, stream, P);
  return hipGetLastError();
}
// the paired filter (fused launch only): S super-pieces of two Q-row sub-pieces each
template <int Q, int S>
static hipError_t launch_filter_pair_q(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = (size_t)kWavesPerGroup * P.lds_per_wave;
  const LaunchEvents ev = g_launch_events;
  g_launch_events = LaunchEvents{};
  if (!P.fused) return hipErrorInvalidValue;
  if (ev.start)
    hipExtLaunchKernelGGL((filter_dna_kernel<Q, S, true, false, true>), dim3(grid), dim3(256), smem, stream, ev.start, ev.stop, 0, P);
  else hipLaunchKernelGGL((filter_dna_kernel<Q, S, true, false, true>), dim3(grid), dim3(256), smem, stream, P);
  return hipGetLastError();
}
template <int Q>
static hipError_t launch_filter_pair_s(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  switch (P.pair) {
    case 1: return launch_filter_pair_q<Q, 1>(P, grid, stream);
    case 2: return launch_filter_pair_q<Q, 2>(P, grid, stream);
    case 3: return launch_filter_pair_q<Q, 3>(P, grid, stream);
    case 4: return launch_filter_pair_q<Q, 4>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}
static hipError_t launch_filter_pair(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  switch (P.piece_len) {
    case 5: return launch_filter_pair_s<5>(P, grid, stream);
    case 6: return launch_filter_pair_s<6>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}
template <int NPG>
static hipError_t launch_filter_planes(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  switch (P.piece_len) {  // 7 .. 12 by default; shorter pieces with SASSY_HIP_PREFILTER=1 / sassy_hip_set_prefilter(s, 1)
    case 2: return launch_filter_planes_q<2, NPG>(P, grid, stream);
    case 3: return launch_filter_planes_q<3, NPG>(P, grid, stream);
    case 4: return launch_filter_planes_q<4, NPG>(P, grid, stream);
    case 5: return launch_filter_planes_q<5, NPG>(P, grid, stream);
    case 6: return launch_filter_planes_q<6, NPG>(P, grid, stream);
    case 7: return launch_filter_planes_q<7, NPG>(P, grid, stream);
    case 8: return launch_filter_planes_q<8, NPG>(P, grid, stream);
    case 9: return launch_filter_planes_q<9, NPG>(P, grid, stream);
    case 10: return launch_filter_planes_q<10, NPG>(P, grid, stream);
    case 11: return launch_filter_planes_q<11, NPG>(P, grid, stream);
    case 12: return launch_filter_planes_q<12, NPG>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}
template <int Q>
static hipError_t launch_filter_table_q(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = (1u << (2 * Q - 3)) + 4 * 4096u;
  hipLaunchKernelGGL((filter_table_kernel<Q>), dim3(grid), dim3(256), smem, stream, P);
  return hipGetLastError();
}
template <int Q>
static hipError_t launch_filter_dna_multi_q(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL((filter_dna_multi_kernel<2, Q>), dim3(grid), dim3(256), (size_t)kWavesPerGroup * P.lds_per_wave, stream, P);
  return hipGetLastError();
}
// piece lengths 6 .. 12 (stage_blocks = 2)
hipError_t launch_filter_dna_multi(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  switch (P.piece_len) {
    case 6: return launch_filter_dna_multi_q<6>(P, grid, stream);
    case 7: return launch_filter_dna_multi_q<7>(P, grid, stream);
    case 8: return launch_filter_dna_multi_q<8>(P, grid, stream);
    case 9: return launch_filter_dna_multi_q<9>(P, grid, stream);
    case 10: return launch_filter_dna_multi_q<10>(P, grid, stream);
    case 11: return launch_filter_dna_multi_q<11>(P, grid, stream);
    case 12: return launch_filter_dna_multi_q<12>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}
// the q-gram table filter is profile-independent code: it lives in the Dna translation unit
hipError_t launch_filter_table(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  switch (P.piece_len) {
    case 7: return launch_filter_table_q<7>(P, grid, stream);
    case 8: return launch_filter_table_q<8>(P, grid, stream);
    case 9: return launch_filter_table_q<9>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_filter_dna(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  if (P.piece_planes && P.lin_steps) {  // linear streaming variant (grid sized by the host for its wave ranges)
    if (P.piece_groups == 1)
      hipLaunchKernelGGL((filter_dna_linear_kernel<1>), dim3(grid), dim3(256), (size_t)kWavesPerGroup * 8192u, stream, P);
    else
      hipLaunchKernelGGL((filter_dna_linear_kernel<2>), dim3(grid), dim3(256), (size_t)kWavesPerGroup * 8192u, stream, P);
    return hipGetLastError();
  }
  if (P.piece_planes && P.pair) return launch_filter_pair(P, grid, stream);
  if (P.piece_planes)  // <= 8 pieces: the bit-plane kernel (lds_per_wave = the staging tile, + the chunk queue when fused)
    return P.piece_groups == 1 ? launch_filter_planes<1>(P, grid, stream) : launch_filter_planes<2>(P, grid, stream);
  return launch_filter_one<PROFILE_DNA, 4>(P, grid, smem, stream);
}
hipError_t launch_list_dna(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  return launch_list_one<PROFILE_DNA, 4>(P, grid, smem, stream);
}
#else
#if SASSY_SCAN_PROFILE == 2
hipError_t launch_scan_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR = PROFILE_IUPAC;
#else
hipError_t launch_scan_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR = PROFILE_ASCII;
#endif
#if SASSY_SCAN_PROFILE == 0
  if (P.profile == PROFILE_ASCII_BYTES) return launch_one<(int)PROFILE_ASCII_BYTES, 8>(P, grid, smem, stream);
#endif
  if (P.nslots <= 4) return launch_one<PR, 4>(P, grid, smem, stream);
  if (P.nslots <= 8) return launch_one<PR, 8>(P, grid, smem, stream);
  if (P.nslots <= 16) return launch_one<PR, 16>(P, grid, smem, stream);
#if SASSY_SCAN_PROFILE == 0
  // Ascii patterns with many distinct bytes (the reference's Ascii profile has 256 slots)
  if (P.nslots <= 32) return launch_one<PR, 32>(P, grid, smem, stream);
  if (P.nslots <= 64) return launch_one<PR, 64>(P, grid, smem, stream);
#endif
  return hipErrorInvalidValue;
}
#if SASSY_SCAN_PROFILE == 2
// plain patterns under the Iupac profile: the fused bit-plane launch with the text check (filter_dna_kernel, CHECK)
template <int Q>
static hipError_t launch_filter_planes_iupac_q(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = (size_t)kWavesPerGroup * P.lds_per_wave;
  if (P.piece_groups != 1) return hipErrorInvalidValue;  // (eight pieces + the check do not fit 128 VGPRs: the host asks for <= 4)
  const LaunchEvents ev = g_launch_events;
  g_launch_events = LaunchEvents{};
  if (ev.start)
    hipExtLaunchKernelGGL((filter_dna_kernel<Q, 1, true, true>), dim3(grid), dim3(256), smem, stream, ev.start, ev.stop, 0, P);
  else hipLaunchKernelGGL((filter_dna_kernel<Q, 1, true, true>), dim3(grid), dim3(256), smem, stream, P);
  return hipGetLastError();
}
template <int Q, int S, int DPNS>
static hipError_t launch_filter_pair_iupac_ns(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  const size_t smem = (size_t)kWavesPerGroup * P.lds_per_wave;
  const LaunchEvents ev = g_launch_events;
  g_launch_events = LaunchEvents{};
  if (ev.start)
    hipExtLaunchKernelGGL((filter_dna_kernel<Q, S, true, true, true, DPNS>), dim3(grid), dim3(256), smem, stream, ev.start, ev.stop, 0, P);
  else hipLaunchKernelGGL((filter_dna_kernel<Q, S, true, true, true, DPNS>), dim3(grid), dim3(256), smem, stream, P);
  return hipGetLastError();
}
template <int Q, int S>
static hipError_t launch_filter_pair_iupac_q(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  if (P.nslots > 8 || (P.nslots > 4 && P.nwords > 4)) return hipErrorInvalidValue;  // (the host asks for what fits the tile)
  return P.nslots <= 4 ? launch_filter_pair_iupac_ns<Q, S, 4>(P, grid, stream) : launch_filter_pair_iupac_ns<Q, S, 8>(P, grid, stream);
}
static hipError_t launch_filter_planes_iupac(const ScanParams& P, uint32_t grid, hipStream_t stream) {
  if (P.pair) {  // the paired filter with the text check (the host asks for at most three super-pieces here)
    const uint32_t key = P.piece_len * 8u + P.pair;
    switch (key) {
      case 5 * 8 + 1: return launch_filter_pair_iupac_q<5, 1>(P, grid, stream);
      case 5 * 8 + 2: return launch_filter_pair_iupac_q<5, 2>(P, grid, stream);
      case 5 * 8 + 3: return launch_filter_pair_iupac_q<5, 3>(P, grid, stream);
      case 6 * 8 + 1: return launch_filter_pair_iupac_q<6, 1>(P, grid, stream);
      case 6 * 8 + 2: return launch_filter_pair_iupac_q<6, 2>(P, grid, stream);
      case 6 * 8 + 3: return launch_filter_pair_iupac_q<6, 3>(P, grid, stream);
      default: return hipErrorInvalidValue;
    }
  }
  switch (P.piece_len) {
    case 5: return launch_filter_planes_iupac_q<5>(P, grid, stream);
    case 6: return launch_filter_planes_iupac_q<6>(P, grid, stream);
    case 7: return launch_filter_planes_iupac_q<7>(P, grid, stream);
    case 8: return launch_filter_planes_iupac_q<8>(P, grid, stream);
    case 9: return launch_filter_planes_iupac_q<9>(P, grid, stream);
    case 10: return launch_filter_planes_iupac_q<10>(P, grid, stream);
    case 11: return launch_filter_planes_iupac_q<11>(P, grid, stream);
    case 12: return launch_filter_planes_iupac_q<12>(P, grid, stream);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_filter_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR2 = PROFILE_IUPAC;
  if (P.piece_planes) return P.fused ? launch_filter_planes_iupac(P, grid, stream) : hipErrorInvalidValue;
#else
hipError_t launch_filter_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR2 = PROFILE_ASCII;
#endif
  if (P.nslots <= 4) return launch_filter_one<PR2, 4>(P, grid, smem, stream);
  if (P.nslots <= 8) return launch_filter_one<PR2, 8>(P, grid, smem, stream);
  if (P.nslots <= 16) return launch_filter_one<PR2, 16>(P, grid, smem, stream);
  return hipErrorInvalidValue;
}
#if SASSY_SCAN_PROFILE == 2
hipError_t launch_list_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR3 = PROFILE_IUPAC;
#else
hipError_t launch_list_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR3 = PROFILE_ASCII;
#endif
#if SASSY_SCAN_PROFILE == 0
  if (P.profile == PROFILE_ASCII_BYTES) return launch_list_one<(int)PROFILE_ASCII_BYTES, 8>(P, grid, smem, stream);
#endif
  if (P.nslots <= 4) return launch_list_one<PR3, 4>(P, grid, smem, stream);
  if (P.nslots <= 8) return launch_list_one<PR3, 8>(P, grid, smem, stream);
  if (P.nslots <= 16) return launch_list_one<PR3, 16>(P, grid, smem, stream);
#if SASSY_SCAN_PROFILE == 0
  if (P.nslots <= 32) return launch_list_one<PR3, 32>(P, grid, smem, stream);
  if (P.nslots <= 64) return launch_list_one<PR3, 64>(P, grid, smem, stream);
#endif
  return hipErrorInvalidValue;
}
#endif

}  // namespace sassy_hip
