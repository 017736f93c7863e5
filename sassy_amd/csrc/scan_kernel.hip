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
//     (one full cache line, 2 blocks) for each of the 64 lane chunks of a wave.
//   * the per-block equality masks ("profile") are built wave-cooperatively: all 64 lanes read
//     one byte of a block from the tile, one v_cmp per profile slot yields the 64-bit mask in an
//     SGPR pair (the ballot), v_writelane drops it into the owner lane's register; masks then
//     live in LDS so that a pattern row fetches its Eq word with one conflict-free ds_read_b64
//     at a wave-uniform slot offset.
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

#include "common.h"

namespace sassy_hip {

__device__ __forceinline__ uint32_t lo32(uint64_t x) { return (uint32_t)x; }
__device__ __forceinline__ uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }

// v_writelane_b32: drop a wave-uniform (SGPR) value into one lane of a VGPR.  clang has no builtin
// for it.  gfx9 VALU instructions may read only one SGPR (constant bus limit 1), so the lane
// select travels in M0 (which does not count) and the data in an SGPR.  M0 is written by scalar
// code, one wait state before its first VALU reader; four 64-bit masks go per statement.
__device__ __forceinline__ void writelane_x8(uint32_t (&v)[8], const uint32_t (&sv)[8], int L) {
  asm volatile(
      "s_mov_b32 m0, %16\n\t"
      "s_nop 0\n\t"
      "v_writelane_b32 %0, %8, m0\n\t"
      "v_writelane_b32 %1, %9, m0\n\t"
      "v_writelane_b32 %2, %10, m0\n\t"
      "v_writelane_b32 %3, %11, m0\n\t"
      "v_writelane_b32 %4, %12, m0\n\t"
      "v_writelane_b32 %5, %13, m0\n\t"
      "v_writelane_b32 %6, %14, m0\n\t"
      "v_writelane_b32 %7, %15, m0"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
      : "s"(sv[0]), "s"(sv[1]), "s"(sv[2]), "s"(sv[3]), "s"(sv[4]), "s"(sv[5]), "s"(sv[6]), "s"(sv[7]),
        "s"(L));  // M0 is reserved (not allocated) on gfx9, so it needs no clobber entry
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
__device__ __forceinline__ void dp_row(DpWord& V, uint2 eq, uint32_t hp0, uint32_t hm0,
                                       uint32_t& nhp, uint32_t& nhm) {
  const uint32_t vxl = eq.x | V.vml, vxh = eq.y | V.vmh;
  const uint32_t e2l = eq.x | hm0, e2h = eq.y;
  const uint64_t t = ((uint64_t)(e2h & V.vph) << 32) | (e2l & V.vpl);
  const uint64_t vp64 = ((uint64_t)V.vph << 32) | V.vpl;
  const uint64_t sum = t + vp64;
  const uint32_t hxl = ((uint32_t)sum ^ V.vpl) | e2l;
  const uint32_t hxh = ((uint32_t)(sum >> 32) ^ V.vph) | e2h;
  const uint32_t Hpl = V.vml | ~(hxl | V.vpl), Hph = V.vmh | ~(hxh | V.vph);
  const uint32_t Hml = V.vpl & hxl, Hmh = V.vph & hxh;
  nhp = __builtin_amdgcn_alignbit(nhp, Hph, 31);  // (nhp << 1) | (Hph >> 31)
  nhm = __builtin_amdgcn_alignbit(nhm, Hmh, 31);
  const uint32_t Hp2h = __builtin_amdgcn_alignbit(Hph, Hpl, 31), Hp2l = (Hpl << 1) | hp0;
  const uint32_t Hm2h = __builtin_amdgcn_alignbit(Hmh, Hml, 31), Hm2l = (Hml << 1) | hm0;
  V.vpl = Hm2l | ~(vxl | Hp2l);
  V.vph = Hm2h | ~(vxh | Hp2h);
  V.vml = Hp2l & vxl;
  V.vmh = Hp2h & vxh;
}

// Lower bound on the minimum of the 65 cells of a row: cell b = ds + P_b - M_b with P_b / M_b the
// number of +1 / -1 deltas among the first b columns.  Inside byte q of the word every cell is
// >= ds + P_{8q} - M_{8q+8}.  Returns true when some cell MAY be <= k (never false for a live row).
__device__ __forceinline__ bool row_maybe_live(int ds, uint64_t vp, uint64_t vm, int k) {
  const uint32_t pl = lo32(vp), ph = hi32(vp), ml = lo32(vm), mh = hi32(vm);
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
};

__device__ __forceinline__ void emit(const EmitCtx& P, uint64_t gpos, int cost, uint32_t flags) {
  const uint32_t idx = atomicAdd(P.cand_count, 1u);
  if (idx < P.cand_cap) {
    Candidate c;
    c.pos = gpos;
    c.cost = cost;
    c.flags = flags;
    P.cand[idx] = c;
  }
}

// Exact walk over the 64 columns of a block whose last row may contain a cell <= k.
// Same decisions as the reference's find_minima_with_overhang with alpha = None
// (reference: src/search.rs:1286-1369), plus the seam bookkeeping.
//   b: block index inside the buffer; owned: the block belongs to this lane's chunk (reports are
//   emitted) or is warm-up (state only); x0: first local column whose <=k values are exact
//   (-1: all); last_warm: this is the block right before the chunk's first owned block.
__device__ __noinline__ uint32_t scan_block(const EmitCtx P, uint64_t vp, uint64_t vm, int ds,
                                        uint64_t b, bool owned, bool last_warm, int64_t x0,
                                        uint32_t state) {
  bool dec = (state & kStDec) != 0, amb = (state & kStAmb) != 0;
  const int k = (int)P.k;
  const bool all = (P.flags & kScanAllMinima) != 0;
  const uint64_t base = b * 64;
  const uint64_t max_pos = P.text_len;
  if (base >= max_pos) return state;
  int cost = ds, prev_cost = ds;
  uint64_t prev_pos = base;
  if (all && owned && cost <= k && base == 0 && P.global_offset == 0 && (P.flags & kScanTextStart))
    emit(P, 0, cost, 0);
  bool determined = (x0 < 0);
  for (int bit = 1; bit <= 64; ++bit) {
    const uint64_t pos = base + (uint64_t)bit;
    if (pos > max_pos) break;
    cost += (int)((vp >> (bit - 1)) & 1);
    cost -= (int)((vm >> (bit - 1)) & 1);
    if (all) {
      if (owned && cost <= k) emit(P, P.global_offset + pos, cost, 0);
    } else {
      const bool rising = cost > prev_cost, falling = cost < prev_cost;
      if (dec && rising && prev_cost <= k && owned)
        emit(P, P.global_offset + prev_pos, prev_cost, amb ? kCandCond : 0u);
      dec = falling || (dec && !rising);
      const bool event = rising || falling || cost > k || prev_cost > k;
      if (event) {
        if (owned) amb = false;
        else if ((int64_t)pos > x0) determined = true;
      }
    }
    prev_cost = cost;
    prev_pos = pos;
  }
  if (!all) {
    if (last_warm) amb = !determined;
    if (owned && (P.flags & kScanTextEnd) && prev_pos == max_pos && dec && prev_cost <= k)
      emit(P, P.global_offset + prev_pos, prev_cost, amb ? kCandCond : 0u);
  }
  return (dec ? kStDec : 0u) | (amb ? kStAmb : 0u);
}

// 16 text bytes that straddle or lie past the end of the buffer (cold path): bytes past the end
// read as 'X' (reference: src/search.rs:202-207).
__device__ __noinline__ uint4 load_tail16(const uint8_t* text, uint64_t off, uint64_t text_len) {
  uint64_t lo = 0x5858585858585858ull, hi = 0x5858585858585858ull;
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

template <int PROFILE>
__device__ __forceinline__ uint32_t text_xform(uint32_t c, const unsigned char* nibtab) {
  if (PROFILE == PROFILE_DNA) return (c >> 1) & 3u;   // reference: src/profiles/dna.rs:100-102
  if (PROFILE == PROFILE_IUPAC) return nibtab[c & 31u];  // reference: src/profiles/iupac.rs:281-330
  return c;
}
template <int PROFILE>
__device__ __forceinline__ bool slot_test(uint32_t t, uint32_t sv) {
  if (PROFILE == PROFILE_IUPAC) return (t & sv) != 0;  // base sets intersect (iupac.rs:104-126)
  return t == sv;                                      // Dna code / Ascii byte equality
}

// Masks of the 64 blocks of a wave, one block per step: lane i reads byte i of block L from the
// tile, one compare per slot gives the 64-bit equality mask as a wave-uniform value, which is
// written into lane L (the block's owner).  All NS slots are always built (the host pads unused
// slots with a value that never matches) so the body is branch-free.
template <int PROFILE, int NS>
__device__ __forceinline__ void build_masks(const ScanParams& P, const unsigned char* src,
                                            const unsigned char* nibtab, uint32_t (&msk)[NS / 4][8]) {
  static_assert(NS % 4 == 0, "slots are handled in groups of four");
  uint32_t sv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s)  // unused slots: Iupac empty set (0) / a value no byte equals
    sv[s] = (s < (int)P.nslots) ? (uint32_t)P.slot_val[s] : (PROFILE == PROFILE_IUPAC ? 0u : 0x100u);
#pragma unroll 4
  for (int L = 0; L < 64; ++L) {
    const uint32_t c = src[L * 128];
    const uint32_t t = text_xform<PROFILE>(c, nibtab);
#pragma unroll
    for (int g = 0; g < NS / 4; ++g) {
      uint32_t bal[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint64_t b64 = __ballot(slot_test<PROFILE>(t, sv[g * 4 + q]));
        bal[2 * q] = lo32(b64);
        bal[2 * q + 1] = hi32(b64);
      }
      writelane_x8(msk[g], bal, L);
    }
  }
}

// first block a chunk touches: its first owned block minus the warm-up, clipped at the buffer
// start, rounded down to an even block so that a staged pair of blocks is one 128-byte line.
__device__ __forceinline__ uint64_t chunk_blk0(const ScanParams& P, uint64_t chunk) {
  const uint64_t start = P.first_owned_block + chunk * (uint64_t)P.bpl;
  const uint64_t b0 = start > P.wb ? start - P.wb : 0;
  return b0 & ~1ull;
}

template <int PROFILE, int NS>
__global__ __launch_bounds__(256) void scan_kernel(const ScanParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  unsigned char* nibtab = smem;  // 32 bytes used
  unsigned char* wbase = smem + kGroupHeaderBytes + (size_t)wave * P.lds_per_wave;
  unsigned char* tile = wbase;
  unsigned char* mask_bytes = wbase + kTileBytes;
  uint32_t* carry = reinterpret_cast<uint32_t*>(wbase + kTileBytes + NS * 512);  // [word][hp|hm][lane]

  if (PROFILE == PROFILE_IUPAC) {
    // letter index (c & 31) -> low nibble of the IUPAC base set; non-letters act as N (=15), X = 0
    if (threadIdx.x < 32) {
      const unsigned t = threadIdx.x;
      unsigned v = 15;
      switch (t) {
        case 1: v = 1; break;    // A
        case 3: v = 2; break;    // C
        case 20: v = 4; break;   // T
        case 21: v = 4; break;   // U
        case 7: v = 8; break;    // G
        case 14: v = 15; break;  // N
        case 18: v = 9; break;   // R = A|G
        case 25: v = 6; break;   // Y = C|T
        case 19: v = 10; break;  // S = G|C
        case 23: v = 5; break;   // W = A|T
        case 11: v = 12; break;  // K = G|T
        case 13: v = 3; break;   // M = A|C
        case 2: v = 14; break;   // B = C|G|T
        case 4: v = 13; break;   // D = A|G|T
        case 8: v = 7; break;    // H = A|C|T
        case 22: v = 11; break;  // V = A|C|G
        case 24: v = 0; break;   // X
        default: break;
      }
      nibtab[t] = (unsigned char)v;
    }
    __syncthreads();
  }

  const uint64_t wave_chunk0 = ((uint64_t)blockIdx.x * kWavesPerGroup + wave) * kWave;
  if (wave_chunk0 >= P.n_chunks) return;  // wave-uniform
  const uint64_t chunk = wave_chunk0 + lane;

  const uint64_t own_lo = P.first_owned_block + chunk * (uint64_t)P.bpl;
  uint64_t own_hi = own_lo + P.bpl;
  if (own_hi > P.n_blocks) own_hi = P.n_blocks;
  const bool has_chunk = chunk < P.n_chunks && own_lo < P.n_blocks;
  const uint64_t blk0 = chunk_blk0(P, chunk);
  // The chunk's fresh start is the true DP boundary only at column 0 of the whole text.
  const bool exact_start = (blk0 == 0) && (P.flags & kScanTextStart);
  const int64_t x0 = exact_start ? -1 : (int64_t)(blk0 * 64 + P.m + P.k);

  const int k = (int)P.k;
  const uint32_t m = P.m;
  const uint32_t nwords = P.nwords;
  const uint32_t last_rows = m - 32 * (nwords - 1);
  const uint32_t last_word_init = last_rows == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> last_rows);

  // per-row carries between horizontally adjacent blocks, 32 rows per word, row r of a word at
  // bit 31-r.  Fresh start: every vertical delta on the left edge is +1 (D[j][start] = j).
  uint32_t c_hp = nwords == 1 ? last_word_init : 0xFFFFFFFFu, c_hm = 0;  // word 0 in registers
  if (nwords > 1) {
    for (uint32_t w = 0; w < nwords; ++w) {
      carry[(w * 2 + 0) * 64 + lane] = (w == nwords - 1) ? last_word_init : 0xFFFFFFFFu;
      carry[(w * 2 + 1) * 64 + lane] = 0;
    }
  }

  uint32_t st = kStDec;  // dec = true, amb = false
  EmitCtx ctx;
  ctx.cand = P.cand;
  ctx.cand_count = P.cand_count;
  ctx.text_len = P.text_len;
  ctx.global_offset = P.global_offset;
  ctx.cand_cap = P.cand_cap;
  ctx.k = P.k;
  ctx.flags = P.flags;

  unsigned long long cnt_rows = 0, cnt_blocks = 0;

  for (uint32_t it = 0; it < P.n_iter; ++it) {
    const uint32_t sub = it & 1u;
    if (sub == 0) {
      // ---- stage 2 blocks (128 B) for each of the 64 lane chunks: 8 x (64 lanes x 16 B) ----
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t owner = (uint32_t)i * 8u + ((uint32_t)lane >> 3);
        const uint32_t part = (uint32_t)lane & 7u;
        const uint64_t ob = chunk_blk0(P, wave_chunk0 + owner) + it;
        const uint64_t off = ob * 64 + part * 16;
        uint4 v;
        if (off + 16 <= P.text_len) {
          v = *reinterpret_cast<const uint4*>(P.text + off);
        } else {
          // tail of the text: bytes past the end read as 'X' (reference: src/search.rs:202-207)
          v = load_tail16(P.text, off, P.text_len);
        }
        *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
      }
    }

    // ---- profile: one 64-bit equality mask per slot for each lane's block ----
    uint32_t msk[NS / 4][8];  // [slot group][slot-in-group * 2 + (lo|hi)]
#pragma unroll
    for (int g = 0; g < NS / 4; ++g)
#pragma unroll
      for (int q = 0; q < 8; ++q) msk[g][q] = 0;
    build_masks<PROFILE, NS>(P, tile + sub * 64 + lane, nibtab, msk);
#pragma unroll
    for (int s = 0; s < NS; ++s)
      *reinterpret_cast<uint2*>(mask_bytes + s * 512 + lane * 8) =
          make_uint2(msk[s / 4][(s % 4) * 2], msk[s / 4][(s % 4) * 2 + 1]);

    // ---- the DP rows of this block ----
    DpWord V;
    V.vpl = V.vph = V.vml = V.vmh = 0;  // row 0 of the matrix is all 0: horizontal deltas 0
    int ds = 0;                         // cost at the block's left edge in the last row
    const unsigned char* my_masks = mask_bytes + lane * 8;
    for (uint32_t w = 0; w < nwords; ++w) {
      uint32_t ohp, ohm;
      if (nwords == 1) {
        ohp = c_hp; ohm = c_hm;
      } else {
        ohp = carry[(w * 2 + 0) * 64 + lane];
        ohm = carry[(w * 2 + 1) * 64 + lane];
      }
      ds += __popc(ohp) - __popc(ohm);
      const uint32_t rows = (w == nwords - 1) ? last_rows : 32u;
      // Row -> LDS offset of its slot mask.  Constant address space: wave-uniform reads become
      // scalar loads, and for m <= 32 they are loop invariant (hoisted out of the block loop).
      typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;
      const_u32_ptr ro = (const_u32_ptr)(P.row_off) + 32 * w;
      uint32_t roff[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) roff[r] = ro[r];  // the table is padded to 32*nwords entries
      uint32_t nhp = 0, nhm = 0;
      uint32_t done = 0;
      uint2 eqn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) eqn[u] = *reinterpret_cast<const uint2*>(my_masks + roff[u]);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (4u * g + 4u <= rows) {
          uint2 eqc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) eqc[u] = eqn[u];
          if (g < 7) {  // prefetch the next group's Eq words while this group computes
#pragma unroll
            for (int u = 0; u < 4; ++u)
              eqn[u] = *reinterpret_cast<const uint2*>(my_masks + roff[4 * g + 4 + u]);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            dp_row(V, eqc[u], (ohp >> (31 - (4 * g + u))) & 1u, (ohm >> (31 - (4 * g + u))) & 1u, nhp, nhm);
          done = 4u * g + 4u;
        }
      }
      // up to three leftover rows when m is not a multiple of 4
      for (uint32_t r = done; r < rows; ++r) {
        const uint2 eq = *reinterpret_cast<const uint2*>(my_masks + ro[r]);
        dp_row(V, eq, (ohp >> (31 - r)) & 1u, (ohm >> (31 - r)) & 1u, nhp, nhm);
      }
      if (rows < 32) { nhp <<= (32 - rows); nhm <<= (32 - rows); }
      if (nwords == 1) {
        c_hp = nhp; c_hm = nhm;
      } else {
        carry[(w * 2 + 0) * 64 + lane] = nhp;
        carry[(w * 2 + 1) * 64 + lane] = nhm;
      }
    }
    const uint64_t vp = ((uint64_t)V.vph << 32) | V.vpl, vm = ((uint64_t)V.vmh << 32) | V.vml;

    // ---- last row of the block: anything <= k ? ----
    const uint64_t b = blk0 + it;
    const bool active = has_chunk && b < own_hi;
    if (active) {
      if (P.counters) { cnt_rows += m; cnt_blocks += 1; }
      if (row_maybe_live(ds, vp, vm, k)) {
        st = scan_block(ctx, vp, vm, ds, b, b >= own_lo, b + 1 == own_lo, x0, st);
      } else {
        st = kStDec;  // dec = true: a later <=k run can only be entered by a decrease; amb = false
      }
    }
  }

  if (chunk < P.n_chunks) P.chunk_state[chunk] = (st & kStAmb) ? kStatePass : ((st & kStDec) ? kStateDecTrue : kStateDecFalse);
  if (P.counters) {
    atomicAdd(&P.counters[0], cnt_rows);
    atomicAdd(&P.counters[1], cnt_blocks);
  }
}

// ------------------------------------------------------------------ launcher
template <int PROFILE, int NS>
static hipError_t launch_one(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  static bool attr_set = false;  // LDS beyond the 64 KiB default needs an explicit opt-in
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_kernel<PROFILE, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((scan_kernel<PROFILE, NS>), dim3(grid), dim3(256), smem, stream, P);
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
#else
#if SASSY_SCAN_PROFILE == 2
hipError_t launch_scan_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR = PROFILE_IUPAC;
#else
hipError_t launch_scan_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream) {
  constexpr int PR = PROFILE_ASCII;
#endif
  if (P.nslots <= 4) return launch_one<PR, 4>(P, grid, smem, stream);
  if (P.nslots <= 8) return launch_one<PR, 8>(P, grid, smem, stream);
  if (P.nslots <= 16) return launch_one<PR, 16>(P, grid, smem, stream);
  return hipErrorInvalidValue;
}
#endif

}  // namespace sassy_hip
