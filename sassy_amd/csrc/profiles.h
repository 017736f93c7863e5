// profiles.h -- host side of the alphabet profiles (the reference's Profile trait for this path).
//
// Mirrors, per profile, exactly what the reference's Profile implementations expose to the search
// driver: pattern validation + encoding into profile slots, the scan equality, the traceback
// equality (is_match), complement / reverse complement.
//   Dna   : reference src/profiles/dna.rs:14-138
//   Iupac : reference src/profiles/iupac.rs:13-344
//   Ascii : reference src/profiles/ascii.rs:13-73 (case sensitive, what src/c.rs:64 instantiates)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

namespace sassy_hip {

// IUPAC letter (5 low bits) -> base set, A=1 C=2 T=4 G=8; 255 = not a letter; X = empty set
// (reference: src/profiles/iupac.rs:281-317).
inline uint8_t iupac_code(uint8_t c) {
  static const uint8_t tab[32] = {
      /* 0 @ */ 255, /* A */ 1,   /* B */ 14,  /* C */ 2,   /* D */ 13,  /* E */ 255, /* F */ 255,
      /* G */ 8,     /* H */ 7,   /* I */ 255, /* J */ 255, /* K */ 12,  /* L */ 255, /* M */ 3,
      /* N */ 15,    /* O */ 255, /* P */ 255, /* Q */ 255, /* R */ 9,   /* S */ 10,  /* T */ 4,
      /* U */ 4,     /* V */ 11,  /* W */ 5,   /* X */ 0,   /* Y */ 6,   /* Z */ 255, 255,
      255,           255,         255,         255};
  return tab[c & 31];
}

// Scan equality = what the bit-parallel profile encodes.
inline bool scan_eq(Profile pr, uint8_t p, uint8_t t) {
  switch (pr) {
    case PROFILE_DNA: return ((p >> 1) & 3) == ((t >> 1) & 3);
    case PROFILE_IUPAC: return ((iupac_code(p) & iupac_code(t)) & 0x0F) != 0;
    default: return p == t;
  }
}
// Traceback equality = Profile::is_match (dna.rs:48-50, iupac.rs:136-138, ascii.rs:44-51).
inline bool trace_is_match(Profile pr, uint8_t p, uint8_t t) {
  switch (pr) {
    case PROFILE_DNA: return (p | 0x20) == (t | 0x20);
    case PROFILE_IUPAC: return (iupac_code(p) & iupac_code(t)) > 0;
    default: return p == t;
  }
}

// Iupac::valid_seq (iupac.rs:156-204).  Dna / Ascii patterns are never rejected by the reference.
inline bool valid_pattern(Profile pr, const uint8_t* p, size_t m) {
  if (pr != PROFILE_IUPAC) return true;
  for (size_t i = 0; i < m; ++i) {
    const uint8_t c = p[i] & (uint8_t)~0x20;
    if (c <= '@' || c >= 'Z' || iupac_code(c) == 255) return false;
  }
  return true;
}

inline uint8_t complement_char(Profile pr, uint8_t c) {
  if (pr == PROFILE_DNA) {  // dna.rs:121-133: upper-case ACGT only
    switch (c) {
      case 'A': return 'T';
      case 'C': return 'G';
      case 'T': return 'A';
      case 'G': return 'C';
      default: return c;
    }
  }
  // iupac.rs:235-278: IUPAC letters, both cases
  static const char from[] = "ACTGRYSWKMBDHVNX";
  static const char to[] = "TGACYRSWMKVHDBNX";
  for (int i = 0; from[i]; ++i) {
    if (c == (uint8_t)from[i]) return (uint8_t)to[i];
    if (c == (uint8_t)(from[i] | 0x20)) return (uint8_t)(to[i] | 0x20);
  }
  return c;
}

// The encoded pattern: which profile slot every row compares against, and what each slot tests.
struct PatternPlan {
  uint32_t m = 0;
  uint32_t nslots = 0;
  uint32_t nwords = 0;               // ceil(m / 32)
  uint8_t slot_val[kMaxSlots] = {};  // Dna: 2-bit code; Iupac: base-set nibble; Ascii: byte
  std::vector<uint32_t> row_tab;     // one byte per row = 2 * its profile slot, 4 rows per word,
                                     // 8 words per 32 rows (padded with slot 0)
  bool bytes = false;                // Ascii with more than kMaxSlots distinct bytes (PROFILE_ASCII_BYTES): nslots = 8
                                     // (the bit planes), the row table holds the pattern bytes themselves
};

// Profile::encode_pattern (dna.rs:19-23, iupac.rs:18-36, ascii.rs:18-29).
// Returns false with `err` set for what the reference would panic on.
inline bool make_plan(Profile pr, const uint8_t* pat, size_t m, PatternPlan& plan, std::string& err) {
  if (m == 0) { err = "empty pattern"; return false; }
  if (m > (1u << 20)) { err = "pattern longer than 2^20 is not supported"; return false; }
  plan.m = (uint32_t)m;
  plan.nwords = (uint32_t)((m + 31) / 32);
  plan.row_tab.assign((size_t)plan.nwords * 8, 0u);
  auto set_row = [&](size_t j, uint32_t slot) { plan.row_tab[j >> 2] |= (2u * slot) << (8 * (j & 3)); };
  std::vector<uint8_t> letters;
  if (pr == PROFILE_DNA) {
    plan.nslots = 4;
    for (int s = 0; s < 4; ++s) plan.slot_val[s] = (uint8_t)s;
    for (size_t j = 0; j < m; ++j) set_row(j, (uint32_t)((pat[j] >> 1) & 3));
    return true;
  }
  if (pr == PROFILE_IUPAC) {
    if (!valid_pattern(pr, pat, m)) {
      err = "Pattern is not valid IUPAC";  // iupac.rs:19-24 panics with this message
      return false;
    }
    letters = {'A', 'C', 'T', 'G'};
  }
  for (size_t j = 0; j < m; ++j) {
    const uint8_t c = pr == PROFILE_IUPAC ? (uint8_t)(pat[j] & ~0x20) : pat[j];
    size_t s = 0;
    while (s < letters.size() && letters[s] != c) ++s;
    if (s == letters.size()) letters.push_back(c);
    if (s < (size_t)kMaxSlots) set_row(j, (uint32_t)s);
  }
  // Iupac: the reference's profile holds 16 masks and asserts (iupac.rs:69).  Ascii: 256 slots in the
  // reference (ascii.rs:13-29); here every distinct pattern byte takes one LDS mask slot per lane, up to
  // 64 of them; patterns with more distinct bytes run in byte mode (below).
  if (pr == PROFILE_IUPAC && letters.size() > 16) {
    err = "pattern uses more than 16 distinct letters";
    return false;
  }
  if (letters.size() > (size_t)kMaxSlots) {
    // Ascii, more distinct bytes than mask slots (the reference's profile has 256, ascii.rs:13-29): byte mode
    plan.bytes = true;
    plan.nslots = 8;
    plan.row_tab.assign((size_t)plan.nwords * 8, 0u);
    for (size_t j = 0; j < m; ++j) plan.row_tab[j >> 2] |= (uint32_t)pat[j] << (8 * (j & 3));
    return true;
  }
  plan.nslots = (uint32_t)letters.size();
  for (size_t s = 0; s < letters.size(); ++s)
    plan.slot_val[s] = pr == PROFILE_IUPAC ? (uint8_t)(iupac_code(letters[s]) & 0x0F) : letters[s];
  return true;
}

}  // namespace sassy_hip
