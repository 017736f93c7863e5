// scan_driver.hip -- one pattern over one buffer: the scan job (prepare / enqueue / finish: prefilter -> chunk DP -> rank ->
// traceback on one lane), a search cut into sub-shards, the reference-lane mode, both strands, the report filters.
// Mirror of the reference's Searcher<P>::search path (src/search.rs:510-937).
#include "host_internal.h"

PinPool g_pin_pool;

namespace sassy_hip {
thread_local LaunchEvents g_launch_events;

// Prefilter geometry: k+1 disjoint pattern pieces of q rows.  Enabled when the pieces are long
// enough to be selective (expected hit blocks on random DNA: 64*(k+1)/4^q of all blocks).
// mode: the searcher's own setting (sassy_hip_set_prefilter), -1 = the process default (SASSY_HIP_PREFILTER)
static int prefilter_mode(const sassy_SearcherType* S) {
  return S->prefilter >= 0 ? S->prefilter : (int)S->sw.prefilter;
}
static uint32_t filter_piece_len(const PatternPlan& plan, uint32_t k, const sassy_SearcherType* S) {
  const int env = prefilter_mode(S);
  if (env == 0) return 0;
  const uint64_t pieces = (uint64_t)k + 1;
  uint64_t q = plan.m / pieces;
  if (q > 12) q = 12;
  if (q < (env == 1 ? 2u : 7u)) return 0;    // too unselective: stream the full DP instead
  return (uint32_t)q;
}

// The paired filter's geometry for a shape (filter_dna_kernel<.., PAIR>): S = ceil((k+1)/2) super-pieces of two sub-pieces
// of Q = m / (2 S) rows each.  Taken where the k+1 pigeonhole pieces are shorter than 7 rows and Q is 5 or 6 (m = 23, k = 3;
// m = 32, k = 4, 5; m = 12, k = 1; ...).  False: the shape is not one of them.
static bool pair_geometry(uint32_t m, uint32_t k, uint32_t* s_out, uint32_t* q_out) {
  if (k < 1 || m / (k + 1) >= 7) return false;
  const uint32_t s = (k + 2) / 2;
  if (s > 4) return false;
  const uint32_t q = m / (2 * s);
  if (q != 5 && q != 6) return false;
  *s_out = s;
  *q_out = q;
  return true;
}
// rows of the pattern, from row 0 on, that are plain bases
static size_t plain_prefix(const uint8_t* pat, size_t m) {
  size_t j = 0;
  for (; j < m; ++j) {
    const uint8_t u = pat[j] & 0xDFu;
    if (u != 'A' && u != 'C' && u != 'G' && u != 'T') break;
  }
  return j;
}
static bool plain_acgt(const uint8_t* pat, size_t m) { return plain_prefix(pat, m) == m; }


// Bit table of every q-gram (2 bits per char, first piece row most significant; codes A0 C1 T2 G3)
// that some piece accepts; rows with ambiguity letters are expanded.  False if that takes more
// than `limit` q-grams (then the table says nothing useful anyway).
static bool build_qgram_table(Profile pr, const uint8_t* pat, uint32_t q, uint32_t pieces, std::vector<uint8_t>& tab) {
  const size_t limit = 1u << 16;
  tab.assign((size_t)1 << (2 * q - 3), 0);
  const uint32_t low_bits = 2 * q - 3;
  std::vector<uint32_t> cur, nxt;
  size_t total = 0;
  for (uint32_t p = 0; p < pieces; ++p) {
    cur.assign(1, 0u);
    for (uint32_t j = 0; j < q && !cur.empty(); ++j) {
      const uint8_t c = pat[p * q + j];
      // base set of the row as a nibble whose bit index is the 2-bit text code
      const uint32_t set = pr == PROFILE_IUPAC ? (iupac_code(c) & 15u) : (1u << ((c >> 1) & 3u));
      nxt.clear();
      for (uint32_t code : cur)
        for (uint32_t b = 0; b < 4; ++b)
          if ((set >> b) & 1u) nxt.push_back((code << 2) | b);
      if (nxt.size() + total > limit) return false;
      cur.swap(nxt);
    }
    total += cur.size();
    for (uint32_t code : cur) tab[code & ((1u << low_bits) - 1u)] |= (uint8_t)(1u << (code >> low_bits));
  }
  return true;
}

// The counting filter's table (count_filter.hip): H = every Q-gram some Q consecutive pattern rows
// accept (2 bits per letter, first row most significant, codes A0 C1 T2 G3; ambiguous rows are
// expanded); entry w of the table, w a (Q+R-1)-gram, = how many of the R Q-grams w ends with are
// in H.  density = |H| / 4^Q, the chance that a random position counts.  False if the expansion
// takes more than `limit` Q-grams.
static bool build_count_table(Profile pr, const uint8_t* pat, const uint8_t* pat2, uint32_t m, uint32_t Q, uint32_t R,
                              std::vector<uint8_t>& tab, double* density) {
  const size_t limit = 1u << 20;
  const uint32_t nq = 1u << (2 * Q);
  std::vector<uint8_t> H(nq, 0);
  std::vector<uint32_t> cur, nxt;
  size_t total = 0;
  // pat2: a second pattern whose q-grams also count (the Rc strand's, in forward orientation)
  for (uint32_t o = 0; o + Q <= (pat2 ? 2 * m : m); ++o) {
    if (o + Q > m && o < m) continue;  // no q-gram across the two patterns
    cur.assign(1, 0u);
    for (uint32_t j = 0; j < Q; ++j) {
      const uint8_t c = o < m ? pat[o + j] : pat2[o - m + j];
      const uint32_t set = pr == PROFILE_IUPAC ? (iupac_code(c) & 15u) : (1u << ((c >> 1) & 3u));
      nxt.clear();
      for (uint32_t code : cur)
        for (uint32_t b = 0; b < 4; ++b)
          if ((set >> b) & 1u) nxt.push_back((code << 2) | b);
      if (nxt.size() + total > limit) return false;
      cur.swap(nxt);
    }
    total += cur.size();
    for (uint32_t code : cur) H[code] = 1;
  }
  size_t set_bits = 0;
  for (uint8_t v : H) set_bits += v;
  *density = (double)set_bits / (double)nq;
  const uint32_t nw = 1u << (2 * (Q + R - 1));
  tab.assign(nw, 0);
  for (uint32_t w = 0; w < nw; ++w) {
    uint32_t c = 0;
    for (uint32_t r = 0; r < R; ++r) c += H[(w >> (2 * r)) & (nq - 1)];
    tab[w] = (uint8_t)c;
  }
  return true;
}

// How often a window of random text reaches the threshold t when it holds lambda q-gram hits on
// average.  Hits come in clumps (a text stretch that equals L >= Q pattern rows gives L - Q + 1 of
// them): clumps arrive Poisson(lambda (1 - r)) with geometric sizes, P(j) = (1 - r) r^(j-1), r = 1/4
// the chance that the next letter extends the stretch.  P(S >= t) by Panjer's recursion.
static double clumped_tail(double lambda, uint32_t t) {
  if (t == 0) return 1.0;
  if (lambda <= 0) return 0.0;
  if (lambda >= (double)t) return 1.0;  // at or above the mean: no filter
  const double r = 0.25, lc = lambda * (1.0 - r);
  std::vector<double> p(t, 0.0);
  p[0] = std::exp(-lc);
  if (p[0] <= 0) return 1.0;
  double below = p[0];
  for (uint32_t s = 1; s < t; ++s) {
    double acc = 0, g = 1.0 - r;  // g = P(size j)
    for (uint32_t j = 1; j <= s && j <= 48; ++j, g *= r) acc += (double)j * g * p[s - j];
    p[s] = lc / (double)s * acc;
    below += p[s];
  }
  return std::min(1.0, std::max(0.0, 1.0 - below));
}

static hipError_t launch_scan_any(Profile pr, const ScanParams& P, uint32_t grid, size_t smem, hipStream_t st) {
  switch (pr) {
    case PROFILE_DNA: return launch_scan_dna(P, grid, smem, st);
    case PROFILE_IUPAC: return launch_scan_iupac(P, grid, smem, st);
    default: return launch_scan_ascii(P, grid, smem, st);
  }
}
static hipError_t launch_filter_any(Profile pr, const ScanParams& P, uint32_t grid, size_t smem, hipStream_t st) {
  switch (pr) {
    case PROFILE_DNA: return launch_filter_dna(P, grid, smem, st);
    case PROFILE_IUPAC: return launch_filter_iupac(P, grid, smem, st);
    default: return launch_filter_ascii(P, grid, smem, st);
  }
}
static hipError_t launch_list_any(Profile pr, const ScanParams& P, uint32_t grid, size_t smem, hipStream_t st) {
  switch (pr) {
    case PROFILE_DNA: return launch_list_dna(P, grid, smem, st);
    case PROFILE_IUPAC: return launch_list_iupac(P, grid, smem, st);
    default: return launch_list_ascii(P, grid, smem, st);
  }
}

// Chunk geometry of a streaming kernel: enough lanes to fill 256 CUs several times over, chunks
// long enough that the extra blocks in front of each chunk stay a few percent of the work.
// wpc: resident waves per CU of the kernel (its workgroups are launched in two full rounds)
int stream_geometry(ScanParams& P, uint64_t owned, uint32_t extra_front, uint32_t* grid, int wpc,
                           GeoTuner* tuner, const void* tune_text, uint64_t tune_len,
                           uint32_t tune_kind) {
  const uint64_t target_lanes = 256ull * wpc * 64 * 2;
  uint64_t bpl = (owned + target_lanes - 1) / target_lanes;
  const uint64_t min_bpl = std::max<uint64_t>(8, 6ull * extra_front);
  if (bpl < min_bpl) bpl = min_bpl;
  bpl += bpl & 1;  // even: a staged pair of blocks is then always one aligned 128-byte line
  // The lanes of a wave read addresses bpl * 64 bytes apart and the chip holds ~1.6 rounds of the grid:
  // both the stride and the lane count decide how evenly the HBM channels are loaded, and the kernel
  // time is sensitive to it (bit-plane filter, % of the 8 TB/s roofline: 3.0 GB bpl 88 / 90 / 92 / 94 ->
  // 51 / 64 / 64 / 53; 2.7 GB 78 / 82 / 84 -> 61 / 53 / 58; 2.0 GB 58 / 60 / 64 -> 54 / 63 / 47).  No static
  // rule fits every size (multiples of 6 blocks are never bad but not always best): GeoTuner tries the
  // neighbouring even values during the first searches of a resident text and keeps the fastest (opt-in: `tune`).
  // (opt-in since round 2: with two searches in flight -- the way a stream of searches runs -- the geometry
  // moves the time per search by 0-2 %; it still matters for the latency of a lone search at unlucky sizes,
  // 2.7 GB: 0.80 -> 0.72 ms, which is what SASSY_HIP_TUNE=1 is for; profiles/r02_geometry_sweep.txt)
  // (the callers pass a tuner only when the searcher asks for one: sassy_hip_set_geometry_tuner / SASSY_HIP_TUNE=1)
  if (tuner != nullptr && owned * 64 >= (256ull << 20)) {
    const uint32_t t = tuner->next(tune_text, tune_len, owned, tune_kind, extra_front, (uint32_t)bpl, (uint32_t)(min_bpl + (min_bpl & 1)));
    if (t) bpl = t;
  }
  if (bpl > 0xFFFFFFFFull / 2) return fail(SASSY_HIP_EUNSUPPORTED, "text too large for one launch");
  P.bpl = (uint32_t)bpl;
  P.n_chunks = (owned + bpl - 1) / bpl;
  P.n_iter = extra_front + 1 + P.bpl;
  const uint64_t groups = (P.n_chunks + 255) / 256;
  if (groups > 0x7FFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "grid too large");
  *grid = (uint32_t)groups;
  return 0;
}

int ScanJob::prepare() {
  t_enter = now_ms();
  L.h_up_used = 0;  // the lane's previous job is finished: its upload staging is free again
  // overhang (reference: get_overhang_steps, src/search.rs:347-356): the text is virtually extended
  // by ov_steps 'N' columns, f32 arithmetic as there
  const bool overhang = !std::isnan(S->alpha);
  uint32_t ov_steps = 0;
  if (overhang && sh.text_end) {
    uint64_t st = plan.m;
    if (S->alpha > 0.0f) {
      const float qf = std::ceil(((float)k + S->alpha) / S->alpha);
      if (qf < (float)st) st = (uint64_t)qf;
    }
    if (S->max_overhang >= 0) st = std::min<uint64_t>(st, (uint64_t)S->max_overhang);
    ov_steps = (uint32_t)st;
  }
  n_blocks = (sh.text_len + ov_steps + 63) / 64;
  first_owned = sh.halo_len / 64;
  if (n_blocks <= first_owned) { empty = true; return 0; }  // nothing owned (empty text)
  if (n_blocks > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "text longer than 2^38 bytes per buffer");
  owned = n_blocks - first_owned;

  P = ScanParams{};
  P.text = sh.d_text;
  P.text_len = sh.text_len;
  P.n_blocks = n_blocks;
  P.first_owned_block = first_owned;
  P.global_offset = sh.global_offset;
  P.m = plan.m;
  P.k = k;
  P.nwords = plan.nwords;
  P.nslots = plan.nslots;
  P.profile = plan.bytes ? PROFILE_ASCII_BYTES : (uint32_t)S->profile;
  P.wb = warmup_blocks(plan.m, k);
  P.flags = (all_minima ? kScanAllMinima : 0u) | (sh.text_start ? kScanTextStart : 0u) |
            (sh.text_end ? kScanTextEnd : 0u) | (overhang ? kScanOverhang : 0u);
  const Switches& sw = S->sw;
  if (sw.row_cut == 0) P.flags |= kScanNoRowCut;
  P.alpha = overhang ? S->alpha : 0.0f;
  P.ov_steps = ov_steps;
  P.rev_n = rev_n;
  bucket = plan.nslots <= 4 ? 4 : plan.nslots <= 8 ? 8 : plan.nslots <= 16 ? 16 : plan.nslots <= 32 ? 32 : 64;
  const int env_sb = (int)sw.stage_blocks;
  P.stage_blocks = env_sb == 1 || env_sb == 2 ? (uint32_t)env_sb : 1u;
  for (int s = 0; s < kMaxSlots; ++s) P.slot_val[s] = plan.slot_val[s];
  q = filter_piece_len(plan, k, S);
  // a match that hangs over an end of the text contains only part of the pattern: the pigeonhole
  // argument of the prefilter does not cover it, so overhang searches stream the full DP
  if (overhang) q = 0;
  // Ascii patterns with more than 16 distinct bytes: only the DP kernels carry that many slot masks (or, byte mode,
  // compare bytes instead of looking slots up)
  if (plan.nslots > 16 || plan.bytes) q = 0;
  if (ext_bitmap) q = ext_q;
  if (ext_desc) q = 1;  // list mode without a filter
  // which prefilter kernel (SASSY_HIP_FILTER_KIND=1|2|3|4 forces one where it applies)
  const int env_kind = (int)sw.filter_kind;
  const int env_pre = prefilter_mode(S);
  fkind = kFilterGeneric;
  if (ext_bitmap || ext_desc) fkind = kFilterPlanes;  // (ext_bitmap: marked like filter_dna_kernel does)
  if (ext_desc) {
    P.flags |= kScanPerText;
    P.texts_start = texts.start;
    P.texts_len = texts.len;
  }
  const uint32_t pieces = k + 1;
  // The fused launch (filter + chunk DP in one kernel, see below) takes one strand of one text whose reports the
  // traceback waves rank themselves.  (trace_wave_ok mirrors use_wave of the traceback set-up further down.)
  const int env_selfrank0 = (int)sw.self_rank, env_lin0 = (int)sw.filter_linear, env_wave0 = (int)sw.trace_wave;
  const bool trace_wave_ok = [&] {
    const uint64_t cell = (k + 1 <= 255) ? 1 : 2;
    const uint64_t band = ((uint64_t)(plan.m + 1) * (2ull * k + 3) * cell + 3) / 4 * 4;
    const uint64_t raw = band + ((uint64_t)plan.m + k + 15 + 15) / 16 * 16 + ((uint64_t)plan.m + k + 1 + 3) / 4 * 4 +
                         ((2ull * (plan.m + k + 1) + 2 + 15) / 16 * 16);
    return env_wave0 != 0 && 2ull * k + 3 <= 64 && 4 * (((uint64_t)plan.m + 15) / 16 * 16) + 4 * ((raw + 15) / 16 * 16) <= 160 * 1024;
  }();
  const bool fuse_ok = !ext_bitmap && !ext_desc && rc_bitmap == nullptr && rev_n == 0 && S->fuse && !no_fuse &&
                       L.fuse_backoff == 0 && env_lin0 <= 0 && env_selfrank0 != 0 && do_trace && trace_wave_ok &&
                       texts.n == 0 && plan.nwords <= 8 && n_blocks < 0x7FFFFFFFull && !S->want_counters;
  // Iupac searcher, pattern of plain A C G T, <= 4 pieces: the Dna bit-plane filter with a check of the text
  // (filter_dna_kernel, CHECK) -- as the fused launch only.  Where the text holds other letters (N runs, ambiguity codes,
  // anything) the lane that owns the block queues the columns a match touching them can end in, like a piece
  // occurrence, and the chunk DP of such a launch builds the Iupac profile's masks: exact on any text.
  const int env_iupac_planes = (int)sw.iupac_planes;
  bool plain_pattern = S->profile == PROFILE_IUPAC && env_iupac_planes != 0 && !overhang;
  for (uint32_t j = 0; plain_pattern && j < plan.m; ++j) {
    const uint8_t u = pat[j] & 0xDFu;
    plain_pattern = u == 'A' || u == 'C' || u == 'G' || u == 'T';
  }
  bool iupac_planes = plain_pattern && fuse_ok && q >= 6 && q <= 12 && pieces <= 4 && plan.nslots <= 4;
  bool can_planes = q > 0 && pieces <= 8 && (S->profile == PROFILE_DNA || iupac_planes);
  // Pieces of 6 rows, at most four of them, where the q-gram counting filter below finds nothing selective (m = 24, k = 3;
  // m = 18, k = 2; m = 12, k = 1): a window chunk in every sixteenth block is still less work for the fused launch than
  // the streaming DP over every block -- 0.85 against 1.03 ms per 3 GB (Iupac searcher: 0.94 against 1.29), m = 12, k = 1 with
  // its 13 764 matches 0.99 against 1.21.  Where the counting filter applies it stays (a 20-mer with k = 2: 0.76 against 0.79;
  // m = 27, k = 3: 0.72 against 0.87); five pieces, or pieces of 5 rows, lose against the streaming DP
  // (tools/probe_short_pieces.py).  SASSY_HIP_SHORT_PIECES=0: never.
  const bool env_short = sw.short_pieces != 0;
  const bool short_ok = q == 0 && env_pre < 0 && env_short && fuse_ok && !overhang && !ext_bitmap && !ext_desc && plan.nslots <= 16 &&
                        !plan.bytes && (S->profile == PROFILE_DNA || plain_pattern) && pieces <= 4 && plan.m / pieces == 6;
  // (5-row pieces lose everywhere: m = 11, k = 1 takes 2.6 ms against 1.7 on the streaming DP, m = 15, k = 2 2.2 against 1.2)
  // The paired filter (filter_dna_kernel<.., PAIR>): S = ceil((k+1)/2) super-pieces of 2 Q rows, each with at most one of
  // the k edits -- one half exact, the other half with <= 1 edit right next to it, tested on the bit planes the lane
  // holds.  For the shapes whose k+1 pigeonhole pieces are 5 or 6 rows (m = 23, k = 3; m = 32, k = 4, 5; ...): the fused
  // launch, and only it (what it cannot finish goes to the paths below, as before).  SASSY_HIP_PAIR=0: never; 2: the
  // q-gram counting filter keeps the shapes it is selective for.
  const int env_pair = (int)sw.pair;
  uint32_t pair_s = 0, pair_q = 0;
  const bool pair_ok = env_pair != 0 && q == 0 && env_pre < 0 && fuse_ok && !overhang && !ext_bitmap && !ext_desc && !plan.bytes &&
                       pair_geometry(plan.m, k, &pair_s, &pair_q) &&
                       // (an Iupac searcher: the filter's 2 S Q rows are plain bases -- the rows behind them may hold
                       // ambiguity letters, a guide's NGG: the chunk DP then builds up to eight slot masks)
                       (S->profile == PROFILE_DNA ||
                        (S->profile == PROFILE_IUPAC && env_iupac_planes != 0 && pair_s <= 3 &&
                         plain_prefix(pat, plan.m) >= (size_t)2 * pair_s * pair_q &&
                         (plan.nslots <= 4 || (plan.nslots <= 8 && plan.nwords <= 4)))) &&
                       (env_kind == 0 || env_kind == kFilterPlanes);
  pair = 0;
  // q-gram counting (count_filter.hip): per (Q, R) variant the threshold t = m + 1 - (k+1) Q, the
  // window W, and how often a window of random text reaches t by chance (the pattern's q-grams,
  // ambiguity letters expanded, against 4^Q; Poisson tail).  Taken when that beats the expected
  // hit blocks of the k+1 pieces, except where the cheaper bit-plane kernel applies (one strand: both
  // strands in one pass cost the bit-plane kernel 8 pieces, 0.85 ms per 3 GB, the counting kernel nothing extra).
  count_r = 0;
  if (!overhang && !ext_bitmap && !ext_desc && S->profile != PROFILE_ASCII && env_pre != 0 &&
      (env_kind == 0 || env_kind == kFilterCount) && !(can_planes && env_kind == 0 && rc_bitmap == nullptr) &&
      !(pair_ok && env_pair != 2)) {
    // two positions per lookup first (half the LDS traffic of (7,1)); the 7-gram variant only where
    // the shorter q-grams are not selective enough
    static const uint32_t variants[][2] = {{6, 2}, {5, 2}, {7, 1}};
    double best = 1.0;
    uint32_t bq = 0, br = 0;
    // the same pattern as in the last call on this lane: the decision and the table are still there
    const bool with_rc = rc_bitmap != nullptr;
    std::vector<uint8_t> rc_fwd;  // the Rc strand's pattern as it reads on the forward text: reversed
    if (with_rc) rc_fwd.assign(std::reverse_iterator<const uint8_t*>(rc_pat + plan.m), std::reverse_iterator<const uint8_t*>(rc_pat));
    const bool same_as_last = L.table_r != 0 && L.table_k == k && L.table_profile == (int)S->profile && L.table_rc == with_rc &&
                              L.table_pattern.size() == plan.m && memcmp(L.table_pattern.data(), pat, plan.m) == 0;
    if (same_as_last) { bq = L.table_q; br = L.table_r; best = 0.0; }
    for (const auto& v : variants) {
      if (same_as_last) break;
      const uint32_t Q = v[0];
      if (v[1] == 1 && best < 1e-3) break;
      if ((uint64_t)pieces * Q > plan.m) continue;  // t >= 1
      const uint32_t t = plan.m + 1 - pieces * Q;
      const uint32_t W = (plan.m + k - Q + 63) / 64 + 1;
      if (W > 64) continue;
      double grams = 0;  // expected size of H: the product of the rows' base-set sizes, per q-gram
      for (uint32_t o = 0; o + Q <= plan.m; ++o) {
        double e = 1;
        for (uint32_t j = 0; j < Q; ++j)
          e *= S->profile == PROFILE_IUPAC ? (double)__builtin_popcount(iupac_code(pat[o + j]) & 15u) : 1.0;
        grams += e;
      }
      const double dens = std::min(1.0, (with_rc ? 2.0 : 1.0) * grams / std::pow(4.0, (double)Q));
      const double tail = clumped_tail(64.0 * W * dens, t);
      if (tail < (v[1] == 1 ? 0.1 * best : best)) { best = tail; bq = Q; br = v[1]; }
    }
    // (the piece-table kernel this competes with is the slower kernel -- 1.0 against 0.64 ms per 3 GB -- so a
    // modest candidate rate is enough; beyond ~5 % of the blocks the chunk DP behind it would dominate)
    if (bq && best < 0.05) {
      const bool cached = L.table_q == bq && L.table_r == br && L.table_k == k && L.table_profile == (int)S->profile &&
                          L.table_rc == with_rc &&
                          L.table_pattern.size() == plan.m && memcmp(L.table_pattern.data(), pat, plan.m) == 0;
      bool ok = true;
      if (!cached) {
        ok = build_count_table(S->profile, pat, with_rc ? rc_fwd.data() : nullptr, plan.m, bq, br, L.h_table, &L.table_density);
        if (ok) {
          if (int rc = L.d_table.reserve(L.h_table.size())) return rc;
          if (int rc = L.upload(L.d_table.p, L.h_table.data(), L.h_table.size())) return rc;
          L.table_q = bq; L.table_r = br; L.table_k = k; L.table_profile = (int)S->profile;
          L.table_rc = with_rc;
          L.table_pattern.assign(pat, pat + plan.m);
        } else {
          L.table_q = 0;
        }
      }
      if (ok) {
        fkind = kFilterCount;
        rc_marked = with_rc;
        q = bq;
        count_r = br;
        count_w = (plan.m + k - bq + 63) / 64 + 1;
        count_t = plan.m + 1 - pieces * bq;
        count_tail = clumped_tail(64.0 * count_w * L.table_density, count_t);
      }
    }
  }
  if (pair_ok && fkind != kFilterCount) {
    pair = pair_s;
    q = pair_q;
    iupac_planes = S->profile == PROFILE_IUPAC;
    can_planes = true;
  } else if (short_ok && fkind != kFilterCount) {
    q = plan.m / pieces;
    iupac_planes = plain_pattern && plan.nslots <= 4;
    can_planes = S->profile == PROFILE_DNA || iupac_planes;
  }
  filtered = q > 0;
  if (filtered && !ext_bitmap && !ext_desc && fkind != kFilterCount) {
    const bool can_table = S->profile != PROFILE_ASCII && q >= 7;
    const bool can_generic = (uint64_t)pieces * q <= 255;   // its term table holds 256 piece rows
    if (can_planes && (env_kind == 0 || env_kind == kFilterPlanes)) fkind = kFilterPlanes;
    else if (can_table && (env_kind == 0 || env_kind == kFilterTable || !can_generic)) fkind = kFilterTable;
    else if (!can_generic) { q = 0; filtered = false; }  // too many piece rows for any filter: stream the full DP
    if (fkind == kFilterTable) {
      const uint32_t tq = std::min<uint32_t>(q, 9);
      const bool cached = L.table_q == tq && L.table_r == 0 && L.table_k == k && L.table_profile == (int)S->profile &&
                          L.table_pattern.size() == plan.m && memcmp(L.table_pattern.data(), pat, plan.m) == 0;
      if (!cached) {
        if (build_qgram_table(S->profile, pat, tq, pieces, L.h_table)) {
          if (int rc = L.d_table.reserve(L.h_table.size())) return rc;
          if (int rc = L.upload(L.d_table.p, L.h_table.data(), L.h_table.size())) return rc;
          L.table_q = tq; L.table_r = 0; L.table_k = k; L.table_profile = (int)S->profile;
          L.table_pattern.assign(pat, pat + plan.m);
        } else {
          L.table_q = 0;
          fkind = kFilterGeneric;
          if (!can_generic) { q = 0; filtered = false; }
        }
      }
      if (fkind == kFilterTable) q = tq;
    }
  }

  // pattern-dependent device data is uploaded only when the pattern changed since the last call
  if (int rc = L.d_rowoff.reserve(plan.row_tab.size())) return rc;
  if (int rc = L.d_pattern.reserve(plan.m)) return rc;
  {
    const bool same = L.up_profile == (int)S->profile && L.up_pattern.size() == plan.m &&
                      memcmp(L.up_pattern.data(), pat, plan.m) == 0 && L.up_rowtab == plan.row_tab;
    if (!same) {
      L.up_pattern.assign(pat, pat + plan.m);
      L.up_rowtab = plan.row_tab;
      L.up_profile = (int)S->profile;
      // the sources must stay valid until the copies ran: use the searcher-owned copies
      if (int rc = L.upload(L.d_rowoff.p, L.up_rowtab.data(), L.up_rowtab.size() * sizeof(uint32_t))) return rc;
      if (int rc = L.upload(L.d_pattern.p, L.up_pattern.data(), plan.m)) return rc;
    }
  }
  if (overhang) {
    // left-edge vertical deltas at the text start: floor((i+1) alpha) - floor(i alpha) for the first
    // min(m, max_overhang) rows, 1 below (reference: src/search.rs:1713-1731); row r of word w at bit 31-r
    std::vector<uint32_t> tab(plan.nwords, 0u);
    const uint64_t mo = S->max_overhang >= 0 ? (uint64_t)S->max_overhang : UINT64_MAX;
    for (uint32_t i = 0; i < plan.m; ++i) {
      uint32_t d = 1;
      if (i < mo) d = (uint32_t)((uint64_t)std::floor((float)(i + 1) * S->alpha) - (uint64_t)std::floor((float)i * S->alpha));
      tab[i >> 5] |= (d & 1u) << (31 - (i & 31));
    }
    if (int rc = L.d_ovtab.reserve(plan.nwords)) return rc;
    if (tab != L.up_ovtab) {
      L.up_ovtab = tab;
      if (int rc = L.upload(L.d_ovtab.p, L.up_ovtab.data(), plan.nwords * sizeof(uint32_t))) return rc;
    }
    P.ov_tab = L.d_ovtab.p;
  }
  // One zero-initialised device area per call, cleared by a single memset:
  //   [0, 64)   control block: u32 [0] reports, [1] chunk descriptors | +16: u64 counters
  //             [0] word rows, [1] blocks, [2] hit blocks
  //   [64, ..)  rank counters of the first kRankLimit reports
  //   [kCtlHead, ..)  the prefilter's hit bitmap (one bit per text block)
  // The counting filter of ONE strand files its chunk descriptors itself (count_filter.hip, DIRECT): no hit bitmap (and no
  // 6 MB memset per 3 GB), no chunk-list launch.  Switch count_fused = 0: bitmap + build_chunks_kernel as before.
  count_direct = filtered && fkind == kFilterCount && !rc_marked && !ext_bitmap && !ext_desc && sw.count_fused != 0 &&
                 sw.count_stage_blocks != 1 && n_blocks < 0xFFFFFFFFull && !no_fuse && L.fuse_backoff == 0;
  n_words = filtered && !count_direct ? (n_blocks + 63) / 64 : 0;
  {
    // this search takes the lane's other control block (see ScanLane::d_ctl_twin)
    L.ctl_cur ^= 1;
    DevBuf<uint8_t>& C = L.ctl_cur ? L.d_ctl_twin : L.d_ctl;
    const size_t cap0 = C.cap;
    if (int rc = C.reserve(kCtlHead + (filtered && !ext_bitmap && !ext_desc && !count_direct ? (n_words + 2) * 8 : 0))) return rc;
    if (C.cap != cap0) L.ctl_clean[L.ctl_cur] = false;  // (a new allocation)
    ctl_base = C.p;
    ctl_pre_cleared = L.ctl_clean[L.ctl_cur];
    L.ctl_clean[L.ctl_cur] = false;  // (in use from here on)
  }
  d_bitmap = ext_bitmap ? ext_bitmap : reinterpret_cast<unsigned long long*>(ctl_base + kCtlHead);
  if (L.d_cand.cap == 0)
    if (int rc = L.d_cand.reserve(1u << 16)) return rc;
  d_counts = reinterpret_cast<uint32_t*>(ctl_base);
  d_counters = reinterpret_cast<unsigned long long*>(ctl_base + 16);
  P.row_tab = L.d_rowoff.p;
  P.cand_count = d_counts;
  P.counters = S->want_counters ? d_counters : nullptr;

  // device traceback (K3) runs right behind the scan on the same stream: one host sync per strand.
  // Two kernel shapes, chosen on the device by the number of reports (each launch returns at once
  // when the count is outside its window):
  //   Tw  one wavefront per report  -- latency-optimal, up to kTraceWaveMax reports (needs a band of
  //       <= 64 columns and four slices in LDS);
  //   Tt  one thread per report     -- throughput-optimal for dense results (k <= 6: band row in
  //       registers), and the only shape for very wide bands.
  T = TraceParams{};
  Tw = TraceParams{};
  trace_blocks = wave_blocks = 0;
  use_wave = use_thread = false;
  if (do_trace) {
    const uint64_t cell = (k + 1 <= 255) ? 1 : 2;
    const uint64_t band = ((uint64_t)(plan.m + 1) * (2ull * k + 3) * cell + 3) / 4 * 4;
    const uint64_t win = ((uint64_t)plan.m + k + 15 + 15) / 16 * 16;  // whole 16-byte chunks
    const uint64_t opsb = ((uint64_t)plan.m + k + 1 + 3) / 4 * 4;
    const uint64_t strb = ((2ull * (plan.m + k + 1) + 2 + 15) / 16 * 16);  // = T.str_stride
    const uint64_t raw = band + win + opsb + strb;
    const uint64_t pat_bytes = ((uint64_t)plan.m + 15) / 16 * 16;
    const int env_wave = (int)sw.trace_wave;
    // (+ 128: a byte pair per lane behind the slice -- the band rows' lanes outside the band store there, unmasked)
    const uint64_t wstride = (raw + kTraceWaveDummy + 15) / 16 * 16;
    use_wave = env_wave != 0 && 2ull * k + 3 <= 64 && 4 * pat_bytes + 4 * wstride <= 160 * 1024;
    use_thread = !use_wave || (k <= 6 && !overhang);  // overhang: wave shape or the generic thread shape
    uint64_t stride = raw;
    if ((stride / 4) % 2 == 0) stride += 4;  // odd number of LDS words: conflict-free slices
    if (stride > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "pattern/k too large for the traceback band");
    if (use_thread) {
      // Threads of the thread-per-report launch.  With the slices in LDS (64 per workgroup) as many workgroups as the chip
      // holds at once -- a dense result (10^5 .. 10^6 reports) is bound by how many reports are in flight: 16 384 threads
      // were one wave on a quarter of the SIMDs, 2.3 ms for 743 000 reports.  Slices in global memory: 256 MB of them.
      const int env_tt = (int)sw.trace_threads;
      const bool slices_in_lds = 64 * stride + pat_bytes <= kTraceLdsLimit;
      uint64_t nthreads = (256ull << 20) / stride;
      uint64_t cap_threads = 16384;
      if (slices_in_lds) cap_threads = std::min<uint64_t>(256ull * 64ull * std::max<uint64_t>(1, (160ull * 1024) / (64 * stride + pat_bytes)), 131072);
      if (env_tt >= 64) cap_threads = (uint64_t)env_tt;
      nthreads = std::max<uint64_t>(64, std::min<uint64_t>(cap_threads, slices_in_lds ? cap_threads : nthreads)) / 64 * 64;
      trace_blocks = (uint32_t)(nthreads / 64);
      if (64 * stride + pat_bytes > kTraceLdsLimit)  // slices in global memory
        if (int rc = L.d_scratch.reserve(nthreads * stride)) return rc;
    }
    wave_blocks = 1024;  // 4096 wavefronts, grid-stride over the reports
    T.band_bytes = (uint32_t)band;
    T.win_bytes = (uint32_t)win;
    T.text = sh.d_text;
    T.rev_n = rev_n;
    T.global_offset = sh.global_offset;
    T.total_len = total_len;
    T.cand_count = d_counts;
    T.m = plan.m;
    T.k = k;
    T.profile = (uint32_t)S->profile;
    T.pattern = L.d_pattern.p;
    T.scratch = L.d_scratch.p;
    T.scratch_stride = (uint32_t)stride;
    T.str_stride = (2 * (plan.m + k + 1) + 2 + 15) / 16 * 16;
    T.ops_bytes = (uint32_t)opsb;
    T.use_alpha = overhang ? 1u : 0u;
    T.alpha = overhang ? S->alpha : 0.0f;
    T.max_overhang = S->max_overhang >= 0 ? (uint32_t)std::min<long>(S->max_overhang, 0x7FFFFFFF) : 0xFFFFFFFFu;
    T.wave_mode = 0;
    T.count_min = use_wave ? kTraceWaveMax : 0;   // runs when count_min < count <= count_max
    T.count_max = 0xFFFFFFFFu;
    Tw = T;
    Tw.wave_mode = 1;
    Tw.scratch_stride = (uint32_t)wstride;
    Tw.count_min = 0;
    Tw.count_max = use_thread ? kTraceWaveMax : 0xFFFFFFFFu;
  }

  // ---- one launch for filter + chunk DP?  (bit-plane filter, one strand, one text, reports ranked by the
  // traceback waves themselves; the chunk DP's masks and carries must fit the filter's 8 KiB tile)
  fused = filtered && fkind == kFilterPlanes && fuse_ok && use_wave;
  if (S->profile == PROFILE_IUPAC && fkind == kFilterPlanes && !fused && !ext_bitmap && !ext_desc)
    return fail(SASSY_HIP_EUNSUPPORTED, "internal: the Iupac bit-plane filter exists as the fused launch only");
  if (L.fuse_backoff && !no_fuse) --L.fuse_backoff;

  // ---- geometry of the streaming kernel (full DP, or the prefilter) ----
  grid = 0;
  F = P;           // prefilter launch
  fgrid = 0;
  if (!filtered) {
    tuned = S->tune && S->timing >= 1 && !ext_desc;  // (level 1 times the streaming DP when there is no filter)
    // long patterns: the per-row carries (64 bytes per 32 rows and lane) of four waves no longer fit a workgroup's
    // 160 KiB of LDS -- fewer waves per workgroup then; beyond ~9 800 rows not even one wave's: the carries go to global
    // memory (scan_kernel<.., GC>: 512 bytes per pattern word and wave)
    const bool gc = 4096u * P.stage_blocks + bucket * 512u + (size_t)plan.nwords * 512u > 160 * 1024;
    if (gc) P.stage_blocks = 1;
    if (int rc = stream_geometry(P, owned, P.wb, &grid, 16, tuned ? &S->tuner_scan : nullptr, sh.d_text, sh.text_len,
                                 1000u + plan.nwords)) return rc;
    P.lds_per_wave = 4096u * P.stage_blocks + bucket * 512u + (gc ? 0u : plan.nwords * 512u);
    P.waves_per_group = (uint32_t)std::min<size_t>(kWavesPerGroup, (160 * 1024) / P.lds_per_wave);
    grid = (uint32_t)((P.n_chunks + 64ull * P.waves_per_group - 1) / (64ull * P.waves_per_group));
    if (gc) {
      if (int rc = L.d_carry.reserve((size_t)grid * P.waves_per_group * plan.nwords * 128u)) return rc;
      P.carry_global = L.d_carry.p;
    }
    if (int rc = L.d_state.reserve(P.n_chunks)) return rc;
    P.chunk_state = L.d_state.p;
  } else {
    // K0 also looks at the last halo blocks: a piece that ends there can belong to a match that
    // ends in the first owned blocks, and K0b must know whether the block left of the first
    // owned one is affected.
    const uint64_t look = std::min<uint64_t>(first_owned, (uint64_t)P.wb + 2);
    F.first_owned_block = first_owned - look;
    F.n_pieces = pair ? 2 * pair : k + 1;
    F.pair = pair;
    F.piece_len = q;
    F.piece_groups = F.n_pieces <= 4 ? 1u : F.n_pieces <= 8 ? 2u : 0u;
    if (F.piece_groups) {
      auto row_byte = [&](uint32_t r) { return (plan.row_tab[r >> 2] >> (8 * (r & 3))) & 0xFFu; };
      for (uint32_t g = 0; g < F.piece_groups; ++g) {
        for (uint32_t j = 0; j < 12; ++j) F.piece_tab[g][j] = 0;
        F.piece_last[g] = 0;
        for (uint32_t pp = 0; pp < 4; ++pp) {
          uint32_t piece = 4 * g + pp;
          if (piece >= F.n_pieces) piece = 0;  // a repeated piece changes nothing
          for (uint32_t j = 0; j + 1 < q; ++j) F.piece_tab[g][j] |= row_byte(piece * q + j) << (8 * pp);
          F.piece_last[g] |= row_byte(piece * q + q - 1) << (8 * pp);
        }
      }
    }
    // Dna with <= 8 pieces: the filter works on the two code bit planes (filter_dna_kernel)
    F.piece_planes = fkind == kFilterPlanes ? 1u : 0u;
    F.qgram_table = fkind == kFilterTable || fkind == kFilterCount ? L.d_table.p : nullptr;
    F.count_r = count_r;
    F.count_window = count_w;
    F.count_thresh = count_t;
    F.piece_mirror = 0;
    F.hit_bitmap_rc = rc_bitmap;
    F.count_rc = fkind == kFilterCount && rc_marked ? 1u : 0u;
    if (F.piece_planes) {
      // piece `piece` of the forward pattern, or (mirror) of the Rc strand's pattern with its string
      // reversed: rows q-1 .. 0 of complement(pattern)'s piece, as they read on the forward text
      auto set_piece = [&](ScanParams& X, uint32_t pp, uint32_t piece, bool mirror) {
        uint32_t b0 = 0, b1 = 0;
        for (uint32_t j = 0; j < q; ++j) {
          const uint8_t ch = mirror ? rc_pat[piece * q + (q - 1 - j)] : pat[piece * q + j];
          const uint32_t code = (ch >> 1) & 3u;  // src/profiles/dna.rs:19-40
          b0 |= (code & 1u) << j;
          b1 |= (code >> 1) << j;
        }
        X.piece_bits[pp][0] = b0;
        X.piece_bits[pp][1] = b1;
        X.piece_rem[pp] = plan.m - (piece + 1) * q;
        // (paired filter: an A-type sub-piece is detected q + 2 columns behind its end)
        if (pair && (piece & 1u) == 0) X.piece_rem[pp] = (uint32_t)((int32_t)X.piece_rem[pp] - (int32_t)(q + 2));
        if (mirror) X.piece_mirror |= 1u << pp;
      };
      if (pair) {
        for (uint32_t w = 0; w < 4; ++w) F.pair_y[w] = 0;
        for (uint32_t pp = 0; pp < 2 * pair; ++pp) {
          const uint32_t sib = pp ^ 1u;
          for (uint32_t j = 0; j < q; ++j) {
            // piece pp even (A): its B read forwards; odd (B): its A read backwards
            const uint32_t code = (pat[sib * q + ((pp & 1u) ? q - 1 - j : j)] >> 1) & 3u;
            F.pair_y[2 * (pp >> 2)] |= (code & 1u) << (8 * (pp & 3u) + j);
            F.pair_y[2 * (pp >> 2) + 1] |= (code >> 1) << (8 * (pp & 3u) + j);
          }
        }
      }
      const uint32_t np = k + 1;
      const bool with_rc = rc_bitmap != nullptr && !ext_bitmap && !ext_desc;
      if (with_rc && np <= 4) {  // both strands' pieces in one launch (a repeated piece changes nothing)
        for (uint32_t pp = 0; pp < 4; ++pp) set_piece(F, pp, pp < np ? pp : 0, false);
        for (uint32_t pp = 0; pp < 4; ++pp) set_piece(F, 4 + pp, pp < np ? pp : 0, true);
        F.n_pieces = 8;
        F.piece_groups = 2;
        rc_marked = true;
      } else {
        for (uint32_t pp = 0; pp < 8; ++pp) set_piece(F, pp, pp < F.n_pieces ? pp : 0, false);
        if (with_rc) {  // 5 .. 8 pieces per strand: a second launch for the Rc strand's pieces
          rc_marked = rc_second_pass = true;
        }
      }
    }
    F.stage_blocks = 2u;
    int fwpc = 16;
    // fused: ONE round of workgroups (as many as are resident at once) -- every workgroup ends with the chunk DP of
    // what it found, a phase in which it does not stream; with two rounds the chip goes through that twice (3 GB:
    // 0.595 ms against 0.572 with one round, the same launch without the chunk DP 0.530 / 0.535)
    if (fused) fwpc = 8;
    if (fkind == kFilterTable) {
      // one 4 KiB tile per wave + the table per workgroup decide how many workgroups a CU holds
      F.stage_blocks = 1;
      const uint32_t wg_lds = (1u << (2 * q - 3)) + 4 * 4096u;
      fwpc = 4 * (int)std::min<uint32_t>(8, (160u * 1024u) / wg_lds);
    }
    uint32_t extra_front = 1;
    if (fkind == kFilterCount) {
      const int env_csb = (int)sw.count_stage_blocks;
      F.stage_blocks = env_csb == 1 ? 1u : 2u;  // (whole 128-byte lines per lane and step: read with non-temporal loads)
      const uint32_t per_wave = 4096u * F.stage_blocks + 64u * count_w, table = 1u << (2 * (q + count_r - 1));
      // the table is per workgroup: sixteen waves around one copy where that fits a CU's LDS, else four
      const int env_wpg = (int)sw.count_wpg;
      count_wpg = (env_wpg == 4 || env_wpg == 16) ? (uint32_t)env_wpg : 16u;
      if (table + 16u * per_wave > 160u * 1024u) count_wpg = 4;
      if (count_r == 1 && env_wpg != 16) count_wpg = 4;  // (the R = 1 variants need 157 VGPRs: 1024 threads would spill)
      fwpc = count_wpg == 16 ? 16 : 4 * (int)std::min<uint32_t>(8, (160u * 1024u) / (table + 4 * per_wave));
      extra_front = count_w + 1;
    }
    // (timing level >= 1 records the two events around the filter: that is what the tuner learns from)
    tuned = S->tune && S->timing >= 1 && !ext_bitmap && !ext_desc;
    if (int rc = stream_geometry(F, n_blocks - F.first_owned_block, extra_front, &fgrid, fwpc, tuned ? &S->tuner : nullptr,
                                 sh.d_text, sh.text_len, (uint32_t)fkind * 16u + (rc_marked ? 1u : 0u))) return rc;
    if (fkind == kFilterPlanes) F.stage_blocks = 2;  // (the bit-plane kernel stages whole 128-byte lines only)
    F.lds_per_wave = 4096u * F.stage_blocks + (F.piece_planes ? 0u : 2u * bucket * 512u);
    if (fkind == kFilterCount) {
      F.lds_per_wave = 4096u * F.stage_blocks + 64u * count_w;
      F.waves_per_group = count_wpg;
      fgrid = (uint32_t)((F.n_chunks + 64ull * count_wpg - 1) / (64ull * count_wpg));
    }
    F.fused = 0;
    if (fused) {
      const int env_probe = (int)sw.fused_probe;
      F.fused = 1u | (env_probe == 1 ? 2u : env_probe == 2 ? 6u : 0u);
      F.dp_first_owned = first_owned;
      // chunks per wave between two chunk-DP passes: a wave runs a pass when more than cap - 128 are queued (a full
      // batch of 64 lanes), so the queue never overflows; + the count (16 bytes);
      // 4 workgroups per CU still fit the LDS: 4 x 4 x (8192 + 1536 + 16) = 155 904 bytes
      F.fuse_queue_cap = 192u;
      F.lds_per_wave += F.fuse_queue_cap * 8u + 16u;
      const int env_press = (int)sw.fused_press;
      F.fuse_press = F.fuse_queue_cap - 128u;
      if (env_press > 0 && (uint32_t)env_press < F.fuse_press) F.fuse_press = (uint32_t)env_press;
    }
    {
      // Searches in flight on several lanes: the filter's long-lived workgroups would fill every CU (4 waves
      // per SIMD x 112 VGPRs leave no room for a list / traceback wave), and the previous search's tail
      // kernels would only run in the gaps between filter rounds.  Asking for 56 KB of LDS per workgroup
      // caps the filter at 2 workgroups = 8 waves per CU: two filters in flight still fill the chip, and a
      // tail kernel always finds registers, LDS and wave slots (measured, 3 GB, two searches in flight:
      // 0.63 -> 0.585 ms per search; one search alone: 0.745 -> 0.80 ms, hence only when pipelined).
      const uint32_t pad = pipelined ? 24u * 1024u : 0u;
      if (fkind == kFilterPlanes && pad) F.lds_per_wave += pad / 4u / 16u * 16u;
    }
    // the bit-plane filter as a linear stream (filter_dna_linear_kernel): every wave owns one contiguous
    // range of 128-block steps; SASSY_HIP_FILTER_LINEAR=<waves> sets how many waves the text is cut into
    F.lin_steps = 0;
    const int env_lin = (int)sw.filter_linear;
    if (fkind == kFilterPlanes && env_lin > 0 && !ext_bitmap && !ext_desc) {
      const uint64_t cover = n_blocks - (F.first_owned_block & ~1ull);
      const uint64_t steps = std::max<uint64_t>(1, (cover + 128ull * env_lin - 1) / (128ull * env_lin));
      F.lin_steps = (uint32_t)std::min<uint64_t>(steps, 0x7FFFFFFFu);
      const uint64_t waves = (cover + 128ull * F.lin_steps - 1) / (128ull * F.lin_steps);
      fgrid = (uint32_t)((waves + kWavesPerGroup - 1) / kWavesPerGroup);
    }
    F.hit_bitmap = d_bitmap;
    {
      // room for the expected number of chunks on random text (64 (k+1) / 4^q of the blocks hold a piece
      // end); a denser text overflows into the grow-and-retry path of finish()
      double frac = 64.0 * (k + 1.0) / std::pow(4.0, (double)q);
      if (fkind == kFilterCount) frac = 2.0 * count_tail;
      const size_t expect = (size_t)std::min<double>(1.5 * frac * (double)n_blocks, (double)n_blocks) + 1024;
      if (!ext_desc)
        if (int rc = L.d_desc.reserve(std::max<size_t>(1u << 18, expect))) return rc;
    }
    if (rc_second_pass) {  // same launch, the Rc strand's pieces (all mirrored) instead of the forward ones
      F2 = F;
      F2.piece_mirror = 0;
      for (uint32_t pp = 0; pp < 8; ++pp) {
        const uint32_t piece = pp < k + 1 ? pp : 0;
        uint32_t b0 = 0, b1 = 0;
        for (uint32_t j = 0; j < q; ++j) {
          const uint32_t code = (rc_pat[piece * q + (q - 1 - j)] >> 1) & 3u;
          b0 |= (code & 1u) << j;
          b1 |= (code >> 1) << j;
        }
        F2.piece_bits[pp][0] = b0;
        F2.piece_bits[pp][1] = b1;
        F2.piece_rem[pp] = plan.m - (piece + 1) * q;
        F2.piece_mirror |= 1u << pp;
      }
    }
    // (the chunk DP's carries: in LDS, fewer waves per workgroup for long patterns; beyond ~10 000 rows in global memory and
    // a fixed number of waves -- list_kernel<.., GC>)
    const bool gc = bucket * 512u + (size_t)plan.nwords * 512u > 160 * 1024;
    P.lds_per_wave = bucket * 512u + (gc ? 0u : plan.nwords * 512u);
    P.waves_per_group = (uint32_t)std::min<size_t>(kWavesPerGroup, (160 * 1024) / P.lds_per_wave);
    if (gc) {
      if (int rc = L.d_carry.reserve((size_t)kCarryListGroups * P.waves_per_group * plan.nwords * 128u)) return rc;
      P.carry_global = L.d_carry.p;
    }
  }

  t_mark = t_enter;
  if (int rc = L.reserve_pinned(pin_ops + (size_t)kSpec * (do_trace ? T.str_stride : 0) + 64)) return rc;
  counts[0] = counts[1] = 0;
  timing = S->timing;
  desc_cap = 0;
  return 0;
}

int ScanJob::enqueue(int attempt) {
  P.cand = L.d_cand.p;
  P.cand_cap = (uint32_t)std::min<size_t>(L.d_cand.cap, 0xFFFFFFFFu);
  // (the buffer may be large from an earlier, denser search: the cigar pool of this one holds at most 4 GiB)
  if (do_trace) P.cand_cap = (uint32_t)std::min<uint64_t>(P.cand_cap, 0xFFFFFFFFull / T.str_stride);
  if (int rc = L.d_sorted.reserve(P.cand_cap)) return rc;
  if (do_trace) {
    if (int rc = L.d_trace.reserve(P.cand_cap)) return rc;
    if (int rc = L.d_str.reserve((size_t)P.cand_cap * T.str_stride)) return rc;
    T.cand = Tw.cand = L.d_sorted.p;
    T.cand_cap = Tw.cand_cap = P.cand_cap;
    T.out = Tw.out = L.d_trace.p;
    T.out_str = Tw.out_str = L.d_str.p;
  }
  // control block, rank counters and (first attempt: the filter runs once) the hit bitmap
  // (fused: no bitmap -- and the rank counters behind the control block are not used either)
  // (the rank counters behind the control block only where the rank kernels run: not when the traceback waves rank
  // their reports themselves -- with the counting filter's own chunk list the whole clear is 64 bytes)
  const bool ranks_itself = S->sw.self_rank != 0 && do_trace && use_wave && texts.n == 0;
  const bool bitmap_here = filtered && !ext_bitmap && !ext_desc && !count_direct && attempt == 0;
  const size_t clear_bytes = fused || (ranks_itself && !bitmap_here) ? 64 : kCtlHead + (bitmap_here ? (n_words + 2) * 8 : 0);
  // (the previous search on this lane cleared the 64 bytes behind its last kernel: no memset launch in front of the filter)
  if (!(ctl_pre_cleared && attempt == 0 && clear_bytes == 64)) HIP_TRY(hipMemsetAsync(ctl_base, 0, clear_bytes, L.stream));
  ctl_pre_cleared = false;
  wait_ev_done = false;
  if (ext_wait && attempt == 0) HIP_TRY(hipStreamWaitEvent(L.stream, ext_wait, 0));
  // pipelined sub-shards: this lane's filter starts when the previous sub-shard's filter is done,
  // so that the previous lane's DP / rank / traceback kernels overlap this bandwidth-bound one
  if (wait_for && attempt == 0) HIP_TRY(hipStreamWaitEvent(L.stream, wait_for, 0));
  // (a job that only consumes a bitmap has no filter to time: no events at level 1, each costs ~6 us of stream idle)
  const bool time_head = timing >= 2 || (timing == 1 && !ext_bitmap);
  // (the fused launch carries its events itself: LaunchEvents)
  const bool env_ext_ev = S->sw.ext_events != 0;
  const bool ext_events = time_head && filtered && fused && attempt == 0 && env_ext_ev;
  if (time_head && !ext_events) HIP_TRY(hipEventRecord(L.ev_a, L.stream));
  hipError_t le;
  if (!filtered) {
    le = launch_scan_any(S->profile, P, grid, (size_t)P.waves_per_group * P.lds_per_wave, L.stream);
    if (le != hipSuccess) return hip_fail(le, "scan kernel launch");
  } else {
    if (fused) {  // the filter appends the reports itself: it needs the list (every attempt runs the whole launch)
      F.cand = P.cand;
      F.cand_cap = P.cand_cap;
      F.cand_count = P.cand_count;
      if (int rc = L.d_stash.reserve(std::min<size_t>(P.cand_cap, 1u << 18))) return rc;
      F.stash = L.d_stash.p;
      F.stash_cap = (uint32_t)std::min<size_t>(L.d_stash.cap, 0xFFFFFEu);
      F.counters = nullptr;
      F.row_tab = P.row_tab;
      if (ext_events) g_launch_events = LaunchEvents{L.ev_a, L.ev_f};
      le = launch_filter_any(S->profile, F, fgrid, 1024 + (size_t)kWavesPerGroup * F.lds_per_wave, L.stream);
      g_launch_events = LaunchEvents{};
      if (le != hipSuccess) return hip_fail(le, "fused filter kernel launch");
    } else if ((attempt == 0 || count_direct) && !ext_bitmap && !ext_desc) {  // the hit bitmap does not depend on buffer sizes: build it once
      // (count_direct: the filter files the descriptors itself -- into a list that may have grown: every attempt runs it)
      if (count_direct) {
        F.count_direct = 1;
        uint32_t ml = 16;
        while (ml < 8u * P.wb && ml < 128u) ml <<= 1;
        F.count_maxlen = ml;
        F.dp_first_owned = first_owned;
        const uint32_t n_regions = (uint32_t)((F.n_chunks + 63) / 64);  // a region of the list per wave of the launch
        if (int rc = L.d_regions.reserve((size_t)n_regions * kRegionSlots)) return rc;
        if (int rc = L.d_region_count.reserve(n_regions)) return rc;
        F.desc = L.d_regions.p;
        F.region_count = L.d_region_count.p;
        F.hit_count = nullptr;
      }
      if (rc_marked) HIP_TRY(hipMemsetAsync(rc_bitmap, 0, (n_words + 2) * 8, L.stream));
      if (rc_second_pass) {
        le = launch_filter_any(S->profile, F2, fgrid, 1024 + (size_t)kWavesPerGroup * F2.lds_per_wave, L.stream);
        if (le != hipSuccess) return hip_fail(le, "filter kernel launch (Rc pieces)");
      }
      le = fkind == kFilterCount ? launch_filter_count(F, fgrid, L.stream)
           : fkind == kFilterTable
               ? launch_filter_table(F, fgrid, L.stream)
               : launch_filter_any(S->profile, F, fgrid, 1024 + (size_t)kWavesPerGroup * F.lds_per_wave, L.stream);
      if (le != hipSuccess) return hip_fail(le, "filter kernel launch");
    }
    if (time_head && attempt == 0 && !ext_events) HIP_TRY(hipEventRecord(L.ev_f, L.stream));
    if (signal_filter_done && attempt == 0) HIP_TRY(hipEventRecord(L.ev_filter_done, L.stream));
    if (!fused) {
    maxlen = 16;
    while (maxlen < 8u * P.wb && maxlen < 128u) maxlen <<= 1;
    desc_cap = ext_desc ? ext_ndesc : (uint32_t)std::min<size_t>(L.d_desc.cap, 0x7FFFFFFFu);
    if (int rc = L.d_state.reserve(std::max<uint32_t>(desc_cap, 1))) return rc;
    P.chunk_state = L.d_state.p;
    if (ext_desc)  // the descriptor count the list kernel reads
      if (int rc = L.upload(d_counts + 1, &ext_ndesc, sizeof(uint32_t))) return rc;
    // right dilation: blocks a match END can reach from a piece occurrence; the bit-plane filter
    // marks those blocks itself (it knows the piece), the other filters mark the occurrence's block
    if (count_direct) {  // the regions the filter's waves filled -> the dense list (and its count) the list kernels read
      le = launch_compact_chunks(L.d_regions.p, L.d_region_count.p, (uint32_t)((F.n_chunks + 63) / 64), L.d_desc.p, d_counts + 1, desc_cap,
                                 d_counts + kCtlFuseWord, L.stream);
      if (le != hipSuccess) return hip_fail(le, "chunk list packer launch");
    }
    if (!ext_desc && !count_direct) {
      le = launch_build_chunks(d_bitmap, n_words, n_blocks, first_owned, P.wb, fkind == kFilterPlanes || fkind == kFilterCount ? 0u : P.wb, maxlen, L.d_desc.p,
                               d_counts + 1, desc_cap, d_counters + 2, L.stream);
      if (le != hipSuccess) return hip_fail(le, "chunk builder launch");
    }
    P.desc = ext_desc ? ext_desc : L.d_desc.p;
    P.desc_count = d_counts + 1;
    P.desc_cap = desc_cap;
    // multi-word patterns with few chunks: one lane per pattern word instead of one lane per chunk
    // (up to 8192 waves' worth of chunks; beyond that the lane-per-chunk kernel fills the chip anyway)
    // Round 6: one lane per BLOCK of a chunk (list_rows_kernel: m + blocks - 1 dependent rows per chunk instead of
    // (blocks + words - 1) x 32; switch list_words = 2: the word-pipelined kernel, 0: the lane-per-chunk kernel only).
    P.list_words_max = 0;
    P.list_group_log = 0;
    P.list_rows = 0;
    const int env_words = (int)S->sw.list_words;
    if (env_words && !rows_declined && !ext_desc && plan.nwords >= 2 && plan.nwords <= 64 && !(P.flags & kScanOverhang)) {
      // groups of G lanes: a chunk's warm-up blocks and ten blocks of end positions in one pass (longer chunks take more
      // passes: with G = wb + 6 half the waves of config 3 ran two -- 92 us instead of 46); list_words >= 4: that many lanes
      // (the sizes that leave no lane of the wave over; a run of candidate blocks behind the counting filter is 5 .. 10 blocks)
      uint32_t G = 64;
      for (uint32_t g : {8u, 9u, 10u, 12u, 16u, 21u, 32u, 64u})
        if (g >= P.wb + 10u) { G = g; break; }
      if (env_words >= 4) G = std::min<uint32_t>(64u, (uint32_t)env_words);
      const size_t m16 = ((size_t)plan.m + 15u) & ~(size_t)15;
      const size_t rows_lds = 128 + m16 + 16 + (size_t)kWavesPerGroup * ((size_t)bucket * 512u + (64u / G) * m16);
      if (env_words != 2 && !plan.bytes && !S->want_counters && rows_lds <= 150 * 1024) {
        P.list_rows = 1;
        P.list_group = G;
        P.list_words_max = 8192u * (64u / G);
      } else {
        uint32_t glog = 1;
        while ((1u << glog) < plan.nwords) ++glog;
        P.list_group_log = glog;
        P.list_words_max = (8192u * 64u) >> glog;
      }
    }
    // the descriptor count lives on the device: launch for the capacity, idle waves exit at once
    // (with the few-chunks kernel in front the lane-per-chunk kernel would only start 1 000 workgroups that look at the
    // count and leave -- 4.7 us of stream time: left out; finish_once() sends a longer list through here again)
    uint32_t lgrid = P.list_words_max ? 0u : (desc_cap + 64u * P.waves_per_group - 1) / (64u * P.waves_per_group);
    if (P.carry_global) lgrid = std::min(lgrid, kCarryListGroups);
    le = launch_list_any(S->profile, P, lgrid, (size_t)P.waves_per_group * P.lds_per_wave, L.stream);
    if (le != hipSuccess) return hip_fail(le, "list kernel launch");
    }
  }
  ev_scan = timing >= 2 || (timing == 1 && !filtered);
  if (ev_scan) HIP_TRY(hipEventRecord(L.ev_b, L.stream));
  // reports into result order (by end position) -- the head of the list and the control block
  // straight into the pinned host buffer --, then their traceback
  const uint32_t host_cap = std::min<uint32_t>(kSpec, P.cand_cap);
  // One text, traceback by the wave kernel: that kernel ranks its reports itself (up to kTraceWaveMax of
  // them; finish() falls back to the ranking kernels beyond) -- two launches fewer per search.
  const int env_selfrank = (int)S->sw.self_rank;
  self_rank = env_selfrank != 0 && do_trace && use_wave && texts.n == 0;
  if (!self_rank) {
    le = launch_rank(L.d_cand.p, d_counts, P.cand_cap, reinterpret_cast<uint32_t*>(ctl_base + 64),
                     L.d_sorted.p, reinterpret_cast<Candidate*>(L.h_pin_dev + pin_cands), host_cap,
                     L.h_pin_dev + kPinCounts, texts, L.stream);
    if (le != hipSuccess) return hip_fail(le, "rank kernel launch");
  }
  if (do_trace) {
    T.texts = Tw.texts = texts;
    T.host_out = Tw.host_out = reinterpret_cast<MatchOut*>(L.h_pin_dev + pin_recs);
    T.host_str = Tw.host_str = L.h_pin_dev + pin_ops;
    T.host_cap = Tw.host_cap = host_cap;
    T.unsorted = nullptr;
    Tw.unsorted = self_rank ? L.d_cand.p : nullptr;
    Tw.host_cand = reinterpret_cast<Candidate*>(L.h_pin_dev + pin_cands);
    Tw.host_ctl = reinterpret_cast<uint4*>(L.h_pin_dev + kPinCounts);
    // (the traceback waves tell the host whether any record needs its attention: see finish_once, adoption)
    *reinterpret_cast<volatile uint32_t*>(L.h_pin + kPinFlags) = 0u;
    Tw.host_flags = self_rank ? reinterpret_cast<uint32_t*>(L.h_pin_dev + kPinFlags) : nullptr;
    // (the end position ON a shard border, first_owned * 64: under the report rule it is decided by whoever sees the
    // column behind it -- this shard; a list of ALL end positions <= k has it from the shard on the left already)
    Tw.min_pos = sh.global_offset + first_owned * 64 + (first_owned && all_minima ? 1 : 0);
    T.host_flags = nullptr;
    if (self_rank) Tw.count_max = kTraceWaveMax;
    Tw.dedup = fused ? 1u : 0u;
    Tw.stash = fused ? L.d_stash.p : nullptr;
    Tw.stash_cap = fused ? (uint32_t)std::min<size_t>(L.d_stash.cap, 0xFFFFFEu) : 0u;
    Tw.probe = nullptr;
    const bool env_tprobe = S->sw.trace_probe != 0;
    if (env_tprobe) {  // eight counters per traceback wave
      if (int rc = L.d_probe.reserve((size_t)wave_blocks * 4 * 8)) return rc;
      Tw.probe = L.d_probe.p;
      HIP_TRY(hipMemsetAsync(L.d_probe.p, 0, (size_t)wave_blocks * 4 * 8 * 8, L.stream));
    }
    Tw.rank_lds = 0;
    if (self_rank) {
      // room for the end positions of up to 4096 reports behind the slices, as long as four workgroups still fit a CU
      // (wide bands -- config 3: 38 KB of slices per workgroup -- rank from the list in L2: with half the waves
      // resident the traceback of 2 900 reports took 258 instead of 140 us)
      const size_t used = (size_t)4 * ((plan.m + 15u) & ~15u) + (size_t)4 * Tw.scratch_stride;
      const size_t room = used < 39 * 1024 ? (39 * 1024 - used) / 8 : 0;
      Tw.rank_lds = (uint32_t)std::min<size_t>(4096, room);
    }
    if (use_wave) {
      le = launch_trace(Tw, wave_blocks, L.stream);
      if (le != hipSuccess) return hip_fail(le, "trace kernel launch");
    }
    // the thread-per-report kernel only acts on more than kTraceWaveMax reports: when the wave kernel
    // covers the usual case its launch (an empty kernel otherwise, ~8 us of stream time) is left to
    // finish(), which knows the count
    if (use_thread && !use_wave) {
      le = launch_trace(T, trace_blocks, L.stream);
      if (le != hipSuccess) return hip_fail(le, "trace kernel launch");
    }
    if (timing >= 2) HIP_TRY(hipEventRecord(L.ev_c, L.stream));
  }
  // the lane's other control block, for the next search: cleared behind this search's kernels; the host waits for the event
  // in front of the clear
  {
    const int other = L.ctl_cur ^ 1;
    DevBuf<uint8_t>& O = other ? L.d_ctl_twin : L.d_ctl;
    if (S->sw.ctl_twin != 0 && !L.ctl_clean[other] && O.p != nullptr && O.cap >= 64) {
      HIP_TRY(hipEventRecord(L.ev_done, L.stream));
      HIP_TRY(hipMemsetAsync(O.p, 0, 64, L.stream));
      L.ctl_clean[other] = true;
      wait_ev_done = true;
    }
  }
  return 0;
}

// finish_once() may find that the fused launch could not complete the search (a wave's chunk queue overflowed, a
// report hangs on a chunk seam, the shard's exit state needs the chunk chain): the job then runs again as the
// classic chain, which resolves all of that.
int ScanJob::finish(ScanOut& out) {
  bool redo = false;
  const sassy_hip_Stats before = S->stats;
  if (int rc = finish_once(out, redo)) return rc;
  if (!redo) return 0;
  no_fuse = true;
  L.fuse_backoff = 16;  // and so do the lane's next searches: what sends one search to the classic chain sends the next
  // (only the attempt that produces the result counts: kernel times, bytes and launches of the abandoned one are dropped;
  // the host's waiting stays)
  const double waited = S->stats.host_wait_ms - before.host_wait_ms, queued = S->stats.host_enqueue_ms - before.host_enqueue_ms;
  S->stats = before;
  S->stats.host_wait_ms += waited;
  S->stats.host_enqueue_ms += queued;
  if (int rc = prepare()) return rc;
  if (!empty)
    if (int rc = enqueue(0)) return rc;
  return finish_once(out, redo);
}

int ScanJob::finish_once(ScanOut& out, bool& redo) {
  redo = false;
  out = ScanOut();
  if (empty) return 0;
  bool sorted_on_device = false;
  bool big = false;  // the result's rows and strings lie in the lane's pinned block at these offsets (dense results)
  size_t big_rows_off = 0, big_strs_off = 0, big_cands_off = 0, big_pool_bytes = 0;
  const Candidate* big_list = nullptr;  // ... and the (sorted, deduplicated) reports they belong to on the device
  for (int attempt = 0;; ++attempt) {
    // the only synchronisation of the call; the kernels have written the results into h_pin
    const double t_sync0 = now_ms();
    if (wait_ev_done) HIP_TRY(hipEventSynchronize(L.ev_done));  // (the twin's clear may still run)
    else HIP_TRY(hipStreamSynchronize(L.stream));
    const double t_sync1 = now_ms();
    S->stats.host_enqueue_ms += t_sync0 - t_mark;
    S->stats.host_wait_ms += t_sync1 - t_sync0;
    t_mark = t_sync1;
    memcpy(counts, L.h_pin + kPinCounts, sizeof counts);
    float ms = 0;
    if (ev_scan) {
      HIP_TRY(hipEventElapsedTime(&ms, L.ev_a, L.ev_b));
      S->stats.scan_ms += ms;
      if (!filtered && tuned && attempt == 0) S->tuner_scan.report(P.bpl, ms);
    }
    S->stats.scan_launches += 1;
    if (filtered && attempt == 0 && (timing >= 2 || (timing == 1 && !ext_bitmap))) {
      HIP_TRY(hipEventElapsedTime(&ms, L.ev_a, L.ev_f));
      S->stats.filter_ms += ms;
      if (tuned) S->tuner.report(F.bpl, ms);
    }
    if (do_trace && timing >= 2) {
      HIP_TRY(hipEventElapsedTime(&ms, L.ev_b, L.ev_c));
      S->stats.trace_ms += ms;
    }
    if (S->sw.trace_probe != 0 && do_trace) {
      std::vector<unsigned long long> all((size_t)wave_blocks * 4 * 8);
      HIP_TRY(hipMemcpy(all.data(), L.d_probe.p, all.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t i = 0; i < all.size(); ++i) pr[i & 7] += all[i];
      const double nrep = (double)std::max<unsigned long long>(1, pr[7]);
      fprintf(stderr, "[sassy-hip] trace waves (us per report): rank %.2f window %.2f fill %.2f walk %.2f out %.2f (%llu reports)\n",
              pr[0] / nrep / 100, pr[1] / nrep / 100, pr[2] / nrep / 100, pr[3] / nrep / 100, pr[4] / nrep / 100, pr[7]);
    }
    if (fused || count_direct) {
      uint32_t fw = 0;
      memcpy(&fw, L.h_pin + kPinCounts + 4 * kCtlFuseWord, sizeof fw);
      if (fw != 0) {
        redo = true;
        return 0;
      }
    }
    bool again = false;
    if (filtered && !fused && counts[1] > desc_cap) {  // more chunks than descriptors fit: grow, rebuild
      if (int rc = L.d_desc.reserve((size_t)counts[1] + 1024)) return rc;
      again = true;
    }
    if (filtered && !fused && P.list_words_max && counts[1] > P.list_words_max && counts[1] <= desc_cap) {
      rows_declined = true;  // more chunks than the few-chunks kernel takes: it left them all to the lane-per-chunk kernel
      again = true;
    }
    if (counts[0] > P.cand_cap) {  // more reports than the buffer holds (dense matches)
      if (do_trace && ((uint64_t)counts[0] + 1024) * T.str_stride > 0xFFFFFFFFull)
        return fail(SASSY_HIP_EUNSUPPORTED, "too many reports for one cigar pool (> 4 GiB of cigar text)");
      if (int rc = L.d_cand.reserve((size_t)counts[0] + 1024)) return rc;
      again = true;
    }
    if (!again) {
      if (do_trace && use_wave && counts[0] > kTraceWaveMax && (use_thread || self_rank)) {
        // many reports: what enqueue() left out -- the ranking kernels (self-ranking mode), then the
        // thread-per-report traceback, or the wave kernel on the ranked list where only it applies
        hipError_t le = hipSuccess;
        if (texts.n == 0) {
          // more reports than the traceback waves rank for themselves: radix sort on the device
          // (sort_kernels.hip), then the traceback on the sorted list -- the records arrive in result order,
          // the host sorts nothing (the counting ranker is quadratic: 27 000 reports took it 0.32 ms, the
          // host's std::sort 63 ms for 740 000)
          const size_t need = std::max(sort_scratch_bytes(counts[0]), unique_scratch_bytes(counts[0]));
          if (int rc = L.d_sort.reserve(need)) return rc;
          if (int rc = L.d_flags.reserve(64)) return rc;
          int key_bits = 8;  // (a radix pass per 8 bits of the largest end position)
          while (key_bits < 64 && ((sh.global_offset + sh.text_len + plan.m + 64) >> key_bits) != 0) key_bits += 8;
          le = launch_sort_candidates(L.d_cand.p, L.d_sorted.p, counts[0], L.d_sort.p, L.d_sort.cap, L.stream, 0, key_bits);
          if (le != hipSuccess) return hip_fail(le, "report sort launch");
          sorted_on_device = true;
          // Dense results (10^4 .. 10^6 rows): rows and cigar strings go, with two DMA copies behind the traceback, into
          // ONE pinned block sized for this result; a result that needs no editing keeps it (as the small ones keep the
          // lane's block) -- the host used to move 160 bytes per match through bounce buffers, vectors and loops,
          // 26 ms for 743 000 matches.  What would need editing (a conditional report, a copy, a failed traceback) is
          // found on the device (report_flags_kernel, the traceback kernels) and told in the block's flag word.
          const bool env_nobig = S->sw.big_pin == 0;
          const size_t cnt = counts[0];
          big_rows_off = 256;
          big_strs_off = (big_rows_off + cnt * sizeof(MatchOut) + 255) / 256 * 256;
          big_cands_off = (big_strs_off + cnt * T.str_stride + 255) / 256 * 256;
          const size_t big_bytes = big_cands_off + cnt * sizeof(Candidate) + 256;
          if (!env_nobig && do_trace) {
            unsigned char ctl_save[128];
            memcpy(ctl_save, L.h_pin, sizeof ctl_save);
            if (L.reserve_pinned(big_bytes) == 0) {
              memcpy(L.h_pin, ctl_save, sizeof ctl_save);  // (the block may be another one now)
              big = true;
            } else {
              (void)hipGetLastError();
              if (L.reserve_pinned(pin_ops + (size_t)kSpec * T.str_stride + 64)) return fail(SASSY_HIP_ENOMEM, "no pinned memory");
              memcpy(L.h_pin, ctl_save, sizeof ctl_save);
            }
          }
        } else if (self_rank) {
          le = launch_rank(L.d_cand.p, d_counts, P.cand_cap, reinterpret_cast<uint32_t*>(ctl_base + 64), L.d_sorted.p,
                           reinterpret_cast<Candidate*>(L.h_pin_dev + pin_cands), std::min<uint32_t>(kSpec, P.cand_cap),
                           L.h_pin_dev + kPinCounts, texts, L.stream);
          if (le != hipSuccess) return hip_fail(le, "rank kernel launch");
        }
        TraceParams Tall = use_thread ? T : Tw;
        if (!use_thread) {
          Tall.unsorted = nullptr;
          Tall.count_max = 0xFFFFFFFFu;
        }
        // (this launch traces whatever the list holds by now: the dedup below may leave fewer than kTraceWaveMax reports,
        // the count the thread kernel otherwise leaves to the wave kernel -- it returned at once, and the rows of an
        // earlier search went out: fuzz, search_all over N runs, 7 360 reports out of > 8 192 with copies)
        Tall.count_min = 0;
        if (big) {
          big_list = L.d_sorted.p;
          if (fused) {
            // the fused filter's overlapping windows report some positions twice, windows that begin in the halo report
            // the previous shard's: the list loses them here (the count in the control block follows), not on the host
            le = launch_unique_reports(L.d_sorted.p, counts[0], Tw.min_pos, L.d_cand.p, d_counts, L.d_sort.p, L.d_sort.cap, L.stream);
            if (le != hipSuccess) return hip_fail(le, "report dedup launch");
            big_list = L.d_cand.p;
          }
          *reinterpret_cast<volatile uint32_t*>(L.h_pin + kPinFlags) = 0u;
          Tall.cand = big_list;
          Tall.host_flags = reinterpret_cast<uint32_t*>(L.h_pin_dev + kPinFlags);  // (failed tracebacks: rare, straight to the host)
          Tall.host_cap = 0;  // (no second copy of the head of the list: everything travels by DMA)
          HIP_TRY(hipMemsetAsync(L.d_flags.p, 0, 4, L.stream));
          le = launch_report_flags(big_list, counts[0], d_counts, 0, L.d_flags.p, L.stream);
          if (le != hipSuccess) return hip_fail(le, "report flags launch");
        } else if (sorted_on_device) {
          Tall.host_flags = nullptr;
        }
        le = launch_trace(Tall, use_thread ? trace_blocks : wave_blocks, L.stream);
        if (le != hipSuccess) return hip_fail(le, "trace kernel launch");
        if (big) {
          // the strings without their slots' padding (SASSY_HIP_COMPACT_CIGARS=0: the slots as they are)
          const bool env_nocompact = S->sw.compact_cigars == 0;
          const char* d_pool = reinterpret_cast<const char*>(L.d_str.p);
          big_pool_bytes = (size_t)counts[0] * T.str_stride;
          if (!env_nocompact) {
            if (int rc = L.d_scratch2.reserve(compact_scratch_bytes(counts[0], T.str_stride))) return rc;
            HIP_TRY(hipMemsetAsync(L.d_flags.p + 1, 0, 4, L.stream));
            le = launch_compact_cigars(L.d_trace.p, reinterpret_cast<const char*>(L.d_str.p), counts[0], d_counts, T.str_stride, L.d_flags.p + 1,
                                       L.d_scratch2.p, L.d_scratch2.cap, &d_pool, L.stream);
            if (le != hipSuccess) return hip_fail(le, "cigar compaction launch");
            uint32_t total = 0;
            HIP_TRY(hipMemcpyAsync(&total, L.d_flags.p + 1, 4, hipMemcpyDeviceToHost, L.stream));
            HIP_TRY(hipStreamSynchronize(L.stream));
            big_pool_bytes = total;
          }
          HIP_TRY(hipMemcpyAsync(L.h_pin + kPinFlags2, L.d_flags.p, 4, hipMemcpyDeviceToHost, L.stream));
          HIP_TRY(hipMemcpyAsync(L.h_pin + kPinCount2, d_counts, 4, hipMemcpyDeviceToHost, L.stream));
          HIP_TRY(hipMemcpyAsync(L.h_pin + big_rows_off, L.d_trace.p, (size_t)counts[0] * sizeof(MatchOut), hipMemcpyDeviceToHost, L.stream));
          if (big_pool_bytes) HIP_TRY(hipMemcpyAsync(L.h_pin + big_strs_off, d_pool, big_pool_bytes, hipMemcpyDeviceToHost, L.stream));
          if (!sh.adopt_ok)  // (a caller that edits the list wants the reports themselves as well)
            HIP_TRY(hipMemcpyAsync(L.h_pin + big_cands_off, big_list, (size_t)counts[0] * sizeof(Candidate), hipMemcpyDeviceToHost, L.stream));
        }
        HIP_TRY(hipStreamSynchronize(L.stream));
        if (big) {
          uint32_t c2 = 0;
          memcpy(&c2, L.h_pin + kPinCount2, sizeof c2);
          if (c2 > counts[0]) return fail(SASSY_HIP_EINVAL, "internal: report count grew in the dedup");
          counts[0] = c2;
        }
      }
      break;
    }
    if (attempt == 3) return fail(SASSY_HIP_ENOMEM, "candidate / descriptor buffer overflow");
    if (int rc = enqueue(attempt + 1)) return rc;
  }
  const uint32_t count = counts[0];
  const uint32_t n_desc = filtered ? counts[1] : 0;
  S->stats.chunks += filtered ? n_desc : P.n_chunks;
  S->stats.blocks_per_chunk = filtered ? F.bpl : P.bpl;
  S->stats.warmup_blocks = P.wb;
  S->stats.grid = filtered ? fgrid : grid;
  S->stats.text_bytes += sh.text_len - sh.halo_len;
  S->stats.filtered = filtered ? (uint32_t)fkind : 0u;
  S->stats.piece_len = q;
  S->stats.fused = fused ? 1u : 0u;
  S->stats.pair = fused ? pair : 0u;
  {
    unsigned long long c[4];
    memcpy(c, L.h_pin + kPinCounters, sizeof c);
    S->stats.word_rows += c[0];
    S->stats.blocks += c[1];
    S->stats.hit_blocks += c[2];
    S->stats.live_blocks += c[3];
  }

  // Nothing for the host to edit -- every report ranked and traced by the traceback waves, no duplicate, no
  // conditional report, no failed traceback (the waves would have said so in the flag word) -- and a caller that takes
  // the records as they are: the result keeps the pinned block, the lane gets another one.
  uint32_t host_flags = 0;
  memcpy(&host_flags, L.h_pin + kPinFlags, sizeof host_flags);
  if (big) {
    uint32_t f2 = 0;
    memcpy(&f2, L.h_pin + kPinFlags2, sizeof f2);
    host_flags |= f2;
  }
  const bool env_noadopt = S->sw.adopt == 0;
  bool adopt = sh.adopt_ok && !env_noadopt && do_trace && self_rank && !sorted_on_device && count != 0 && count <= kSpec &&
               count <= kTraceWaveMax && texts.n == 0 && host_flags == 0;
  if (big) adopt = sh.adopt_ok && !env_noadopt && host_flags == 0 && count != 0;
  if (adopt) adopt = g_pin_pool.may_adopt(L.h_pin_cap);
  struct AdoptSlot {  // the counted slot goes back unless the block really changes hands at the end of this function
    bool held;
    size_t bytes;
    ~AdoptSlot() { if (held) g_pin_pool.adopted_back(bytes); }
  } adopt_slot{adopt, L.h_pin_cap};
  if (adopt) {
    out.ext_matches = reinterpret_cast<const sassy_hip_Match*>(L.h_pin + (big ? big_rows_off : pin_recs));
    out.ext_n = count;
    out.ext_pool = reinterpret_cast<const char*>(L.h_pin + (big ? big_strs_off : pin_ops));
    out.ext_pool_len = big ? big_pool_bytes : (size_t)count * T.str_stride;
  } else if (big) {
    // (host -> host copies out of the pinned block; the reports themselves came along unless the caller was expected to adopt)
    out.cands.resize(count);
    if (!sh.adopt_ok) memcpy(out.cands.data(), L.h_pin + big_cands_off, (size_t)count * sizeof(Candidate));
    else if (int rc = L.download(out.cands.data(), big_list, (size_t)count * sizeof(Candidate))) return rc;
    const sassy_hip_Match* hm = reinterpret_cast<const sassy_hip_Match*>(L.h_pin + big_rows_off);
    out.matches.assign(hm, hm + count);
    out.pool.assign(reinterpret_cast<const char*>(L.h_pin + big_strs_off), big_pool_bytes);
  } else if (count) {
    // (after a device sort the staging area's head holds the unsorted list's records: take everything from the device)
    const uint32_t have = sorted_on_device ? 0u : std::min<uint32_t>(count, kSpec);
    // (assign, not resize + memcpy: one pass over the memory instead of a zero fill and a copy)
    const Candidate* hc = reinterpret_cast<const Candidate*>(L.h_pin + pin_cands);
    out.cands.assign(hc, hc + have);
    out.cands.resize(count);
    if (count > have)
      if (int rc = L.download(out.cands.data() + have, L.d_sorted.p + have, (size_t)(count - have) * sizeof(Candidate))) return rc;
    if (do_trace) {
      const sassy_hip_Match* hm = reinterpret_cast<const sassy_hip_Match*>(L.h_pin + pin_recs);
      out.matches.assign(hm, hm + have);
      out.matches.resize(count);
      out.pool.assign(reinterpret_cast<const char*>(L.h_pin + pin_ops), (size_t)have * T.str_stride);
      out.pool.resize((size_t)count * T.str_stride);
      if (count > have) {
        if (int rc = L.download(out.matches.data() + have, L.d_trace.p + have, (size_t)(count - have) * sizeof(MatchOut))) return rc;
        if (int rc = L.download(&out.pool[0] + (size_t)have * T.str_stride, L.d_str.p + (size_t)have * T.str_stride,
                                (size_t)(count - have) * T.str_stride)) return rc;
      }
    }
  }
  if (texts.n) {  // multi-text buffer, search_all: reports that lie in a separator are no reports
    size_t w = 0;
    for (size_t i = 0; i < out.cands.size(); ++i) {
      if (out.cands[i].flags & kCandDrop) continue;
      out.cands[w] = out.cands[i];
      if (do_trace) out.matches[w] = out.matches[i];
      ++w;
    }
    out.cands.resize(w);
    if (do_trace) out.matches.resize(w);
  }
  // fused launch: window chunks that begin in the halo also report end positions in front of the first owned block
  // (the previous shard's)
  const uint64_t fused_min_pos = sh.global_offset + first_owned * 64 + (first_owned && all_minima ? 1 : 0);
  if (fused && !sorted_on_device) {  // a report two chunks made: the second copy came back as a kCandDrop record
    size_t w = 0;
    for (size_t i = 0; i < out.cands.size(); ++i) {
      if ((out.cands[i].flags & kCandDrop) || out.cands[i].pos < fused_min_pos) continue;
      if (w != i) {
        out.cands[w] = out.cands[i];
        if (do_trace) out.matches[w] = out.matches[i];
      }
      ++w;
    }
    out.cands.resize(w);
    if (do_trace) out.matches.resize(w);
  } else if (fused) {  // sorted on the device (more reports than the traceback waves rank): copies are neighbours
    size_t w = 0;
    for (size_t i = 0; i < out.cands.size(); ++i) {
      if (out.cands[i].pos < fused_min_pos) continue;
      if (w > 0 && out.cands[i].pos == out.cands[w - 1].pos) {
        // (a copy without kCandCond saw what settles the plateau state: the report is certain whatever the others say)
        if (!(out.cands[i].flags & kCandCond)) out.cands[w - 1].flags &= ~kCandCond;
        continue;
      }
      if (w != i) {
        out.cands[w] = out.cands[i];
        if (do_trace) out.matches[w] = out.matches[i];
      }
      ++w;
    }
    out.cands.resize(w);
    if (do_trace) out.matches.resize(w);
  }
  if (do_trace)
    for (const sassy_hip_Match& r : out.matches)
      if (r.pad_[0] == kTraceFailed)
        // the reference asserts both conditions (src/search.rs:1672-1685) and panics in get_trace
        return fail(SASSY_HIP_EINVAL, "traceback failed for a reported end position (internal error)");
  if (count > kRankLimit && !sorted_on_device) {
    // too many reports for the device ranking pass (and not the single-text traceback path, which sorts on the
    // device): they arrived in append order, sort here
    std::vector<uint32_t> order(count);
    for (uint32_t i = 0; i < count; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return out.cands[x].pos < out.cands[y].pos; });
    std::vector<Candidate> sc(count);
    for (uint32_t i = 0; i < count; ++i) sc[i] = out.cands[order[i]];
    out.cands.swap(sc);
    if (do_trace) {
      std::vector<sassy_hip_Match> sm(count);
      for (uint32_t i = 0; i < count; ++i) sm[i] = out.matches[order[i]];
      out.matches.swap(sm);
    }
  }
  S->stats.candidates += count;

  // ---- seams: reports that depend on how a plateau was entered left of their chunk ----
  bool any_cond = false;
  for (const Candidate& c : out.cands) any_cond |= (c.flags & kCandCond) != 0;
  if (fused && any_cond) {  // the chunk chain that resolves it exists only in the classic path
    redo = true;
    return 0;
  }
  bool need_state = any_cond || !sh.text_end;  // non-final shards publish their exit state
  if (need_state && !any_cond) {
    // common case: no report hangs on a chunk seam, only the exit state is wanted, and the chunk that
    // ends the buffer has published it in the control block (no chunk there = the last block is > k)
    uint32_t tail[4];
    memcpy(tail, L.h_pin + kPinCounts + 4 * kCtlTailWord, sizeof tail);
    if (!tail[3]) { out.exit_state = kStateDecTrue; need_state = false; }
    else if (tail[1] != kStatePass) { out.exit_state = (int)tail[1]; need_state = false; }
    else if (tail[2] & kDescClearBefore) { out.exit_state = kStateDecTrue; need_state = false; }
    // else: one plateau from the chunk's start to the buffer end -- walk the chain below
    if (need_state && fused) {
      redo = true;
      return 0;
    }
  }
  // chunk table in text order: [own_lo, own_hi), exit state, "its left edge is known to be > k"
  struct ChunkInfo { uint64_t lo, hi; uint8_t state; bool clear_before; };
  std::vector<ChunkInfo> chunks;
  if (need_state) {
    if (!filtered) {
      std::vector<uint8_t> state(P.n_chunks);
      HIP_TRY(hipMemcpy(state.data(), L.d_state.p, P.n_chunks, hipMemcpyDeviceToHost));
      chunks.resize(P.n_chunks);
      for (uint64_t c = 0; c < P.n_chunks; ++c) {
        const uint64_t lo = first_owned + c * P.bpl;
        chunks[c] = ChunkInfo{lo, std::min<uint64_t>(lo + P.bpl, n_blocks), state[c], false};
      }
    } else if (n_desc) {
      std::vector<ChunkDesc> desc(n_desc);
      std::vector<uint8_t> state(n_desc);
      HIP_TRY(hipMemcpy(desc.data(), L.d_desc.p, (size_t)n_desc * sizeof(ChunkDesc), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(state.data(), L.d_state.p, n_desc, hipMemcpyDeviceToHost));
      chunks.resize(n_desc);
      for (uint32_t c = 0; c < n_desc; ++c)
        chunks[c] = ChunkInfo{desc[c].own_lo, desc[c].own_hi, state[c], (desc[c].flags & kDescClearBefore) != 0};
      std::sort(chunks.begin(), chunks.end(), [](const ChunkInfo& a, const ChunkInfo& b) { return a.lo < b.lo; });
    }
  }
  // decreasing-state arriving at the left edge of chunk ci: kStateDecTrue/False, or kStatePass
  // when only the previous shard knows
  auto incoming = [&](size_t ci) -> int {
    for (;;) {
      const ChunkInfo& c = chunks[ci];
      if (c.clear_before) return kStateDecTrue;
      if (ci == 0 || chunks[ci - 1].hi != c.lo) {
        // nothing of this buffer lies directly left of it: text start, a skipped (all > k)
        // block, or the previous shard
        if (c.lo == first_owned && !(sh.text_start && first_owned == 0) && first_owned > 0) return kStatePass;
        if (c.lo == 0 && !sh.text_start) return kStatePass;
        return kStateDecTrue;
      }
      if (chunks[ci - 1].state != kStatePass) return chunks[ci - 1].state;
      --ci;
    }
  };
  if (any_cond) {
    std::vector<Candidate> kept;
    std::vector<sassy_hip_Match> kept_m;
    kept.reserve(out.cands.size());
    if (do_trace) kept_m.reserve(out.cands.size());
    auto keep = [&](const Candidate& c, size_t ci) {
      kept.push_back(c);
      if (do_trace) kept_m.push_back(out.matches[ci]);
    };
    const uint64_t end_global = sh.global_offset + sh.text_len;
    for (size_t ci = 0; ci < out.cands.size(); ++ci) {
      const Candidate& c = out.cands[ci];
      if (!(c.flags & kCandCond)) { keep(c, ci); continue; }
      out.cond_seen++;
      uint64_t blk = (c.pos - sh.global_offset) / 64;
      if (c.pos == end_global && blk >= n_blocks) blk = n_blocks - 1;  // end-of-text report
      // the chunk that owns the block in which this report was decided
      size_t lo = 0, hi = chunks.size();
      while (lo + 1 < hi) {
        const size_t mid = (lo + hi) / 2;
        if (chunks[mid].lo <= blk) lo = mid; else hi = mid;
      }
      const int inc = chunks.empty() ? kStateDecTrue : incoming(lo);
      Candidate cc = c;
      if (inc == kStateDecTrue) { cc.flags &= ~kCandCond; keep(cc, ci); }
      else if (inc == kStatePass) {
        out.conditional_index = (int64_t)kept.size();  // only the previous shard knows
        keep(c, ci);
      }  // kStateDecFalse: the plateau was entered by an increase -> not a report
    }
    out.cands.swap(kept);
    if (do_trace) out.matches.swap(kept_m);
  }
  if (need_state) {
    // exit state = decreasing-state after the last owned block
    out.exit_state = kStateDecTrue;
    if (!chunks.empty() && chunks.back().hi == n_blocks) {
      size_t ci = chunks.size() - 1;
      if (chunks[ci].state != kStatePass) out.exit_state = chunks[ci].state;
      else out.exit_state = incoming(ci);
    }
  }
  S->stats.cond_resolved += out.cond_seen;
  if (adopt) {  // (the lane reserves another block in its next prepare())
    out.pin = L.take_pin();
    adopt_slot.held = false;  // the slot now belongs to the block's owner (ScanOut, then the result)
  }
  return 0;
}

// One buffer on the searcher's first lane: prepare, queue, wait.
static int run_scan_single(sassy_SearcherType* S, const ShardView& sh, const PatternPlan& plan, uint32_t k,
                           bool all_minima, const uint8_t* pat, bool do_trace, uint64_t total_len, ScanOut& out,
                           const TextTable& texts = TextTable{}) {
  ScanJob job(S, S->lanes[0], sh, plan, k, all_minima, pat, do_trace, total_len);
  job.texts = texts;
  job.texts.all_minima = all_minima ? 1u : 0u;
  if (int rc = job.prepare()) return rc;
  if (!job.empty)
    if (int rc = job.enqueue(0)) return rc;
  return job.finish(out);
}

uint64_t required_halo_bytes(size_t pattern_len, size_t k) {
  // warm-up blocks of the scan + the traceback window, whole 128-byte lines
  const uint64_t wb = warmup_blocks((uint32_t)pattern_len, (uint32_t)k);
  uint64_t h = std::max<uint64_t>(64 * (wb + 4), pattern_len + k);
  return (h + 127) / 128 * 128;
}

// Optional (SASSY_HIP_LANES=2..4; default 1 = off): long buffers are cut into sub-shards, one per
// lane.  The prefilter of sub-shard j+1 waits for the prefilter of sub-shard j (they would only
// share the HBM bandwidth), so that the chunk list / DP / rank / traceback kernels of sub-shard j --
// short, latency-bound, few waves -- could run underneath the next prefilter instead of after it.
// The sub-shards are exact shards (halo to the left, seam protocol of DESIGN.md 5.1); their results
// are concatenated with the plateau state handed from one to the next (parity-tested with
// SASSY_HIP_SUBSHARD_MIN=2048).  Measured on MI355X (config 2): 0.81 ms with 1 lane, 0.97 / 1.00 /
// 1.22 ms with 2 / 3 / 4 lanes -- the prefilter's long-lived workgroups fill every CU, so the
// other queue's small kernels do not get scheduled underneath it and the extra launches only add
// time.  Hence off by default; the lanes stay as the unit a future scheduler can build on.
int run_scan(sassy_SearcherType* S, const ShardView& sh, const PatternPlan& plan, uint32_t k,
                    bool all_minima, const uint8_t* pat, bool do_trace, uint64_t total_len, ScanOut& out) {
  const int env_lanes = (int)S->sw.lanes;
  const uint64_t min_sub = (uint64_t)std::max<long>(1, S->sw.subshard_min);
  const uint64_t halo = required_halo_bytes(plan.m, k);
  const uint64_t own0 = sh.halo_len;
  const uint64_t owned_bytes = sh.text_len > own0 ? sh.text_len - own0 : 0;
  uint64_t nl = std::min<uint64_t>(std::min<int>(env_lanes, kMaxLanes), owned_bytes / std::max<uint64_t>(min_sub, 2 * halo + 64));
  if (nl < 2 || filter_piece_len(plan, k, S) == 0)
    return run_scan_single(S, sh, plan, k, all_minima, pat, do_trace, total_len, out);

  const uint64_t owned_blocks = (owned_bytes + 63) / 64;
  const uint64_t per = (owned_blocks + nl - 1) / nl;  // blocks per sub-shard
  std::vector<std::unique_ptr<ScanJob>> jobs;
  for (uint64_t j = 0; j < nl; ++j) {
    const uint64_t a = own0 + j * per * 64;
    if (a >= sh.text_len) break;
    const uint64_t b = std::min<uint64_t>(sh.text_len, a + per * 64);
    const uint64_t h = j == 0 ? own0 : halo;
    ShardView sub{sh.d_text + (a - h), h + (b - a), h, sh.global_offset + (a - h), j == 0 && sh.text_start,
                  b == sh.text_len && sh.text_end};
    jobs.emplace_back(new ScanJob(S, S->lanes[j], sub, plan, k, all_minima, pat, do_trace, total_len));
  }
  const size_t n = jobs.size();
  for (size_t j = 0; j < n; ++j) {
    ScanJob& job = *jobs[j];
    if (j == 1) {
      // the uploads prepare() of sub-shard 0 queued on the searcher's stream must be done before any
      // other lane reads the pattern tables (and the text, if this call uploaded it)
      HIP_TRY(hipEventRecord(S->ev_inputs, S->stream));
    }
    if (j >= 1) HIP_TRY(hipStreamWaitEvent(S->lanes[j].stream, S->ev_inputs, 0));
    if (int rc = job.prepare()) return rc;
    job.wait_for = j >= 1 ? S->lanes[j - 1].ev_filter_done : nullptr;
    job.signal_filter_done = j + 1 < n;
    if (!job.empty)
      if (int rc = job.enqueue(0)) return rc;
  }
  std::vector<ScanOut> outs(n);
  int first_rc = 0;
  for (size_t j = 0; j < n; ++j) {  // always wait for every lane, also after an error
    const int rc = jobs[j]->finish(outs[j]);
    if (rc && !first_rc) first_rc = rc;
  }
  if (first_rc) {
    for (size_t j = 0; j < n; ++j) (void)hipStreamSynchronize(S->lanes[j].stream);
    return first_rc;
  }

  // ---- concatenate, handing the plateau state across the sub-shard seams ----
  out = ScanOut();
  size_t total = 0;
  for (const ScanOut& o : outs) total += o.cands.size();
  out.cands.reserve(total);
  if (do_trace) out.matches.reserve(total);
  int incoming = sh.text_start ? kStateDecTrue : kStatePass;  // decreasing-state arriving at sub-shard j
  for (size_t j = 0; j < n; ++j) {
    ScanOut& o = outs[j];
    const size_t base = out.pool.size();
    if (base + o.pool.size() > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
    out.pool.append(o.pool);
    for (size_t i = 0; i < o.cands.size(); ++i) {
      Candidate c = o.cands[i];
      if ((int64_t)i == o.conditional_index) {
        if (incoming == kStateDecFalse) continue;  // its plateau was entered by an increase: not a report
        if (incoming == kStateDecTrue) c.flags &= ~kCandCond;
        else {
          if (out.conditional_index >= 0)  // two reports that depend on the previous shard: give up pipelining
            return run_scan_single(S, sh, plan, k, all_minima, pat, do_trace, total_len, out);
          out.conditional_index = (int64_t)out.cands.size();
        }
      }
      out.cands.push_back(c);
      if (do_trace) {
        sassy_hip_Match r = o.matches[i];
        r.cigar_off = (uint32_t)(r.cigar_off + base);
        out.matches.push_back(r);
      }
    }
    out.cond_seen += o.cond_seen;
    if (o.exit_state != kStatePass) incoming = o.exit_state;
  }
  out.exit_state = incoming;
  return 0;
}

// The reference's lane reports (opt-in: sassy_hip_set_reference_lanes / SASSY_HIP_REF_LANES = 4 | 8).
// The reference cuts a single text into LANES chunks (4 with AVX2, 8 with AVX-512), lane l walking the blocks
// [l bpc, l bpc + bpc + overlap) with a FRESH start -- D[j][start] = j and decreasing = true
// (src/search.rs:1016-1056) -- and keeps of lane l the reports with lane_end[l-1] <= end < lane_end[l]
// (:1202-1240).  On low-complexity text that yields reports the definition (one left-to-right pass, the
// default here) does not have: a <=k plateau entered by an INCREASE left of a lane's start looks entered by a
// decrease to that lane.  This mode reproduces the reference binary's output for a given
// SIMD width: every lane is searched as a text of its own that starts at the lane's first block (text-start
// semantics: exactly the fresh start), without the end-of-text rule unless the lane reaches the end of the
// text (the lane's walk simply stops), and its reports are cut to the lane's range.  Values <= k at or behind
// lane_end[l-1] >= start + m + k are exact, so only the plateau bookkeeping differs -- as in the reference.
// (The reference may also stop its overlap blocks early, should_terminate_early :1253-1271, which moves
// lane_end; it does so only where no lane can still report, so the reports are the same.)
// Checked against the reference-shaped port oracle/sassy_refstyle.c on periodic fixtures (tests).
int run_scan_ref_lanes(sassy_SearcherType* S, const uint8_t* d_text, uint64_t n, const PatternPlan& plan, uint32_t k,
                              bool all_minima, const uint8_t* pat, bool do_trace, uint32_t lanes, ScanOut& out) {
  out = ScanOut();
  const uint64_t overlap = ((uint64_t)plan.m + k + 63) / 64;
  const uint64_t nblocks = (n + 63) / 64;
  const uint64_t rest = nblocks > overlap ? nblocks - overlap : 0;
  const uint64_t bpc = (rest + lanes - 1) / lanes;
  for (uint32_t l = 0; l < lanes; ++l) {
    const uint64_t a = (uint64_t)l * bpc * 64;
    if (a >= n) break;
    const uint64_t b = std::min<uint64_t>(n, ((uint64_t)l * bpc + bpc + overlap) * 64);
    const uint64_t lo = l == 0 ? 0 : (((uint64_t)(l - 1)) * bpc + bpc + overlap) * 64;
    const uint64_t hi = l + 1 == lanes ? UINT64_MAX : ((uint64_t)l * bpc + bpc + overlap) * 64;
    ShardView sub{d_text + a, b - a, 0, a, true, b == n};
    ScanOut so;
    if (int rc = run_scan(S, sub, plan, k, all_minima, pat, do_trace, n, so)) return rc;
    const size_t base = out.pool.size();
    if (base + so.pool.size() > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
    out.pool.append(so.pool);
    for (size_t i = 0; i < so.cands.size(); ++i) {
      const uint64_t e = so.cands[i].pos;
      if (e < lo || e >= hi) continue;
      out.cands.push_back(so.cands[i]);
      if (do_trace) {
        sassy_hip_Match r = so.matches[i];
        r.cigar_off = (uint32_t)(r.cigar_off + base);
        out.matches.push_back(r);
      }
    }
  }
  return 0;
}

// Copies texts[i] (lens[i] bytes) to dst + start[i] and fills the gap up to the next text's start (or
// `total`) with `pad`; several threads when there is enough to copy (a 100 MB read set: 27 -> 3 ms).
void layout_texts(uint8_t* dst, const uint8_t* const* texts, const size_t* lens, const uint64_t* start, size_t nt,
                         uint64_t total, uint8_t pad) {
  auto work = [&](size_t a, size_t b) {
    for (size_t i = a; i < b; ++i) {
      if (lens[i]) memcpy(dst + start[i], texts[i], lens[i]);
      const uint64_t end = i + 1 < nt ? start[i + 1] : total;
      const uint64_t from = start[i] + lens[i];
      if (end > from) memset(dst + from, pad, end - from);
    }
  };
  const size_t nthreads = (size_t)std::min<uint64_t>(16, std::min<uint64_t>(total >> 22, nt));
  if (nthreads < 2) { work(0, nt); return; }
  std::vector<std::thread> pool;
  const size_t per = (nt + nthreads - 1) / nthreads;
  for (size_t t = 0; t < nthreads; ++t) {
    const size_t a = t * per, b = std::min(nt, a + per);
    if (a < b) pool.emplace_back(work, a, b);
  }
  for (std::thread& th : pool) th.join();
}

// layout_texts with the upload riding along: the batch is cut into segments of about 32 MB; the threads lay the
// segments out one after the other (each thread a share of every segment), and as soon as a segment is complete the
// calling thread queues its host -> device copy -- the PCIe transfer of segment i runs while segment i + 1 is laid
// out (330 MB of reads: 3.7 ms of layout + 6 ms of upload -> 6.5 ms).
int layout_and_upload(uint8_t* dst, uint8_t* d_dst, const uint8_t* const* texts, const size_t* lens,
                             const uint64_t* start, size_t nt, uint64_t total, uint8_t pad, hipStream_t stream) {
  const size_t nthreads = (size_t)std::min<uint64_t>(16, std::min<uint64_t>(total >> 22, nt));
  if (nthreads < 2 || total < (64u << 20)) {
    layout_texts(dst, texts, lens, start, nt, total, pad);
    HIP_TRY(hipMemcpyAsync(d_dst, dst, total, hipMemcpyHostToDevice, stream));
    return 0;
  }
  // segment boundaries (text indices): about 32 MB each
  std::vector<size_t> seg{0};
  for (size_t i = 1; i < nt; ++i)
    if (start[i] - start[seg.back()] >= (32u << 20)) seg.push_back(i);
  seg.push_back(nt);
  const size_t ns = seg.size() - 1;
  std::vector<std::atomic<uint32_t>> done(ns);
  for (auto& d : done) d.store(0, std::memory_order_relaxed);
  auto work = [&](size_t t) {
    for (size_t sg = 0; sg < ns; ++sg) {
      const size_t a0 = seg[sg], n = seg[sg + 1] - a0, per = (n + nthreads - 1) / nthreads;
      const size_t a = a0 + std::min(n, t * per), b = a0 + std::min(n, (t + 1) * per);
      for (size_t i = a; i < b; ++i) {
        if (lens[i]) memcpy(dst + start[i], texts[i], lens[i]);
        const uint64_t end = i + 1 < nt ? start[i + 1] : total;
        const uint64_t from = start[i] + lens[i];
        if (end > from) memset(dst + from, pad, end - from);
      }
      done[sg].fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(work, t);
  hipError_t err = hipSuccess;
  for (size_t sg = 0; sg < ns; ++sg) {
    while (done[sg].load(std::memory_order_acquire) < nthreads) std::this_thread::yield();
    const uint64_t from = sg ? start[seg[sg]] : 0, to = sg + 1 < ns ? start[seg[sg + 1]] : total;
    if (err == hipSuccess && to > from) err = hipMemcpyAsync(d_dst + from, dst + from, to - from, hipMemcpyHostToDevice, stream);
  }
  for (std::thread& th : pool) th.join();
  return err == hipSuccess ? 0 : hip_fail(err, "hipMemcpyAsync");
}

// Append the matches of one scan to a result: the device already produced finished records and
// cigar text (trace_kernel.hip); only the pool offsets are rebased.  Returns the index of the
// first appended match.
int append_matches(ScanOut& so, uint64_t total_len, const PatternPlan& plan, bool without_trace,
                          uint64_t pattern_idx, sassy_hip_Result* R, size_t& first, const HostTexts* ht) {
  first = R->matches.size();
  if (without_trace) {  // reference: src/search.rs:1464-1475
    for (const Candidate& c : so.cands) {
      sassy_hip_Match r{};
      uint64_t ts, te, ti;
      text_bounds(ht, c, total_len, ts, te, ti);
      r.pattern_idx = pattern_idx;
      r.text_idx = ti;
      r.text_start = UINT64_MAX;
      r.text_end = std::min<uint64_t>(c.pos, te) - ts;
      r.pattern_start = UINT64_MAX;
      // an end position past the text (overhang) leaves that many pattern characters outside
      r.pattern_end = plan.m - (c.pos > te ? std::min<uint64_t>(c.pos - te, plan.m) : 0);
      r.cost = c.cost;
      r.cigar_off = (uint32_t)R->pool.size();  // empty string: points at a NUL
      r.cigar_len = 0;
      R->matches.push_back(r);
    }
    if (R->pool.empty()) R->pool.push_back('\0');
    for (size_t i = first; i < R->matches.size(); ++i) R->matches[i].cigar_off = 0;
    return 0;
  }
  if (so.pin.h) {  // the records stay where the kernels wrote them (ScanOut::pin): the result owns the block now
    if (first != 0 || !R->pool.empty() || R->pin.h) return fail(SASSY_HIP_EINVAL, "internal: adopted block into a non-empty result");
    R->pin = so.pin;
    so.pin = PinBlock{};
    R->ext_matches = so.ext_matches;
    R->ext_n = so.ext_n;
    R->ext_pool = so.ext_pool;
    R->ext_pool_len = so.ext_pool_len;
    return 0;
  }
  if (first == 0 && R->pool.empty()) {  // the common single-scan case: adopt the buffers
    R->matches.swap(so.matches);
    R->pool.swap(so.pool);
    if (pattern_idx)
      for (sassy_hip_Match& r : R->matches) r.pattern_idx = pattern_idx;
    if (R->pool.empty()) R->pool.push_back('\0');
    return 0;
  }
  const size_t base = R->pool.size();
  if (base + so.pool.size() > 0xFFFFFFFFull)
    return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
  R->pool.append(so.pool);
  for (sassy_hip_Match r : so.matches) {
    r.pattern_idx = pattern_idx;
    r.cigar_off = (uint32_t)(r.cigar_off + base);
    R->matches.push_back(r);
  }
  return 0;
}

// 'N' counts of text ranges, on the host copy of the text when there is one, else on the device.
static int count_ns(sassy_SearcherType* S, const uint8_t* h_text, const uint8_t* d_text,
                    const std::vector<uint64_t>& ranges, std::vector<uint32_t>& counts) {
  const size_t n = ranges.size() / 2;
  counts.assign(n, 0);
  if (n == 0) return 0;
  if (h_text) {
    for (size_t i = 0; i < n; ++i) {
      uint32_t c = 0;
      for (uint64_t x = ranges[2 * i]; x < ranges[2 * i + 1]; ++x) c += ((h_text[x] | 0x20u) == 'n') ? 1u : 0u;
      counts[i] = c;
    }
    return 0;
  }
  if (int rc = S->d_range.reserve(2 * n)) return rc;
  if (int rc = S->d_ncount.reserve(n)) return rc;
  HIP_TRY(hipMemcpyAsync(S->d_range.p, ranges.data(), 2 * n * sizeof(uint64_t), hipMemcpyHostToDevice, S->stream));
  hipError_t le = launch_count_n(d_text, S->d_range.p, (uint32_t)n, S->d_ncount.p, S->stream);
  if (le != hipSuccess) return hip_fail(le, "N count kernel launch");
  HIP_TRY(hipMemcpyAsync(counts.data(), S->d_ncount.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, S->stream));
  HIP_TRY(hipStreamSynchronize(S->stream));
  return 0;
}

// n_count / denominator <= max_n_frac in f32, as the reference computes it (src/n_filter.rs:8-36)
static bool n_frac_ok(uint32_t n_count, uint64_t denominator, float max_n_frac) {
  return (float)n_count / (float)denominator <= max_n_frac;
}

// What the reference does between finding the end positions and returning the matches of one
// strand (src/search.rs:884-937): end-position callback, N-fraction pre-filter, only_best_match,
// N-fraction filter on the traced span.  All are filters on the report list, so applying them to the
// device-traced records gives the same result as tracing only the survivors.
// h_text / d_text: this strand's text (reversed for Rc) on the host (may be null) and the device.
int post_filter(sassy_SearcherType* S, ScanOut& so, const PatternPlan& plan, const uint8_t* pat, uint32_t k,
                       int strand, const uint8_t* h_text, const uint8_t* d_text, uint64_t tlen, bool with_trace,
                       const EndFilter& ef, const HostTexts* ht) {
  const bool n_filter = !std::isnan(S->max_n_frac);
  if (!ef.fn && !n_filter && !S->only_best) return 0;
  std::vector<char> keep(so.cands.size(), 1);
  auto compact = [&]() {
    size_t w = 0;
    for (size_t i = 0; i < so.cands.size(); ++i) {
      if (!keep[i]) continue;
      if ((int64_t)i == so.conditional_index) so.conditional_index = (int64_t)w;
      so.cands[w] = so.cands[i];
      if (with_trace) so.matches[w] = so.matches[i];
      ++w;
    }
    so.cands.resize(w);
    if (with_trace) so.matches.resize(w);
    keep.assign(w, 1);
  };
  if (ef.fn) {
    if (!h_text) return fail(SASSY_HIP_EINVAL, "search_with_fn needs the text in host memory");
    for (size_t i = 0; i < so.cands.size(); ++i) {
      const uint64_t end = std::min<uint64_t>(so.cands[i].pos, tlen);
      keep[i] = ef.fn(pat, plan.m, h_text, (size_t)end, strand, ef.user) ? 1 : 0;
    }
    compact();
  }
  if (n_filter) {  // satisfy_n_endpoint_filter (src/n_filter.rs:38-52)
    std::vector<uint64_t> ranges;
    ranges.reserve(2 * so.cands.size());
    const uint64_t mandatory = plan.m > k ? plan.m - k : 0;
    std::vector<uint64_t> ends_of_text;
    for (const Candidate& c : so.cands) {
      uint64_t ts, te, ti;
      text_bounds(ht, c, tlen, ts, te, ti);
      const uint64_t end = std::min<uint64_t>(c.pos, te);
      ranges.push_back(end - std::min<uint64_t>(end - ts, mandatory));
      ranges.push_back(end);
      ends_of_text.push_back(te);
    }
    std::vector<uint32_t> counts;
    if (int rc = count_ns(S, h_text, d_text, ranges, counts)) return rc;
    for (size_t i = 0; i < so.cands.size(); ++i) {
      const bool empty = ranges[2 * i] >= ends_of_text[i] || ranges[2 * i] == ranges[2 * i + 1];
      keep[i] = (empty || n_frac_ok(counts[i], (uint64_t)plan.m + k, S->max_n_frac)) ? 1 : 0;
    }
    compact();
  }
  if (S->only_best && !so.cands.empty()) {
    // minimal cost, then rightmost end (src/search.rs:1392-1412); one per text in a multi-text buffer
    // (the reports are sorted by position, so those of one text are adjacent)
    for (size_t i = 0; i < so.cands.size(); ++i) keep[i] = 0;
    size_t g0 = 0;
    while (g0 < so.cands.size()) {
      size_t g1 = g0 + 1;
      if (ht)
        while (g1 < so.cands.size() && (so.cands[g1].flags >> kCandTextShift) == (so.cands[g0].flags >> kCandTextShift)) ++g1;
      else
        g1 = so.cands.size();
      size_t best = g0;
      for (size_t i = g0 + 1; i < g1; ++i)
        if (so.cands[i].cost < so.cands[best].cost ||
            (so.cands[i].cost == so.cands[best].cost && so.cands[i].pos > so.cands[best].pos))
          best = i;
      keep[best] = 1;
      g0 = g1;
    }
    compact();
  }
  if (n_filter && with_trace) {  // traced_satisfy_n_frac (src/n_filter.rs:54-60)
    std::vector<uint64_t> ranges;
    ranges.reserve(2 * so.matches.size());
    std::vector<uint64_t> ends_of_text;
    for (size_t i = 0; i < so.matches.size(); ++i) {
      const sassy_hip_Match& r = so.matches[i];
      uint64_t ts, te, ti;
      text_bounds(ht, so.cands[i], tlen, ts, te, ti);
      ranges.push_back(ts + r.text_start);  // the records carry text-relative coordinates
      ranges.push_back(ts + r.text_end);
      ends_of_text.push_back(te);
    }
    std::vector<uint32_t> counts;
    if (int rc = count_ns(S, h_text, d_text, ranges, counts)) return rc;
    for (size_t i = 0; i < so.matches.size(); ++i) {
      const uint64_t len = ranges[2 * i + 1] - ranges[2 * i];
      keep[i] = (ranges[2 * i] >= ends_of_text[i] || len == 0 || n_frac_ok(counts[i], len, S->max_n_frac)) ? 1 : 0;
    }
    compact();
  }
  return 0;
}

int search_text(sassy_SearcherType* S, const uint8_t* pattern, size_t plen, const uint8_t* text,
                       size_t tlen, size_t k, uint32_t flags, uint64_t pattern_idx, bool fwd_strand,
                       bool rc_strand, sassy_hip_Result* R, const EndFilter& ef,
                       bool already_uploaded) {
  PatternPlan plan;
  std::string err;
  if (!make_plan(S->profile, pattern, plen, plan, err)) return fail(SASSY_HIP_EINVAL, err);
  if (k > 0x7FFFFFFFu) return fail(SASSY_HIP_EINVAL, "k too large");
  if (rc_strand && S->profile == PROFILE_ASCII)
    // the reference constructs such a searcher and panics at its first search: Profile::complement is
    // unimplemented for Ascii (the trait default, src/profiles.rs:57-60), reached from src/search.rs:813-820
    return fail(SASSY_HIP_EUNSUPPORTED, "reverse complement is not defined for the ascii alphabet");
  if (int rc = S->ensure_device()) return rc;
  const bool on_dev = (flags & SASSY_HIP_TEXT_ON_DEVICE) != 0;
  const bool all = (flags & SASSY_HIP_ALL_MINIMA) != 0;
  const bool wo = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
  if (tlen == 0) return 0;  // reference: no reports for an empty text (src/search.rs:1314-1316)

  const uint8_t* d_fwd = text;
  if (!on_dev) {
    if (!already_uploaded) {
      if (int rc = S->d_text.reserve(tlen + 64)) return rc;
      HIP_TRY(hipMemcpyAsync(S->d_text.p, text, tlen, hipMemcpyHostToDevice, S->stream));
    }
    d_fwd = S->d_text.p;
  } else if (((uintptr_t)text & 15) != 0) {
    return fail(SASSY_HIP_EINVAL, "device text pointer must be 16-byte aligned");
  }

  // complement(pattern) for the Rc strand (reference: src/search.rs:813-878)
  std::vector<uint8_t> cp;
  PatternPlan cplan;
  if (rc_strand) {
    cp.resize(plen);
    for (size_t i = 0; i < plen; ++i) cp[i] = complement_char(S->profile, pattern[i]);
    if (!make_plan(S->profile, cp.data(), plen, cplan, err)) return fail(SASSY_HIP_EINVAL, err);
  }
  // Both strands from one pass over the forward text: the forward job's prefilter also marks the Rc
  // strand's candidate blocks (in reversed-text coordinates), and the Rc job's chunk DP and traceback
  // read the forward buffer backwards -- no reversed copy, no second streaming pass.  Needs a filter
  // that can do it (bit-plane / counting) and no option that wants the reversed text as such.
  const int env_fuse = (int)S->sw.rc_fused;
  // the reference's lane reports (run_scan_ref_lanes): single texts, no overhang; each strand lane by lane
  const uint32_t ref_lanes = (S->ref_lanes == 4 || S->ref_lanes == 8) && std::isnan(S->alpha) ? S->ref_lanes : 0u;
  // Shapes of the paired filter: it exists as the fused launch of ONE strand only, and two of them (the Rc strand's on the
  // reversed copy) beat the forward strand's streaming DP with the Rc marks in it (m = 23, k = 3: 1.4 against 1.8 ms).
  // SASSY_HIP_PAIR_RC=0: as before.
  const int env_pair_rc = (int)S->sw.pair_rc;
  uint32_t ps_ = 0, pq_ = 0;
  const bool pair_strands = env_pair_rc != 0 && S->sw.pair != 0 && prefilter_mode(S) < 0 && S->fuse && !wo && k <= 0xFFFFu &&
                            pair_geometry(plan.m, (uint32_t)k, &ps_, &pq_) &&
                            (S->profile == PROFILE_DNA ||
                             (S->profile == PROFILE_IUPAC && ps_ <= 3 && pair_geometry(plan.m, (uint32_t)k, &ps_, &pq_) &&
                              plain_prefix(pattern, plen) >= (size_t)2 * ps_ * pq_));
  const bool can_fuse = fwd_strand && rc_strand && env_fuse != 0 && !ef.fn && std::isnan(S->max_n_frac) &&
                        std::isnan(S->alpha) && S->profile != PROFILE_ASCII && ref_lanes == 0 && !pair_strands;
  bool rc_by_bitmap = false;

  // Two searches, one per strand (the Rc strand's on the reversed copy) -- the paired filter's shapes, searchers with an
  // N filter: both IN FLIGHT, each on a lane of its own, as two tickets of a stream of searches are -- the forward
  // strand's chunk DP tail and traceback run under the Rc strand's filter (m = 23, k = 3: 1.32 -> 1.1x ms).
  // SASSY_HIP_STRANDS_IN_FLIGHT=0: one after the other.
  const bool env_two = S->sw.strands_in_flight != 0;
  const bool env_one_lane = S->sw.lanes <= 1;
  if (fwd_strand && rc_strand && !can_fuse && env_two && env_one_lane && ref_lanes == 0 && !ef.fn && std::isnan(S->alpha) &&
      S->profile != PROFILE_ASCII) {
    const bool reuse = on_dev && (flags & SASSY_HIP_TEXT_UNCHANGED) && S->rev_src == d_fwd && S->rev_len == tlen &&
                       S->d_rev.p != nullptr;
    if (!reuse) {
      S->rev_src = nullptr;
      if (int rc = S->d_rev.reserve(tlen + 64)) return rc;
      hipError_t le = launch_reverse(d_fwd, S->d_rev.p, tlen, S->stream);
      if (le != hipSuccess) return hip_fail(le, "reverse kernel launch");
      if (on_dev) { S->rev_src = d_fwd; S->rev_len = tlen; }
    }
    ScanQueue queue(S, [&](uint64_t strand, ScanOut& so, const PatternPlan& pl, const uint8_t* pat) -> int {
      if (int rc = post_filter(S, so, pl, pat, (uint32_t)k, (int)strand, strand == 0 && !on_dev ? text : nullptr,
                               strand ? S->d_rev.p : d_fwd, tlen, !wo, ef)) return rc;
      size_t first = 0;
      if (int rc = append_matches(so, tlen, pl, wo, pattern_idx, R, first)) return rc;
      if (strand)
        for (size_t i = first; i < R->matches.size(); ++i) {
          sassy_hip_Match& r = R->matches[i];
          const uint64_t rs = r.text_start, re = r.text_end;
          r.strand = 1;
          r.text_start = tlen - re;
          r.text_end = wo ? UINT64_MAX : tlen - rs;  // reference: src/search.rs:868-873
        }
      return 0;
    });
    const TextTable no_texts{};
    if (int rc = queue.submit(plan, pattern, ShardView{d_fwd, tlen, 0, 0, true, true}, no_texts, (uint32_t)k, all, !wo, tlen, 0)) return rc;
    if (int rc = queue.submit(cplan, cp.data(), ShardView{S->d_rev.p, tlen, 0, 0, true, true}, no_texts, (uint32_t)k, all, !wo, tlen, 1)) {
      (void)queue.drain_all();
      return rc;
    }
    return queue.drain_all();
  }

  if (fwd_strand) {
    ShardView sh{d_fwd, tlen, 0, 0, true, true};
    ScanOut so;
    if (can_fuse) {
      const uint64_t nb = ((uint64_t)tlen + 63) / 64;
      if (int rc = S->d_rc_bitmap.reserve((nb + 63) / 64 + 4)) return rc;
      ScanJob job(S, S->lanes[0], sh, plan, (uint32_t)k, all, pattern, !wo, tlen);
      job.texts.all_minima = all ? 1u : 0u;
      job.rc_bitmap = S->d_rc_bitmap.p;
      job.rc_pat = cp.data();
      job.signal_filter_done = true;
      if (int rc = job.prepare()) return rc;
      if (!job.empty)
        if (int rc = job.enqueue(0)) return rc;
      rc_by_bitmap = job.rc_marked && !job.empty;
      // the Rc strand's chunk list / DP / rank / traceback -- short, latency-bound kernels -- run on a
      // second lane next to the forward strand's, behind the shared filter pass
      ScanOut so_rc;
      std::unique_ptr<ScanJob> rj;
      if (rc_by_bitmap) {
        rj.reset(new ScanJob(S, S->lanes[1], sh, cplan, (uint32_t)k, all, cp.data(), !wo, tlen));
        rj->texts.all_minima = all ? 1u : 0u;
        rj->ext_bitmap = S->d_rc_bitmap.p;
        rj->ext_q = job.q;
        rj->ext_wait = S->lanes[0].ev_filter_done;
        rj->rev_n = tlen;
        if (int rc = rj->prepare()) return rc;
        if (!rj->empty)
          if (int rc = rj->enqueue(0)) return rc;
      }
      if (int rc = job.finish(so)) return rc;
      if (int rc = post_filter(S, so, plan, pattern, (uint32_t)k, 0, on_dev ? nullptr : text, d_fwd, tlen, !wo, ef)) return rc;
      size_t first = 0;
      if (int rc = append_matches(so, tlen, plan, wo, pattern_idx, R, first)) return rc;
      if (rj) {
        if (int rc = rj->finish(so_rc)) return rc;
        if (int rc = post_filter(S, so_rc, cplan, cp.data(), (uint32_t)k, 1, nullptr, nullptr, tlen, !wo, ef)) return rc;
        if (int rc = append_matches(so_rc, tlen, cplan, wo, pattern_idx, R, first)) return rc;
        for (size_t i = first; i < R->matches.size(); ++i) {
          sassy_hip_Match& r = R->matches[i];
          const uint64_t rs = r.text_start, re = r.text_end;
          r.strand = 1;
          r.text_start = tlen - re;
          r.text_end = wo ? UINT64_MAX : tlen - rs;  // reference: src/search.rs:868-873
        }
      }
    } else {
      if (ref_lanes) {
        if (int rc = run_scan_ref_lanes(S, d_fwd, tlen, plan, (uint32_t)k, all, pattern, !wo, ref_lanes, so)) return rc;
      } else if (int rc = run_scan(S, sh, plan, (uint32_t)k, all, pattern, !wo, tlen, so)) return rc;
      if (int rc = post_filter(S, so, plan, pattern, (uint32_t)k, 0, on_dev ? nullptr : text, d_fwd, tlen, !wo, ef)) return rc;
      size_t first = 0;
      if (int rc = append_matches(so, tlen, plan, wo, pattern_idx, R, first)) return rc;
    }
  }
  if (rc_strand && rc_by_bitmap) {
    // done above, next to the forward strand
  } else if (rc_strand) {
    // complement(pattern) against a reversed copy of the text, coordinates mapped back
    // the caller may promise that a device text did not change since this searcher last saw it:
    // the reversed copy (n bytes read + n written, more than the search itself) is then still valid
    const bool reuse = on_dev && (flags & SASSY_HIP_TEXT_UNCHANGED) && S->rev_src == d_fwd && S->rev_len == tlen &&
                       S->d_rev.p != nullptr;
    if (!reuse) {
      S->rev_src = nullptr;
      if (int rc = S->d_rev.reserve(tlen + 64)) return rc;
      hipError_t le = launch_reverse(d_fwd, S->d_rev.p, tlen, S->stream);
      if (le != hipSuccess) return hip_fail(le, "reverse kernel launch");
      if (on_dev) { S->rev_src = d_fwd; S->rev_len = tlen; }
    }
    ShardView sh{S->d_rev.p, tlen, 0, 0, true, true};
    ScanOut so;
    if (ref_lanes) {
      if (int rc = run_scan_ref_lanes(S, S->d_rev.p, tlen, cplan, (uint32_t)k, all, cp.data(), !wo, ref_lanes, so)) return rc;
    } else if (int rc = run_scan(S, sh, cplan, (uint32_t)k, all, cp.data(), !wo, tlen, so)) return rc;
    std::vector<uint8_t> h_rev;  // the callback sees the reversed text, like the reference's
    if (ef.fn && !on_dev) h_rev.assign(std::reverse_iterator<const uint8_t*>(text + tlen),
                                       std::reverse_iterator<const uint8_t*>(text));
    if (int rc = post_filter(S, so, cplan, cp.data(), (uint32_t)k, 1, ef.fn && !on_dev ? h_rev.data() : nullptr,
                             S->d_rev.p, tlen, !wo, ef)) return rc;
    size_t first = 0;
    if (int rc = append_matches(so, tlen, cplan, wo, pattern_idx, R, first)) return rc;
    for (size_t i = first; i < R->matches.size(); ++i) {
      sassy_hip_Match& r = R->matches[i];
      const uint64_t rs = r.text_start, re = r.text_end;
      r.strand = 1;
      r.text_start = tlen - re;
      r.text_end = wo ? UINT64_MAX : tlen - rs;  // reference: src/search.rs:868-873
    }
  }
  return 0;
}

}  // namespace sassy_hip
