// many_patterns.hip -- search_encoded_patterns / search_many: every pattern against the text(s) in one pass (pattern-tiled
// scan, seeded search, per-text tiled scan for overhang), the list -> report rule -> traceback -> records tail, and the
// batch layouts of many host texts.  Reference: src/pattern_tiling/*, src/search.rs:404-423, 548-683.
#include "host_internal.h"

namespace sassy_hip {

static int finish_pattern_list(sassy_SearcherType* s, const sassy_hip_Encoded* e, const PatternPlan& plan0,
                               const uint8_t* tptr, const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all,
                               bool wo, uint32_t count, bool copies, sassy_hip_Result* R,
                               const TextTable* tt = nullptr, const HostTexts* ht = nullptr, ManyDefer* defer = nullptr) {
  if (defer && (wo || all || !tt)) return fail(SASSY_HIP_EINVAL, "internal: deferred records need a traced, multi-text search");
  ScanLane& L = s->lanes[0];
  hipStream_t st = s->stream;
  const uint32_t m = (uint32_t)e->plen;
  uint32_t counts[2] = {count, 0};
  // ---- (pattern, position) order, then the report rule ----
  if (int rc = L.d_sorted.reserve(count)) return rc;
  if (int rc = L.d_sort.reserve(std::max(sort_scratch_bytes(count), select_scratch_bytes(count)))) return rc;
  // (key = pattern, position: only the bits the text's length and the number of patterns need -- a radix pass per 8)
  int pos_bits = 8, tag_bits = 1;
  while (pos_bits < 40 && ((text_len + 256) >> pos_bits) != 0) ++pos_bits;
  while (tag_bits < 24 && (e->patterns.size() >> tag_bits) != 0) ++tag_bits;
  hipError_t le = launch_sort_candidates(s->d_tiled_list.p, L.d_sorted.p, count, L.d_sort.p, L.d_sort.cap, st, pos_bits, pos_bits + tag_bits);
  if (le != hipSuccess) return hip_fail(le, "report sort launch");
  const Candidate* d_rep = L.d_sorted.p;
  uint32_t n_rep = count;
  if (!all || copies) {
    if (int rc = s->d_tiled_sel.reserve(count)) return rc;
    le = launch_select_reports(L.d_sorted.p, count, s->d_tiled_sel.p, s->d_tiled_cnt.p + 1, L.d_sort.p, L.d_sort.cap, st,
                               all ? 1 : 0);
    if (le != hipSuccess) return hip_fail(le, "report selection launch");
    HIP_TRY(hipMemcpyAsync(counts + 1, s->d_tiled_cnt.p + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    d_rep = s->d_tiled_sel.p;
    n_rep = counts[1];
  } else {
    HIP_TRY(hipMemcpyAsync(s->d_tiled_cnt.p + 1, &count, 4, hipMemcpyHostToDevice, st));
  }
  if (n_rep == 0) return 0;
  std::vector<uint32_t> rtext;
  if (tt) {  // several texts: which one a report belongs to; reports inside separators
    if (d_rep == L.d_sorted.p) {  // (search_all without copies: the sorted list itself is the report list)
      if (int rc = s->d_tiled_sel.reserve(n_rep)) return rc;
      HIP_TRY(hipMemcpyAsync(s->d_tiled_sel.p, L.d_sorted.p, (size_t)n_rep * sizeof(Candidate), hipMemcpyDeviceToDevice, st));
      d_rep = s->d_tiled_sel.p;
    }
    if (int rc = s->d_tiled_rtext.reserve(n_rep)) return rc;
    le = launch_assign_texts(s->d_tiled_sel.p, n_rep, *tt, s->d_tiled_rtext.p, st);
    if (le != hipSuccess) return hip_fail(le, "text assignment launch");
    rtext.resize(n_rep);
  }

  // ---- traceback: one wavefront per report, the report's pattern comes with it ----
  std::vector<Candidate> reps;  // (sized where the host's way begins: zero-filling 16 bytes per report took 0.8 ms of a read batch)
  std::vector<sassy_hip_Match> rows;
  std::string pool;
  uint32_t str_stride = 0;
  if (!wo) {
    const uint64_t band = ((uint64_t)(m + 1) * (2ull * k + 3) + 3) / 4 * 4;
    const uint64_t win = ((uint64_t)m + k + 15 + 15) / 16 * 16;
    const uint64_t opsb = ((uint64_t)m + k + 1 + 3) / 4 * 4;
    const uint64_t strb = ((2ull * (m + k + 1) + 2 + 15) / 16 * 16);
    const uint64_t wstride = (band + win + opsb + strb + kTraceWaveDummy + 15) / 16 * 16;
    str_stride = (uint32_t)strb;
    if ((uint64_t)n_rep * strb > 0xFFFFFFFFull)
      return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
    ScanLane& LT = defer ? s->lanes[defer->lane] : L;  // (the same stream: only the buffers are the other lane's)
    if (int rc = LT.d_trace.reserve(n_rep)) return rc;
    if (int rc = LT.d_str.reserve((size_t)n_rep * strb)) return rc;
    TraceParams T{};
    T.text = tptr;
    T.total_len = text_len;
    T.cand = d_rep;
    T.cand_count = s->d_tiled_cnt.p + 1;
    T.cand_cap = n_rep;
    T.m = m;
    T.k = k;
    T.profile = (uint32_t)s->profile;
    T.pattern = s->d_tiled_pat.p;
    T.pattern_stride = m;
    T.scratch_stride = (uint32_t)wstride;
    T.band_bytes = (uint32_t)band;
    T.win_bytes = (uint32_t)win;
    T.out = LT.d_trace.p;
    T.out_str = LT.d_str.p;
    T.str_stride = str_stride;
    T.ops_bytes = (uint32_t)opsb;
    T.wave_mode = 1;
    T.count_min = 0;
    T.count_max = 0xFFFFFFFFu;
    T.max_overhang = 0xFFFFFFFFu;
    if (!std::isnan(s->alpha)) {  // overhang (the one-pass search of a batch: search_many_pertext)
      T.use_alpha = 1u;
      T.alpha = s->alpha;
      T.max_overhang = s->max_overhang >= 0 ? (uint32_t)std::min<long>(s->max_overhang, 0x7FFFFFFF) : 0xFFFFFFFFu;
    }
    if (tt) {
      T.texts = *tt;
      T.report_text = s->d_tiled_rtext.p;
    }
    HIP_TRY(hipEventRecord(s->ev_a_multi(), st));
    uint32_t trace_grid = (uint32_t)std::min<uint64_t>(1024, ((uint64_t)n_rep + 3) / 4);
    {
      // Dense lists (a guide set on a genome: 10^7 reports) with a narrow band (k <= 6: the band row in registers): a
      // thread per report, as many workgroups as the chip holds -- 0.7 ns per report against the wavefront shape's 2.3.
      // (SASSY_HIP_ENCODED_TRACE_THREADS=0: never; =<n>: from n reports on -- read per call: tests flip it)
      const int env_tt = (int)s->sw.encoded_trace_threads;
      const bool env_off = env_tt == 0;
      const uint32_t from = env_tt > 0 ? (uint32_t)env_tt : 65536u;
      uint64_t stride_t = band + win + opsb + strb;
      if ((stride_t / 4) % 2 == 0) stride_t += 4;  // odd number of LDS words: conflict-free slices
      const uint64_t pat_bytes = ((uint64_t)m + 15) / 16 * 16;
      if (!env_off && k <= 6 && !T.use_alpha && n_rep >= from && 64 * stride_t + pat_bytes <= kTraceLdsLimit) {
        T.wave_mode = 0;
        T.scratch = nullptr;
        T.scratch_stride = (uint32_t)stride_t;
        const uint64_t nthreads = std::min<uint64_t>(256ull * 64ull * std::max<uint64_t>(1, (160ull * 1024) / (64 * stride_t + pat_bytes)), 131072);
        trace_grid = (uint32_t)(nthreads / 64);
      }
    }
    le = launch_trace(T, trace_grid, st);
    if (le != hipSuccess) return hip_fail(le, "trace kernel launch");
    HIP_TRY(hipEventRecord(s->ev_multi, st));
    if (defer) {
      defer->part = ManyPart{LT.d_trace.p, reinterpret_cast<const char*>(LT.d_str.p), n_rep};
      defer->str_stride = str_stride;
      return 0;
    }
    // Dense results (a CRISPR guide set on a genome: 10^7 matches): every report is a record, nothing is filtered or
    // dropped, the result is empty so far -- the rows get their final pattern index and strand on the device and leave,
    // with the cigar strings, by two DMA copies into ONE pinned block that the result keeps (as assemble_many and the
    // dense single-pattern searches do).  The host used to take 128 bytes per match through zero-filled vectors, pageable
    // copies and three loops: 0.9 of the 1.13 s of 312 guides x both strands on the genome-like text (17 M matches).
    // SASSY_HIP_ENCODED_PIN=0: the host's way (tests compare the two record by record).
    const bool env_nopin = s->sw.encoded_pin == 0;
    const bool no_filters = std::isnan(s->max_n_frac) && !s->only_best;
    if (!env_nopin && no_filters && !tt && R->matches.empty() && R->pool.empty() && !R->pin.h && n_rep >= 1024 &&
        text_len < (1ull << 39) && m + k < 0xFFFFu && k < 0x7FFFu && strb % 16 == 0 &&
        // (the compacted pool's offsets are 32-bit words: a result whose strings could pass 4 GiB takes the host's way)
        (uint64_t)n_rep * strb <= 0xFFFFFFFFull) {
      const size_t n = n_rep;
      ScanLane& LO = s->lanes[2];  // (its record / string buffers take the ordered result)
      // (a second set of device buffers: where they cannot be had, the host's way still works -- as for the pinned block)
      bool have = LO.d_trace.reserve(n) == 0 && LO.d_str.reserve(n * strb) == 0 && L.d_sort.reserve(encoded_scratch_bytes(n_rep)) == 0;
      if (!have) (void)hipGetLastError();
      if (have) {
      int key_bits = 40;
      while (key_bits < 64 && ((uint64_t)e->n_original >> (key_bits - 39)) != 0) ++key_bits;
      if (int rc = L.d_flags.reserve(4)) return rc;
      HIP_TRY(hipMemsetAsync(L.d_flags.p, 0, 8, st));
      le = launch_assemble_encoded(L.d_trace.p, reinterpret_cast<const char*>(L.d_str.p), n_rep, e->n_original, (uint32_t)strb, key_bits,
                                   LO.d_trace.p, reinterpret_cast<char*>(LO.d_str.p), L.d_flags.p, L.d_sort.p, L.d_sort.cap, st);
      if (le != hipSuccess) return hip_fail(le, "result ordering launch");
      uint32_t flags2[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(flags2, L.d_flags.p, 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (flags2[0]) return fail(SASSY_HIP_EINVAL, "traceback failed for a reported end position (internal error)");
      const size_t pool_bytes = flags2[1];  // (the strings without their slots' padding)
      const size_t rows_off = 256, strs_off = (rows_off + n * sizeof(MatchOut) + 255) / 256 * 256;
      const size_t bytes = strs_off + pool_bytes + 256;
      if (L.reserve_pinned(bytes) == 0) {
        HIP_TRY(hipMemcpyAsync(L.h_pin + rows_off, LO.d_trace.p, n * sizeof(MatchOut), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(L.h_pin + strs_off, LO.d_str.p, pool_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_a_multi(), s->ev_multi));
        s->stats.trace_ms += ms;
        const sassy_hip_Match* hm = reinterpret_cast<const sassy_hip_Match*>(L.h_pin + rows_off);
        const char* hs = reinterpret_cast<const char*>(L.h_pin + strs_off);
        if (g_pin_pool.may_adopt(L.h_pin_cap)) {
          R->pin = L.take_pin();
          R->ext_matches = hm;
          R->ext_n = n;
          R->ext_pool = hs;
          R->ext_pool_len = pool_bytes;
        } else {
          R->matches.assign(hm, hm + n);
          R->pool.assign(hs, pool_bytes);
        }
        return 0;
      }
      (void)hipGetLastError();  // (no pinned block of that size: the host's way)
      }
    }
    rows.resize(n_rep);
    pool.resize((size_t)n_rep * strb);
    if (int rc = L.download(rows.data(), L.d_trace.p, (size_t)n_rep * sizeof(MatchOut))) return rc;
    if (int rc = L.download(&pool[0], L.d_str.p, pool.size())) return rc;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_a_multi(), s->ev_multi));
    s->stats.trace_ms += ms;
  }
  reps.resize(n_rep);
  if (int rc = L.download(reps.data(), d_rep, (size_t)n_rep * sizeof(Candidate))) return rc;
  if (tt)
    if (int rc = L.download(rtext.data(), s->d_tiled_rtext.p, (size_t)n_rep * sizeof(uint32_t))) return rc;
  if (tt && all) {  // drop the reports that lie in separators (their records were not written)
    size_t w = 0;
    for (size_t i = 0; i < n_rep; ++i) {
      if (reps[i].flags & kCandDrop) continue;
      reps[w] = reps[i];
      rtext[w] = rtext[i];
      if (!wo) rows[w] = rows[i];
      ++w;
    }
    n_rep = (uint32_t)w;
    reps.resize(w);
    rtext.resize(w);
    if (!wo) rows.resize(w);
  }
  if (!wo)
    for (const sassy_hip_Match& r : rows)
      if (r.pad_[0] == kTraceFailed)
        return fail(SASSY_HIP_EINVAL, "traceback failed for a reported end position (internal error)");

  // ---- per pattern: the searcher's report filters, then the records ----
  const bool filters = !std::isnan(s->max_n_frac) || s->only_best;
  if (!filters && !wo) {  // the records are finished: adopt them (or append them behind what R holds already)
    const size_t first = R->matches.size(), base = R->pool.size();
    if (first == 0 && base == 0) {
      R->matches.swap(rows);
      R->pool.swap(pool);
    } else {
      if (base + pool.size() > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
      R->pool.append(pool);
      R->matches.insert(R->matches.end(), rows.begin(), rows.end());
    }
    for (size_t i = first; i < R->matches.size(); ++i) {
      sassy_hip_Match& r = R->matches[i];
      const uint64_t p = r.pattern_idx;
      r.pattern_idx = p % e->n_original;
      r.strand = p >= e->n_original ? 1 : 0;
      r.cigar_off += (uint32_t)base;
    }
    return 0;
  }
  size_t i0 = 0;
  while (i0 < n_rep) {
    const uint32_t p = reps[i0].flags >> kCandTextShift;
    size_t i1 = i0;
    while (i1 < n_rep && (reps[i1].flags >> kCandTextShift) == p) ++i1;
    ScanOut so;
    so.cands.assign(reps.begin() + i0, reps.begin() + i1);
    for (size_t i = i0; i < i1; ++i) so.cands[i - i0].flags = tt ? rtext[i] << kCandTextShift : 0u;
    if (!wo) {
      so.matches.assign(rows.begin() + i0, rows.begin() + i1);
      // the records' cigar offsets point into the whole pool: this pattern's share is cut out and they are rebased
      // (after dropped reports the records are no longer consecutive in the pool: take the span they cover)
      const size_t lo = so.matches.front().cigar_off, hi = (size_t)so.matches.back().cigar_off + str_stride;
      so.pool.assign(pool, lo, hi - lo);
      for (sassy_hip_Match& r : so.matches) r.cigar_off -= (uint32_t)lo;
    }
    if (int rc = post_filter(s, so, plan0, e->patterns[p].data(), k, 0, h_text, tptr, text_len, !wo, EndFilter(), ht)) return rc;
    size_t first = 0;
    if (int rc = append_matches(so, text_len, plan0, wo, p % e->n_original, R, first, ht)) return rc;
    for (size_t i = first; i < R->matches.size(); ++i) {
      R->matches[i].pattern_idx = p % e->n_original;
      R->matches[i].strand = p >= e->n_original ? 1 : 0;
    }
    i0 = i1;
  }
  return 0;
}

// Overhang in one pass over a batch of texts in whole blocks (tiled_pertext_kernel): where the texts lie, the virtual
// columns behind each, the overhang column every text starts from.
struct TiledPerText {
  const uint64_t* d_start;
  const uint64_t* d_len;
  uint32_t n;
  uint32_t steps;
  float alpha;
  unsigned long long vp;
  int32_t cost0;
  uint32_t edge_cols;  // != 0: only the end positions overhang changes (the seeded search lists the inside of the texts)
};
// The pattern-tiled kernel over one device buffer: every (pattern, end position, cost <= k) into `list` (grown on
// demand; the counter is the device word d_count).  *ok = false: more than 2^26 of them.  classes: 4 (Dna codes) or
// 16 (Iupac base sets; 'X' matches nothing).
static int tiled_scan_list(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* buf, uint64_t len, uint32_t k,
                           uint32_t classes, DevBuf<unsigned long long>& d_peq, DevBuf<Candidate>& list, uint32_t* d_count,
                           uint32_t* count, bool* ok, uint64_t* n_waves, const uint32_t* d_keep_bits = nullptr,
                           const TiledPerText* pt = nullptr) {
  *ok = false;
  *count = 0;
  const size_t npat = e->patterns.size();
  const uint32_t m = (uint32_t)e->plen;
  hipStream_t st = s->stream;
  // ---- match masks: bit j of peq[class][pattern] = row j of the pattern matches a text character of that class ----
  const uint32_t npad = (uint32_t)((npat + 63) / 64 * 64);
  std::vector<unsigned long long> peq((size_t)classes * npad, 0ull);
  for (size_t p = 0; p < npat; ++p) {
    const uint8_t* pt = e->patterns[p].data();
    for (uint32_t j = 0; j < m; ++j) {
      if (classes == 4) {
        peq[(size_t)((pt[j] >> 1) & 3u) * npad + p] |= 1ull << j;  // src/profiles/dna.rs:19-40
      } else {
        const uint32_t set = iupac_code(pt[j]) & 0x0Fu;              // src/profiles/iupac.rs:18-36
        for (uint32_t c = 1; c < 16; ++c)
          if (set & c) peq[(size_t)c * npad + p] |= 1ull << j;
      }
    }
  }
  if (int rc = d_peq.reserve(peq.size())) return rc;
  HIP_TRY(hipMemcpyAsync(d_peq.p, peq.data(), peq.size() * 8, hipMemcpyHostToDevice, st));

  TiledParams P{};
  P.skew = (uint32_t)((uintptr_t)buf & 63u);
  P.text_aligned = buf - P.skew;
  P.text_len = len;
  P.peq = d_peq.p;
  P.npat = (uint32_t)npat;
  P.npat_padded = npad;
  P.n_groups = npad / 64;
  P.m = m;
  P.k = k;
  P.classes = classes;
  P.warm_blocks = (m + k + 63) / 64;
  P.keep_bits = d_keep_bits;
  // (the zones' list -- the one caller with keep bits -- is read by map_zone_list_kernel, which skips empty records)
  // Every wave that lists anything takes a whole range of 1 024 slots (a range holds at least one emit call's 512
  // records: tiled_kernel.hip, EmitCursor), and the counter counts slots: the first attempt reserves a range per wave.
  P.cand_chunk = (d_keep_bits && !pt) ? 1024u : 0u;
  {
    const uint64_t span = (uint64_t)P.skew + len;
    const uint64_t waves_wanted = 16384;
    const uint64_t chunks_wanted = std::max<uint64_t>(1, waves_wanted / P.n_groups);
    uint64_t chunk = std::max<uint64_t>(512, (span + chunks_wanted - 1) / chunks_wanted);
    chunk = std::min<uint64_t>((chunk + 63) / 64 * 64, 1u << 20);
    P.chunk = (uint32_t)chunk;
    P.n_chunks = (span + chunk - 1) / chunk;
  }
  *n_waves = P.n_chunks * P.n_groups;
  if (pt) {  // a batch of texts, each from its own overhang column to its last virtual column
    P.skew = 0;
    P.text_aligned = buf;
    P.texts_start = pt->d_start;
    P.texts_len = pt->d_len;
    P.n_texts = pt->n;
    P.ov_steps = pt->steps;
    P.alpha = pt->alpha;
    P.ov_vp = pt->vp;
    P.ov_cost0 = pt->cost0;
    P.edge_cols = pt->edge_cols;
    const uint64_t waves_wanted = 32768;
    P.texts_per_wave = (uint32_t)std::max<uint64_t>(1, ((uint64_t)pt->n * P.n_groups + waves_wanted - 1) / waves_wanted);
    *n_waves = (((uint64_t)pt->n + P.texts_per_wave - 1) / P.texts_per_wave) * P.n_groups;
  }
  const uint64_t kMaxList = 1ull << 28;  // 4 GiB of (pattern, position, cost) records: beyond that, per-pattern scans
  uint32_t got = 0;
  for (int attempt = 0;; ++attempt) {
    const size_t ranges = P.cand_chunk ? (size_t)std::min<uint64_t>(*n_waves * P.cand_chunk, 1ull << 22) : 0;
    if (int rc = list.reserve(std::max<size_t>((size_t)1 << 18, std::max<size_t>(ranges, got) + 1024))) return rc;
    P.cand = list.p;
    P.cand_cap = (uint32_t)std::min<size_t>(list.cap, 0xFFFFFFFFu);
    P.cand_stop = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(P.cand_cap, kMaxList) + (1u << 20), 0xF0000000ull);
    P.cand_count = d_count;
    HIP_TRY(hipMemsetAsync(d_count, 0, 4, st));
    HIP_TRY(hipEventRecord(s->ev_a_multi(), st));
    hipError_t le = pt ? launch_tiled_pertext(P, st) : launch_tiled_scan(P, st);
    if (le != hipSuccess) return hip_fail(le, "pattern-tiled scan launch");
    HIP_TRY(hipEventRecord(s->ev_multi, st));
    HIP_TRY(hipMemcpyAsync(&got, d_count, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // (`peq` stays alive until here)
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_a_multi(), s->ev_multi));
    s->stats.scan_ms += ms;
    s->stats.scan_launches += 1;
    if (got <= P.cand_cap) break;
    if (got > kMaxList || attempt == 2) {
      return 0;  // *ok stays false
    }
  }
  *count = got;
  *ok = true;
  return 0;
}

// search_encoded_patterns in ONE pass: the pattern-tiled scan (tiled_kernel.hip; reference v2,
// src/pattern_tiling/search.rs:326-425 + general.rs:335-404).  All (rc-expanded) patterns advance together over
// the text, 64 per wavefront; the kernel lists every (pattern, end position) with cost <= k, the device sorts the
// list by (pattern, position), applies the report rule to each run (sort_kernels.hip: flag_reports_kernel) and
// traces the reports (trace_wave_kernel with one pattern per report).  *done = false: too many end positions for
// this shape (k close to m on a long text) -- the caller runs one scan per pattern instead.
int search_encoded_tiled(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr,
                                const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all, bool wo,
                                sassy_hip_Result* R, bool* done, const TextTable* tt,
                                const HostTexts* ht, ManyDefer* defer, const TiledPerText* pt) {
  *done = false;
  const size_t npat = e->patterns.size();
  const uint32_t m = (uint32_t)e->plen;
  std::string err;
  PatternPlan plan0;
  for (size_t p = 0; p < npat; ++p) {  // what the reference's encode would reject (tqueries.rs:60-65, iupac.rs:19-24)
    PatternPlan pl;
    if (!make_plan(s->profile, e->patterns[p].data(), m, p == 0 ? plan0 : pl, err)) return fail(SASSY_HIP_EINVAL, err);
  }
  std::vector<uint8_t> flat(npat * (size_t)m);
  for (size_t p = 0; p < npat; ++p) memcpy(&flat[p * m], e->patterns[p].data(), m);
  if (int rc = s->d_tiled_pat.reserve(flat.size() + 64)) return rc;
  if (int rc = s->d_tiled_cnt.reserve(16)) return rc;
  hipStream_t st = s->stream;
  HIP_TRY(hipMemsetAsync(s->d_tiled_cnt.p, 0, 64, st));
  HIP_TRY(hipMemcpyAsync(s->d_tiled_pat.p, flat.data(), flat.size(), hipMemcpyHostToDevice, st));
  uint32_t count = 0;
  uint64_t n_waves = 0;
  bool ok = false;
  if (int rc = tiled_scan_list(s, e, tptr, text_len, k, s->profile == PROFILE_DNA ? 4u : 16u, s->d_tiled_peq, s->d_tiled_list,
                               s->d_tiled_cnt.p, &count, &ok, &n_waves, nullptr, pt)) return rc;
  if (!ok) return 0;  // *done stays false
  s->stats.text_bytes += text_len;
  s->stats.chunks += n_waves;
  s->stats.filtered = 5;
  s->stats.candidates += count;
  *done = true;
  if (count == 0) return 0;
  return finish_pattern_list(s, e, plan0, tptr, h_text, text_len, k, all, wo, count, false, R, tt, ht, defer);
}

// The seeded search on a text with other letters than ACGT (Iupac searcher; seed_kernels.hip, second half).  The
// seeded pass has filled d_tiled_list with *list_count records that are exact wherever the m + k characters in front
// of the end position are plain.  Here: find the runs of other letters, drop the records whose window touches one,
// and put in their place what the pattern-tiled scan (16 Iupac classes) finds on a gathered copy of the runs'
// neighbourhoods.  A run of full wildcards (N, non-letters) of more than m + 1 characters is not copied
// whole: inside it every pattern's cost is constant, so the list leaves those positions out and marks the last one
// in front of them (kCandCont) -- the report rule then sees one plateau (search_all needs every position: no cut).
// *ok = false: too many runs / too much text around them -- the caller takes another way.
static int seeded_dirty_zones(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr, uint64_t n, uint32_t k,
                              bool all, uint32_t* list_count, bool* ok) {
  *ok = false;
  ScanLane& L = s->lanes[0];
  hipStream_t st = s->stream;
  const uint64_t C = (uint64_t)e->plen + k;
  const uint32_t cap = 1u << 20;  // runs of other letters (a human genome has ~10^3; this synthetic one 10^5)
  uint32_t* d_cnt = s->d_tiled_cnt.p + 8;  // words 8..10: runs' starts, ends, hard letters; 12, 13: list counters
  if (int rc = s->d_zone_u64.reserve(3 * (size_t)cap)) return rc;
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 32, st));
  hipError_t le = launch_dirty_scan(tptr, n, s->d_zone_u64.p, s->d_zone_u64.p + cap, s->d_zone_u64.p + 2 * cap, cap, d_cnt, st);
  if (le != hipSuccess) return hip_fail(le, "letter run scan launch");
  uint32_t cnt[3] = {0, 0, 0};
  HIP_TRY(hipMemcpyAsync(cnt, d_cnt, 12, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (cnt[0] > cap || cnt[1] > cap || cnt[2] > cap) return 0;
  if (cnt[0] != cnt[1]) return fail(SASSY_HIP_EINVAL, "letter run scan: unpaired run ends (internal error)");
  std::vector<unsigned long long> starts(cnt[0]), ends(cnt[1]), hard(cnt[2]);
  if (cnt[0]) {
    HIP_TRY(hipMemcpyAsync(starts.data(), s->d_zone_u64.p, cnt[0] * 8ull, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(ends.data(), s->d_zone_u64.p + cap, cnt[1] * 8ull, hipMemcpyDeviceToHost, st));
  }
  if (cnt[2]) HIP_TRY(hipMemcpyAsync(hard.data(), s->d_zone_u64.p + 2 * cap, cnt[2] * 8ull, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::sort(starts.begin(), starts.end());
  std::sort(ends.begin(), ends.end());
  const size_t n_runs = starts.size();
  std::vector<char> run_hard(n_runs, 0);
  for (unsigned long long h : hard) {
    const size_t r = (size_t)(std::upper_bound(starts.begin(), starts.end(), h) - starts.begin()) - 1;
    run_hard[r] = 1;
  }
  // ---- where the zones are responsible (end positions), what they keep, what they copy ----
  struct Keep { uint64_t lo, hi; bool cont; };
  std::vector<Keep> keep;
  std::vector<unsigned long long> excl;  // pairs
  auto add_keep = [&](uint64_t lo, uint64_t hi, bool cont) {
    if (!keep.empty() && !keep.back().cont && lo <= keep.back().hi + 1) {
      keep.back().hi = std::max(keep.back().hi, hi);
      keep.back().cont = cont;
    } else {
      keep.push_back(Keep{lo, hi, cont});
    }
  };
  for (size_t r = 0; r < n_runs; ++r) {
    const uint64_t rs = starts[r], re = ends[r];
    const uint64_t lo = rs + 1, hi = std::min<uint64_t>(n, re + C);
    if (!excl.empty() && lo <= excl.back() + 1) excl.back() = std::max<unsigned long long>(excl.back(), hi);
    else { excl.push_back(lo); excl.push_back(hi); }
    // A run of full wildcards longer than the pattern: from end position rs + m (the last m characters are wildcards)
    // to re every pattern's cost is one constant -- the list holds the way down to it (kept up to rs + m, marked) and
    // picks up at re, the last position of the stretch.
    // (the way down may be longer when the interval in front reaches into this run: it is merged with it)
    uint64_t left_hi = rs + e->plen;
    if (!keep.empty() && !keep.back().cont && lo <= keep.back().hi + 1) left_hi = std::max(left_hi, keep.back().hi);
    const bool cut = !all && !run_hard[r] && re - rs >= (uint64_t)e->plen + 2 && left_hi + 1 < re;
    if (cut) {
      add_keep(lo, left_hi, true);
      add_keep(re, hi, false);
    } else {
      add_keep(lo, hi, false);
    }
  }
  const size_t n_zones = keep.size();
  std::vector<unsigned long long> tab;  // zones (6 words each), then the segments (4 words each), then excl
  tab.reserve(10 * n_zones + excl.size());
  uint64_t Z = C + 1;
  for (const Keep& kp : keep) {
    // (m + k characters of context: the scan starts fresh behind the separator, exact from the first kept position on)
    const uint64_t a = kp.lo - 1 > C ? kp.lo - 1 - C : 0;
    tab.insert(tab.end(), {(unsigned long long)Z, (unsigned long long)a, (unsigned long long)kp.lo, (unsigned long long)kp.hi,
                           kp.cont ? 1ull : 0ull, 0ull});
    Z += (kp.hi - a) + C + 1;
  }
  // (the tiled scan of the zones at 3.8e10 character x group of 64 patterns per second: at most ~0.2 s of it)
  if (Z > (1ull << 30) || (double)Z * (double)((e->patterns.size() + 63) / 64) > 8e9) return 0;
  const size_t seg_at = tab.size();
  for (size_t z = 0; z < n_zones; ++z)
    tab.insert(tab.end(), {tab[6 * z + 1], tab[6 * z], tab[6 * z + 3] - tab[6 * z + 1], 0ull});
  const size_t excl_at = tab.size();
  tab.insert(tab.end(), excl.begin(), excl.end());
  if (int rc = s->d_zone_tab.reserve(tab.size() + 8)) return rc;
  if (int rc = s->d_zone_text.reserve(Z + 128)) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_zone_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(s->d_zone_text.p, 'X', Z + 64, st));
  le = launch_gather_zones(tptr, s->d_zone_text.p, s->d_zone_tab.p + seg_at, (uint32_t)n_zones, st);
  if (le != hipSuccess) return hip_fail(le, "zone gather launch");
  // (one bit per end position of the zone buffer: only the positions a zone is responsible for are listed)
  std::vector<uint32_t> bits((size_t)(Z + 64) / 32 + 2, 0u);
  for (size_t z = 0; z < n_zones; ++z) {
    const uint64_t q0 = tab[6 * z] + (tab[6 * z + 2] - tab[6 * z + 1]), q1 = tab[6 * z] + (tab[6 * z + 3] - tab[6 * z + 1]);
    for (uint64_t q = q0; q <= q1; ++q) bits[q >> 5] |= 1u << (q & 31);
  }
  if (int rc = s->d_seed_packed.reserve(bits.size())) return rc;  // (free here: the seeded pass is over)
  HIP_TRY(hipMemcpyAsync(s->d_seed_packed.p, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, st));
  // ---- the neighbourhoods through the pattern-tiled scan ----
  uint32_t zc = 0;
  uint64_t waves = 0;
  bool zok = n_zones == 0;
  if (n_zones)
    if (int rc = tiled_scan_list(s, e, s->d_zone_text.p, Z, k, 16u, s->d_zone_peq, s->d_zone_list, d_cnt + 4, &zc, &zok, &waves,
                                 s->d_seed_packed.p))
      return rc;
  if (!zok) return 0;
  // ---- the seeded pass's records outside the zones' intervals, then the zones' records behind them ----
  const uint32_t have = *list_count;
  uint32_t kept = 0;
  if (have) {
    if (int rc = L.d_sorted.reserve(have)) return rc;
    if (int rc = L.d_sort.reserve(select_scratch_bytes(have))) return rc;
    const size_t flag_bytes = ((size_t)have + 255) / 256 * 256;
    unsigned char* d_keep = L.d_sort.p;
    le = launch_drop_excluded(s->d_tiled_list.p, have, s->d_zone_tab.p + excl_at, (uint32_t)(excl.size() / 2), d_keep, st);
    if (le != hipSuccess) return hip_fail(le, "record filter launch");
    le = launch_compact_candidates(s->d_tiled_list.p, have, d_keep, L.d_sorted.p, d_cnt + 5, L.d_sort.p + flag_bytes,
                                   L.d_sort.cap - flag_bytes, st);
    if (le != hipSuccess) return hip_fail(le, "record compaction launch");
    HIP_TRY(hipMemcpyAsync(&kept, d_cnt + 5, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  } else {
    HIP_TRY(hipStreamSynchronize(st));  // (`tab` goes out of scope)
  }
  if ((uint64_t)kept + zc > (1ull << 28) + (1ull << 27)) return 0;
  if (int rc = s->d_tiled_list.reserve((size_t)kept + zc + 1024)) return rc;  // (may move the buffer: its records are in d_sorted)
  if (kept) HIP_TRY(hipMemcpyAsync(s->d_tiled_list.p, L.d_sorted.p, (size_t)kept * sizeof(Candidate), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_cnt + 5, &kept, 4, hipMemcpyHostToDevice, st));
  le = launch_map_zone_list(s->d_zone_list.p, zc, s->d_zone_tab.p, (uint32_t)n_zones, s->d_tiled_list.p, d_cnt + 5,
                            (uint32_t)std::min<size_t>(s->d_tiled_list.cap, 0xFFFFFFFFu), st);
  if (le != hipSuccess) return hip_fail(le, "zone record mapping launch");
  uint32_t total = kept;
  HIP_TRY(hipMemcpyAsync(&total, d_cnt + 5, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *list_count = total;
  s->stats.cond_resolved += n_zones;  // (here: neighbourhoods of other letters that went through the tiled scan)
  *ok = true;
  return 0;
}

// An overhang batch through the seeded search (search_many_pertext): the seeded pass has listed every end position with
// cost <= k of the buffer as if there were no overhang.  Kept: the INSIDE of the texts -- end positions (m + k, len], which
// no alignment that reaches a text's first column can end in, and behind which the virtual columns lie.  Added: what
// overhang changes -- [0, m + k] from the overhang column, (len, len + steps] -- from tiled_pertext_kernel's edge segments.
static int seeded_overhang_edges(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* buf, uint64_t total, uint32_t k,
                                 const TextTable& tt, const TiledPerText& ov, uint32_t* list_count, bool* ok) {
  *ok = false;
  ScanLane& L = s->lanes[0];
  hipStream_t st = s->stream;
  uint32_t* d_cnt = s->d_tiled_cnt.p;
  const uint32_t have = *list_count;
  uint32_t kept = 0;
  if (have) {
    if (int rc = L.d_sorted.reserve(have)) return rc;
    if (int rc = L.d_sort.reserve(select_scratch_bytes(have))) return rc;
    const size_t flag_bytes = ((size_t)have + 255) / 256 * 256;
    unsigned char* d_keep = L.d_sort.p;
    hipError_t le = launch_keep_interior(s->d_tiled_list.p, have, tt, ov.edge_cols, d_keep, st);
    if (le != hipSuccess) return hip_fail(le, "record filter launch");
    le = launch_compact_candidates(s->d_tiled_list.p, have, d_keep, L.d_sorted.p, d_cnt + 5, L.d_sort.p + flag_bytes,
                                   L.d_sort.cap - flag_bytes, st);
    if (le != hipSuccess) return hip_fail(le, "record compaction launch");
    HIP_TRY(hipMemcpyAsync(&kept, d_cnt + 5, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  uint32_t zc = 0;
  uint64_t waves = 0;
  bool zok = false;
  if (int rc = tiled_scan_list(s, e, buf, total, k, 16u, s->d_zone_peq, s->d_zone_list, d_cnt + 4, &zc, &zok, &waves, nullptr, &ov)) return rc;
  if (!zok) return 0;
  if ((uint64_t)kept + zc > (1ull << 28)) return 0;
  if (int rc = s->d_tiled_list.reserve((size_t)kept + zc + 1024)) return rc;  // (may move the buffer: its records are in d_sorted)
  if (kept) HIP_TRY(hipMemcpyAsync(s->d_tiled_list.p, L.d_sorted.p, (size_t)kept * sizeof(Candidate), hipMemcpyDeviceToDevice, st));
  if (zc) HIP_TRY(hipMemcpyAsync(s->d_tiled_list.p + kept, s->d_zone_list.p, (size_t)zc * sizeof(Candidate), hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  *list_count = kept + zc;
  *ok = true;
  return 0;
}

// The seeds of the seeded search (search_encoded_seeded): k+1 DISJOINT pieces of the pattern's rows -- all the pigeonhole
// argument needs, not a cover -- as (end row, length <= kSeedMaxLen), of at most two lengths (the two tables').
// The even cut: k+1 pieces, the first m mod (k+1) one row longer; a seed is the last <= kSeedMaxLen rows of a piece.
// Patterns with ambiguity letters (an Iupac searcher): a seed over such a letter stands for several strings -- the NGG
// of a CRISPR guide makes the last of the four pieces of a 23-mer hit four times as often as the others.  So the
// places are chosen so that the expected number of table hits per text position is smallest -- a small dynamic
// programme over the rows, the mean over up to 512 patterns -- and that layout is taken when it beats the even cut by
// 5 % (plain patterns keep the even cut).  SASSY_HIP_SEED_LAYOUT=0: the even cut.  Pure host arithmetic
// (sassy_hip_seed_layout; tests/test_cabi_symbols.py).
static void seed_layout(bool by_hits, int profile, const uint8_t* const* patterns, size_t npat, uint32_t m, uint32_t k, uint32_t* p_end,
                        uint32_t* p_len) {
  const uint32_t pieces = k + 1, q = m / pieces, spare = m - q * pieces;
  for (uint32_t pc = 0; pc < pieces; ++pc) {
    const uint32_t len = q + (pc < spare ? 1u : 0u);
    p_end[pc] = pc * q + std::min(pc, spare) + len;
    p_len[pc] = std::min(len, kSeedMaxLen);
  }
  if (profile != PROFILE_IUPAC || !by_hits || npat == 0) return;  // (by_hits = false, switch seed_layout = 0: the even cut)
  const size_t sample = std::min<size_t>(npat, 512);
  // rate[a][L] = mean over the sampled patterns of the probability that a random L-gram matches rows [a, a + L)
  std::vector<std::vector<double>> rate(m + 1, std::vector<double>(kSeedMaxLen + 1, 0.0));
  for (size_t p = 0; p < sample; ++p) {
    const uint8_t* pt = patterns[p * (npat / sample)];
    for (uint32_t a = 0; a < m; ++a) {
      double pr = 1.0;
      for (uint32_t L = 1; L <= kSeedMaxLen && a + L <= m; ++L) {
        pr *= (double)__builtin_popcount(iupac_code(pt[a + L - 1]) & 0x0Fu) / 4.0;
        rate[a][L] += pr / (double)sample;
      }
    }
  }
  double even = 0;
  for (uint32_t pc = 0; pc < pieces; ++pc) even += rate[p_end[pc] - p_len[pc]][p_len[pc]];
  double best = even * 0.95;
  uint32_t best_end[8], best_len[8];
  bool found = false;
  for (uint32_t La = 3; La <= kSeedMaxLen; ++La)
    for (uint32_t Lb = La; Lb <= std::min<uint32_t>(kSeedMaxLen, La + 2); ++Lb) {
      if ((uint64_t)La * pieces > m) continue;
      // f[j][i] = least total rate of j pieces within rows [0, i); from[j][i] = the length of the piece that ends at i (0: none)
      const double inf = 1e300;
      std::vector<std::vector<double>> f(pieces + 1, std::vector<double>(m + 1, inf));
      std::vector<std::vector<uint32_t>> from(pieces + 1, std::vector<uint32_t>(m + 1, 0u));
      for (uint32_t i = 0; i <= m; ++i) f[0][i] = 0;
      for (uint32_t j = 1; j <= pieces; ++j)
        for (uint32_t i = 1; i <= m; ++i) {
          f[j][i] = f[j][i - 1];
          from[j][i] = 0;
          for (uint32_t L : {La, Lb})
            if (i >= L && f[j - 1][i - L] < inf && f[j - 1][i - L] + rate[i - L][L] < f[j][i]) {
              f[j][i] = f[j - 1][i - L] + rate[i - L][L];
              from[j][i] = L;
            }
        }
      if (f[pieces][m] >= best) continue;
      best = f[pieces][m];
      found = true;
      uint32_t i = m;
      for (uint32_t j = pieces; j >= 1; --j) {
        while (from[j][i] == 0) --i;
        best_end[j - 1] = i;
        best_len[j - 1] = from[j][i];
        i -= from[j][i];
      }
    }
  if (found)
    for (uint32_t pc = 0; pc < pieces; ++pc) {
      p_end[pc] = best_end[pc];
      p_len[pc] = best_len[pc];
    }
}

// The rows of the sub-piece test in front of the seeded search's verification (common.h: SeedParams::sub) for the seeds
// (p_end, p_len).  For a hit of piece p: k+1 disjoint sub-pieces of the rows within `reach` of the seed, shared out between
// the two sides in proportion to the rows there; one of them must be intact within k characters of the seed's diagonal.
// The test reads ONE window of the 2-bit text for all pieces (seed_kernels.hip: test_issue): it starts *win_left =
// (longest seed) + (most rows used left of a seed) + k characters in front of the seed's end, and every sub-piece must
// start, at its leftmost shift, within 48 characters of that -- the largest reach <= 24 - k that allows it.
// sub[8 p + u] = 2a | (32 - 2 len) << 8 | 2 (off & 15) << 16 | (off >> 4) << 24 for sub-piece u = rows [a, a + len) of
// piece p, off = characters from the window's start to where it lies at its leftmost shift; len is capped so that
// (off & 15) + 2k + len <= 32: the compared bits lie in the 64 the test takes from the window.  sub[8 p] = 0xFF: no
// test for piece p (fewer rows around it than sub-pieces).  *max_off <= 31: four dwords of text suffice (the narrow
// layout).  Pure host arithmetic (sassy_hip_seed_test_rows; tests/test_cabi_symbols.py).
static void seed_test_rows(uint32_t m, uint32_t k, const uint32_t* p_end, const uint32_t* p_len, uint32_t* sub, uint32_t* win_left,
                           uint32_t* max_off) {
  const uint32_t pieces = k + 1;
  struct SubPiece { uint32_t pc, u, a, len, off; };
  std::vector<SubPiece> subs;
  *win_left = 0;
  *max_off = 0;
  for (uint32_t reach = 24 - k; reach >= 4; --reach) {  // (k <= 7)
    subs.clear();
    uint32_t max_nl = 0, max_len = 0;
    for (uint32_t pc = 0; pc < pieces; ++pc) {
      const uint32_t sp = p_end[pc] - p_len[pc], pe = p_end[pc];
      const uint32_t nl = std::min(sp, reach), nr = std::min(m - pe, reach);
      max_len = std::max(max_len, p_len[pc]);
      if (nl + nr < pieces) continue;  // fewer rows than sub-pieces: no test for this piece
      max_nl = std::max(max_nl, nl);
      uint32_t cl = (uint32_t)(((uint64_t)pieces * nl + (nl + nr) / 2) / (nl + nr));
      cl = std::min(cl, nl);
      uint32_t cr = pieces - cl;
      if (cr > nr) { cr = nr; cl = pieces - cr; }
      uint32_t u = 0;
      for (uint32_t x = 0; x < cl; ++x) {  // left of the seed: rows [sp - nl, sp) in cl parts
        const uint32_t a = sp - nl + (uint32_t)((uint64_t)nl * x / cl), b = sp - nl + (uint32_t)((uint64_t)nl * (x + 1) / cl);
        subs.push_back({pc, u++, a, b - a, sp - a});  // (off: for now the rows from a to the seed's start)
      }
      for (uint32_t x = 0; x < cr; ++x) {  // right of it: rows [pe, pe + nr) in cr parts
        const uint32_t a = pe + (uint32_t)((uint64_t)nr * x / cr), b = pe + (uint32_t)((uint64_t)nr * (x + 1) / cr);
        subs.push_back({pc, u++, a, b - a, 0x80000000u | (a - pe)});  // (rows from the seed's end to a)
      }
    }
    *win_left = max_len + max_nl + k;
    *max_off = 0;
    for (SubPiece& q : subs) {
      q.off = (q.off & 0x80000000u) ? *win_left + (q.off & 0x7FFFFFFFu) - k : *win_left - p_len[q.pc] - q.off - k;
      *max_off = std::max(*max_off, q.off);
    }
    if (*max_off <= 47) break;
    subs.clear();
  }
  for (int i = 0; i < 64; ++i) sub[i] = 0xFFu;  // (low byte 0xFF in a piece's first entry: no test for that piece)
  for (const SubPiece& q : subs) {
    const uint32_t len = std::min(q.len, std::min(16u, 17u - 2u * k));
    sub[8 * q.pc + q.u] = (2 * q.a) | ((32 - 2 * len) << 8) | ((2 * (q.off & 15u)) << 16) | ((q.off >> 4) << 24);
  }
}

// search_encoded_patterns for many patterns over a long text: seed -> verify -> report (seed_kernels.hip).  One
// pass over the text -- one launch -- looks every L-gram up in a table of all patterns' pigeonhole pieces; one lane
// per hit runs the pattern over the few dozen characters around it.  Dna codes only (the caller has checked the text is plain ACGT
// when the searcher is Iupac).  *done = false: not this shape after all (lists too large) -- the caller falls back.
int search_encoded_seeded(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr,
                                 const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all, bool wo,
                                 sassy_hip_Result* R, bool* done, const TextTable* tt,
                                 const HostTexts* ht, bool dirty_text, ManyDefer* defer,
                                 const TiledPerText* ov) {
  *done = false;
  hipStream_t st = s->stream;
  const size_t npat = e->patterns.size();
  const uint32_t m = (uint32_t)e->plen;
  std::string err;
  PatternPlan plan0;
  for (size_t p = 0; p < npat; ++p) {
    PatternPlan pl;
    if (!make_plan(s->profile, e->patterns[p].data(), m, p == 0 ? plan0 : pl, err)) return fail(SASSY_HIP_EINVAL, err);
  }
  // ---- the seeds: k+1 disjoint pieces, at most two lengths (seed_layout) ----
  const uint32_t pieces = k + 1;
  uint32_t p_end[8], p_len[8], tab_of[8], tab_len[2] = {0, 0};
  {
    std::vector<const uint8_t*> rows(npat);
    for (size_t p = 0; p < npat; ++p) rows[p] = e->patterns[p].data();
    seed_layout(s->sw.seed_layout != 0, s->profile, rows.data(), npat, m, k, p_end, p_len);
  }
  for (uint32_t pc = 0; pc < pieces; ++pc) {
    if (tab_len[0] == 0 || tab_len[0] == p_len[pc]) { tab_len[0] = p_len[pc]; tab_of[pc] = 0; }
    else { tab_len[1] = p_len[pc]; tab_of[pc] = 1; }
  }
  uint32_t seed_bits_off[2] = {0, 0};
  // ---- direct-address tables: code of a seed = sum of its characters' Dna codes, first character lowest ----
  // Iupac searcher (plain-ACGT text, patterns with ambiguity letters -- a CRISPR guide with its NGG): a seed with such
  // letters stands for every concrete string it matches and gets one table entry per string (A, C, T, G = codes
  // 0..3 = bits 0..3 of the letter's base set).  More than kSeedMaxExpand strings in one seed: not this path.
  const bool iupac_pats = s->profile == PROFILE_IUPAC;
  auto base_set = [&](uint8_t c) -> uint32_t { return iupac_pats ? (uint32_t)(iupac_code(c) & 0x0Fu) : 1u << ((c >> 1) & 3u); };
  constexpr size_t kSeedMaxExpand = 256;
  std::vector<uint32_t> start[2], entries[2];
  for (int t = 0; t < 2; ++t) {
    if (!tab_len[t]) continue;
    const size_t size = (size_t)1 << (2 * tab_len[t]);
    start[t].assign(size + 1, 0u);
    std::vector<std::pair<uint32_t, uint32_t>> code_entry;  // (code, (pattern << 3) | piece)
    code_entry.reserve(npat * pieces);
    std::vector<uint32_t> codes, next;
    for (size_t p = 0; p < npat; ++p)
      for (uint32_t pc = 0; pc < pieces; ++pc) {
        if (tab_of[pc] != (uint32_t)t) continue;
        const uint8_t* src = e->patterns[p].data() + p_end[pc] - p_len[pc];
        {  // the common case, every letter one base: one code, no lists
          uint32_t code = 0;
          bool concrete = true;
          for (uint32_t x = 0; x < p_len[pc] && concrete; ++x) {
            const uint32_t set = base_set(src[x]);
            concrete = set && !(set & (set - 1));
            code |= (set == 1 ? 0u : set == 2 ? 1u : set == 4 ? 2u : 3u) << (2 * x);
          }
          if (concrete) {
            code_entry.emplace_back(code, (uint32_t)(p << 3) | pc);
            start[t][code + 1]++;
            continue;
          }
        }
        codes.assign(1, 0u);
        for (uint32_t x = 0; x < p_len[pc]; ++x) {
          const uint32_t set = base_set(src[x]);
          if (set == 0) { codes.clear(); break; }  // (X: matches nothing -- the piece is never intact)
          next.clear();
          for (uint32_t c : codes)
            for (uint32_t b = 0; b < 4; ++b)
              if (set & (1u << b)) next.push_back(c | (b << (2 * x)));
          if (next.size() > kSeedMaxExpand) return 0;  // *done stays false
          codes.swap(next);
        }
        for (uint32_t c : codes) {
          code_entry.emplace_back(c, (uint32_t)(p << 3) | pc);
          start[t][c + 1]++;
        }
      }
    for (size_t c = 0; c < size; ++c) start[t][c + 1] += start[t][c];
    entries[t].resize(code_entry.size());
    std::vector<uint32_t> cursor(start[t].begin(), start[t].end() - 1);
    for (const auto& ce : code_entry) entries[t][cursor[ce.first]++] = ce.second;
  }
  // ---- match masks per Dna code and the patterns' bytes (traceback) ----
  const bool wide = m > 32;
  std::vector<unsigned long long> peq(npat * 4, 0ull);
  std::vector<uint8_t> flat(npat * (size_t)m);
  for (size_t p = 0; p < npat; ++p) {
    const uint8_t* pt = e->patterns[p].data();
    memcpy(&flat[p * m], pt, m);
    uint32_t* peq32 = reinterpret_cast<uint32_t*>(peq.data());
    for (uint32_t j = 0; j < m; ++j) {
      if (!iupac_pats) {  // one base per letter: its Dna code
        const uint32_t c = (pt[j] >> 1) & 3u;
        if (wide) peq[p * 4 + c] |= 1ull << j;
        else peq32[p * 4 + c] |= 1u << j;
        continue;
      }
      const uint32_t set = base_set(pt[j]);
      for (uint32_t c = 0; c < 4; ++c) {
        if (!(set & (1u << c))) continue;
        if (wide) peq[p * 4 + c] |= 1ull << j;
        else peq32[p * 4 + c] |= 1u << j;
      }
    }
  }
  if (int rc = s->d_tiled_peq.reserve(peq.size())) return rc;
  if (int rc = s->d_tiled_pat.reserve(flat.size() + 64)) return rc;
  if (int rc = s->d_tiled_cnt.reserve(16)) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_tiled_peq.p, peq.data(), (wide ? 8 : 4) * 4 * npat, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->d_tiled_pat.p, flat.data(), flat.size(), hipMemcpyHostToDevice, st));
  {  // one bit per min(len, 8)-gram a seed of the table ends with (staged in LDS by the kernel)
    std::vector<uint32_t> bits;
    uint32_t off[2] = {0, 0};
    for (int t = 0; t < 2; ++t) {
      off[t] = (uint32_t)bits.size();
      if (!tab_len[t]) continue;
      const uint32_t l8 = std::min(tab_len[t], 8u), cut = 2 * (tab_len[t] - l8);
      bits.resize(bits.size() + std::max<size_t>(1, ((size_t)1 << (2 * l8)) / 32), 0u);
      for (size_t c = 0; c + 1 < start[t].size(); ++c)
        if (start[t][c + 1] != start[t][c]) bits[off[t] + ((c >> cut) >> 5)] |= 1u << ((c >> cut) & 31);
    }
    if (int rc = s->d_seed_bits.reserve(bits.size())) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_seed_bits.p, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // (`bits` goes out of scope)
    seed_bits_off[0] = off[0];
    seed_bits_off[1] = off[1];
  }
  for (int t = 0; t < 2; ++t) {
    if (!tab_len[t]) continue;
    if (int rc = s->d_seed_start[t].reserve(start[t].size())) return rc;
    if (int rc = s->d_seed_entries[t].reserve(entries[t].size() + 1)) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_seed_start[t].p, start[t].data(), start[t].size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->d_seed_entries[t].p, entries[t].data(), entries[t].size() * 4, hipMemcpyHostToDevice, st));
  }
  SeedParams SP{};
  SP.text = tptr;
  SP.text_len = text_len;
  for (int t = 0; t < 2; ++t) {
    SP.len[t] = tab_len[t];
    SP.start[t] = s->d_seed_start[t].p;
    SP.entries[t] = s->d_seed_entries[t].p;
  }
  SP.peq = s->d_tiled_peq.p;
  SP.m = m;
  SP.k = k;
  for (uint32_t pc = 0; pc < pieces; ++pc) {
    SP.rem_packed |= (uint64_t)(m - p_end[pc]) << (8 * pc);
  }
  // ---- the sub-piece test in front of the verification (seed_test_rows; patterns of <= 32 rows) ----
  const bool env_sub = s->sw.seed_subtest != 0, env_narrow = s->sw.seed_narrow != 0, env_pos64 = s->sw.seed_pos64 != 0;
  if (!wide && env_sub) {
    uint32_t sub[64], win_left = 0, max_off = 0;
    seed_test_rows(m, k, p_end, p_len, sub, &win_left, &max_off);
    // (an Iupac searcher whose patterns are all plain bases -- 10 000 random 20-mers -- needs no care words; the
    // kernel for positions beyond 32 bits always reads them)
    const bool pos64 = env_pos64 || text_len >= 0xFFFF0000ull;
    bool care_words = pos64;
    if (iupac_pats)
      for (size_t p = 0; p < npat && !care_words; ++p)
        for (uint32_t j = 0; j < m; ++j) {
          const uint32_t set = base_set(e->patterns[p][j]);
          if (!set || (set & (set - 1))) { care_words = true; break; }
        }
    const bool narrow = env_narrow && !care_words && max_off <= 31;
    // the table entries with their patterns' packed rows: row j at bits 2j; with care words a second pair says which
    // rows the test may compare (11: a concrete base, 00: a letter that stands for several -- such a row matches
    // any character here)
    std::vector<unsigned long long> ppk(npat * 2, 0ull);
    for (size_t p = 0; p < npat; ++p)
      for (uint32_t j = 0; j < m; ++j) {
        if (!iupac_pats) {
          ppk[2 * p] |= (unsigned long long)((e->patterns[p][j] >> 1) & 3u) << (2 * j);
          ppk[2 * p + 1] |= 3ull << (2 * j);
          continue;
        }
        const uint32_t set = base_set(e->patterns[p][j]);
        const bool one = set && !(set & (set - 1));
        const uint32_t code = one ? (set == 1 ? 0u : set == 2 ? 1u : set == 4 ? 2u : 3u) : 0u;
        ppk[2 * p] |= (unsigned long long)code << (2 * j);
        if (one) ppk[2 * p + 1] |= 3ull << (2 * j);
      }
    std::vector<uint32_t> e16;
    e16.reserve((care_words ? 8 : 4) * (entries[0].size() + entries[1].size()));
    for (int t = 0; t < 2; ++t)
      for (uint32_t en : entries[t]) {
        const unsigned long long rows = ppk[2 * (en >> 3)], care = ppk[2 * (en >> 3) + 1];
        e16.push_back(en); e16.push_back((uint32_t)rows); e16.push_back((uint32_t)(rows >> 32)); e16.push_back(0u);
        if (care_words) { e16.push_back((uint32_t)care); e16.push_back((uint32_t)(care >> 32)); e16.push_back(0u); e16.push_back(0u); }
      }
    const uint64_t n16 = (text_len + 15) / 16;
    if (int rc = s->d_seed_sub.reserve(64)) return rc;
    if (int rc = s->d_seed_e16.reserve(e16.size() + 8)) return rc;
    if (int rc = s->d_seed_packed.reserve(n16 + 8)) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_seed_sub.p, sub, 64 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->d_seed_e16.p, e16.data(), e16.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(s->d_seed_packed.p + n16, 0, 8 * 4, st));
    hipError_t pe_ = launch_pack_text(tptr, text_len, s->d_seed_packed.p, st);
    if (pe_ != hipSuccess) return hip_fail(pe_, "text packing launch");
    HIP_TRY(hipStreamSynchronize(st));  // (`sub`, `e16` go out of scope)
    SP.sub = s->d_seed_sub.p;
    SP.packed_text = s->d_seed_packed.p;
    SP.pat_care = care_words ? 1u : 0u;
    SP.entries16 = reinterpret_cast<const uint4*>(s->d_seed_e16.p);
    SP.entries16_off1 = (uint32_t)entries[0].size();
    SP.win_left = win_left;
    SP.win_dwords = narrow ? 4u : 5u;
    SP.pos64 = pos64 ? 1u : 0u;
  }
  SP.out_count = s->d_tiled_cnt.p;
  SP.hit_count = reinterpret_cast<unsigned long long*>(s->d_tiled_cnt.p + 4);
  SP.seed_bits = s->d_seed_bits.p;
  SP.bits_off[0] = seed_bits_off[0];
  SP.bits_off[1] = seed_bits_off[1];
  SP.separators = tt ? 1u : 0u;  // several texts in the buffer: 'X' between them
  // 2 KiB of text per wave and step; contiguous runs per wave
  // (65 536 waves: 13 rounds of the chip's 5 120 resident waves -- the last round's ragged end is 4 % of config 4 with 16 384;
  // a smaller text: ten steps per wave, at least 4 096 waves -- a workgroup stages 16 KiB of seed bits before its first step:
  // 330 MB of reads in 65 536 waves of 2.5 steps were 3.9 ms for both strands, 2.3 in 16 384)
  const uint64_t steps_total = std::max<uint64_t>(1, (text_len + 2047) / 2048);
  const uint64_t waves = std::min<uint64_t>(std::max<uint64_t>(4096, std::min<uint64_t>(65536, steps_total / 10)), steps_total);
  const uint32_t grid = (uint32_t)((waves + kWavesPerGroup - 1) / kWavesPerGroup);

  const uint64_t kMaxList = 1ull << 26;
  uint32_t out_count = 0;
  unsigned long long n_hits = 0, n_pass = 0;
  for (int attempt = 0;; ++attempt) {
    if (int rc = s->d_tiled_list.reserve(std::max<size_t>((size_t)1 << 18, (size_t)out_count + 1024))) return rc;
    SP.out = s->d_tiled_list.p;
    SP.out_cap = (uint32_t)std::min<size_t>(s->d_tiled_list.cap, 0xFFFFFFFFu);
    SP.out_stop = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(SP.out_cap, kMaxList) + (1u << 20), 0xF0000000ull);
    HIP_TRY(hipMemsetAsync(s->d_tiled_cnt.p, 0, 64, st));
    HIP_TRY(hipEventRecord(s->ev_a_multi(), st));
    hipError_t le = launch_seed_search(SP, grid, st);
    if (le != hipSuccess) return hip_fail(le, "seeded search launch");
    HIP_TRY(hipEventRecord(s->ev_multi, st));
    uint32_t ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(ctl, s->d_tiled_cnt.p, sizeof ctl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    out_count = ctl[0];
    memcpy(&n_hits, ctl + 4, 8);
    memcpy(&n_pass, ctl + 6, 8);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_a_multi(), s->ev_multi));
    s->stats.scan_ms += ms;
    s->stats.scan_launches += 1;
    if (out_count <= SP.out_cap) break;
    if (out_count > kMaxList || attempt == 2) return 0;  // *done stays false
  }
  if (dirty_text) {  // other letters than ACGT in the text: their neighbourhoods come from the pattern-tiled scan
    bool zones_ok = false;
    if (int rc = seeded_dirty_zones(s, e, tptr, text_len, k, all, &out_count, &zones_ok)) return rc;
    if (!zones_ok) return 0;  // *done stays false
  }
  if (ov) {  // an overhang batch: the texts' edges come from the per-text tiled scan
    if (!tt) return fail(SASSY_HIP_EINVAL, "internal: overhang edges need the batch's text table");
    bool edges_ok = false;
    if (int rc = seeded_overhang_edges(s, e, tptr, text_len, k, *tt, *ov, &out_count, &edges_ok)) return rc;
    if (!edges_ok) return 0;  // *done stays false
  }
  s->stats.text_bytes += text_len;
  s->stats.chunks += waves;
  s->stats.hit_blocks += n_hits;   // (here: table hits ...
  s->stats.live_blocks += n_pass;  //  ... and how many of them passed the sub-piece test)
  s->stats.piece_len = tab_len[0];
  s->stats.filtered = 6;
  s->stats.candidates += out_count;
  *done = true;
  if (out_count == 0) return 0;
  return finish_pattern_list(s, e, plan0, tptr, h_text, text_len, k, all, wo, out_count, true, R, tt, ht, defer);
}

// ---- many host texts: one buffer, one scan per pattern and strand ----
// search_texts / search_many are meant for many short texts (reads).  Running the whole kernel
// pipeline once per (pattern, text) pair would be launch-latency bound (~70 us per pair), so the
// texts of a batch are laid out in ONE device buffer, separated by m+k+1 (or more) 'X' -- the Iupac
// letter that matches nothing -- and every pattern is scanned over that buffer once per strand.
// Exactness ("batched texts"): after m+k+1 non-matching characters every DP column
// equals the text-start column D[j][0] = j, so a text's cells do not depend on its predecessors;
// costs never decrease across a separator, so the only report that can fall into one is the right
// end of a plateau that reached the text's end -- the single-text search reports exactly that one
// at the text end (end-of-text rule); rank_scatter_kernel moves it there (search_all: drops it), and
// the traceback windows are clipped at the text's own start.  A Dna searcher runs these scans with
// the Iupac kernels, which give identical results on ACGT text (other Dna text is outside the
// reference's contract, src/profiles/dna.rs:60-75 valid_seq), and falls back to the pair loop otherwise.
bool acgt_only(const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const uint8_t c = p[i] & (uint8_t)~0x20;
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false;
  }
  return true;
}

// Patterns of one length (<= 64 rows) can take the pattern-tiled scan over the batch instead of one kernel chain per
// pattern and strand; many_tiled_wanted() is the shared estimate.  tiled_only: called ahead of search_many_pertext
// for exactly that -- if the tiled scan does not take the batch after all, nothing is done here (handled = false).
// Expected table hits of the seeded search per (character, pattern) on random text; 0: the shape does not allow it.
double seeded_hit_rate(size_t m, size_t k) {
  if (k + 1 > 8 || m / (k + 1) < 5 || m + 3 * k + 1 > 4 * (size_t)kSeedWindowDwords) return 0.0;
  double rate = 0;
  for (size_t pc = 0; pc < k + 1; ++pc)
    rate += std::pow(0.25, (double)std::min<size_t>(m / (k + 1) + (pc < m % (k + 1) ? 1 : 0), kSeedMaxLen));
  return rate;
}
// The seeded search's estimate (search_encoded_seeded): ~0.3 ms of tables and launches, the seed pass at ~3e11 B/s
// whatever the number of patterns (tools/bench_encoded.py: 64 patterns over 256 MB in 1.2 ms), ~8 ps per table hit
// with the sub-piece test (patterns of <= 32 rows), ~16 ps when every hit is verified (tools/bench_configs.py,
// config 4: 1.1e10 hits, 95 / 197 ms).
double seeded_estimate(size_t m, size_t k, size_t n_patterns, uint64_t text_len) {
  const double hits = seeded_hit_rate(m, k) * (double)text_len * (double)n_patterns;
  return 3e-4 + (double)text_len / 3e11 + hits * (m <= 32 ? 8e-12 : 16e-12);
}

bool many_tiled_wanted(const sassy_SearcherType* s, const size_t* pattern_lens, size_t n_patterns, uint64_t total, size_t k) {
  if (n_patterns == 0 || pattern_lens[0] > 64 || 2 * k + 3 > 64 || n_patterns >= (1u << 24)) return false;
  for (size_t pi = 1; pi < n_patterns; ++pi)
    if (pattern_lens[pi] != pattern_lens[0]) return false;
  const int env_many = (int)s->sw.many_tiled;
  if (env_many >= 0) return env_many != 0;
  if (seeded_hit_rate(pattern_lens[0], k) > 0 &&
      (s->rc ? 2.0 : 1.0) * seeded_estimate(pattern_lens[0], k, n_patterns, total) <
          (s->rc ? 2.0 : 1.0) * (double)n_patterns * (25e-6 + 5.3e-13 * (double)total))
    return true;  // (the one-pass branch decides between the seeded search and the tiled scan once the batch is on the device)
  // Measured with tools/bench_reads.py (96 barcodes of 24 rows, k = 3, both strands, 100 / 330 MB of 1 kb reads): the
  // tiled scan advances 2.8e10 (character x group of 64 patterns) per second here (16 Iupac classes, the last group
  // half empty): 14 / 46 ms; the 192 chains take 15 / 38 ms = 25 us + 5.3e-13 s per byte of the batch each.
  const double strands = s->rc ? 2.0 : 1.0;
  const double est_tiled = strands * ((double)total * (double)((n_patterns + 63) / 64) / 2.8e10 + 1.5e-4);
  const double est_chains = strands * (double)n_patterns * (25e-6 + 5.3e-13 * (double)total);
  return est_tiled < est_chains;
}

// Both strands' passes over one batch have left their records on the device (ManyDefer): one sort of (pattern, text,
// strand) keys, one kernel that writes every record -- final text index, strand, coordinates -- and its cigar string
// to its place in the result order, two DMA copies into a pinned block that the result keeps.  The host used to take
// the records through vectors, an append, a loop per strand and a stable sort of 64-byte rows: 35 of the 49 ms of
// 96 barcodes x 330 000 reads.
static int assemble_many(sassy_SearcherType* s, const ManyDefer& fwd, const ManyDefer& rcd, uint32_t n_texts,
                         const uint64_t* d_text_len, uint64_t first_text, sassy_hip_Result* R, bool flip = true) {
  const uint64_t n = (uint64_t)fwd.part.n + rcd.part.n;
  if (n == 0) return 0;
  if (n > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "more than 2^32 records in one result");
  const uint32_t strb = fwd.part.n ? fwd.str_stride : rcd.str_stride;
  if (n * strb > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
  ScanLane& L = s->lanes[0];
  ScanLane& LO = s->lanes[2];  // (its record / string buffers take the assembled result)
  hipStream_t st = s->stream;
  if (int rc = LO.d_trace.reserve(n)) return rc;
  if (int rc = LO.d_str.reserve(n * strb)) return rc;
  if (int rc = L.d_sort.reserve(many_scratch_bytes((uint32_t)n))) return rc;
  if (int rc = L.d_flags.reserve(4)) return rc;
  HIP_TRY(hipMemsetAsync(L.d_flags.p, 0, 4, st));
  hipError_t le = launch_assemble_many(fwd.part, rcd.part, n_texts, d_text_len, first_text, strb, LO.d_trace.p,
                                       reinterpret_cast<char*>(LO.d_str.p), L.d_flags.p, L.d_sort.p, L.d_sort.cap, st, flip ? 1 : 0);
  if (le != hipSuccess) return hip_fail(le, "result assembly launch");
  const size_t rows_off = 256, strs_off = (rows_off + n * sizeof(MatchOut) + 255) / 256 * 256;
  const size_t bytes = strs_off + n * strb + 256;
  if (int rc = L.reserve_pinned(bytes)) return rc;
  HIP_TRY(hipMemcpyAsync(L.h_pin, L.d_flags.p, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(L.h_pin + rows_off, LO.d_trace.p, n * sizeof(MatchOut), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(L.h_pin + strs_off, LO.d_str.p, n * strb, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  uint32_t flags = 0;
  memcpy(&flags, L.h_pin, 4);
  if (flags) return fail(SASSY_HIP_EINVAL, "traceback failed for a reported end position (internal error)");
  const sassy_hip_Match* hm = reinterpret_cast<const sassy_hip_Match*>(L.h_pin + rows_off);
  const char* hs = reinterpret_cast<const char*>(L.h_pin + strs_off);
  if (g_pin_pool.may_adopt(L.h_pin_cap)) {
    R->pin = L.take_pin();
    R->ext_matches = hm;
    R->ext_n = n;
    R->ext_pool = hs;
    R->ext_pool_len = n * strb;
  } else {
    R->matches.assign(hm, hm + n);
    R->pool.assign(hs, n * strb);
  }
  return 0;
}

int search_many_batched(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                               size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                               size_t k, uint32_t flags, sassy_hip_Result* R, bool& handled, bool tiled_only) {
  handled = false;
  if (n_texts < 2 || n_patterns == 0 || (flags & SASSY_HIP_TEXT_ON_DEVICE) || s->profile == PROFILE_ASCII) return 0;
  if (!std::isnan(s->alpha)) return 0;  // overhang gives every text its own special edges: pair by pair
  if (n_texts >= (1u << (32 - kCandTextShift))) return 0;
  size_t max_m = 0;
  for (size_t pi = 0; pi < n_patterns; ++pi) {
    if (!patterns[pi] || pattern_lens[pi] == 0 || k >= pattern_lens[pi]) return 0;
    max_m = std::max(max_m, pattern_lens[pi]);
    if (s->profile == PROFILE_DNA && !acgt_only(patterns[pi], pattern_lens[pi])) return 0;
  }
  for (size_t ti = 0; ti < n_texts; ++ti) {
    if (!texts[ti] && text_lens[ti]) return fail(SASSY_HIP_EINVAL, "null text");
    if (s->profile == PROFILE_DNA && !acgt_only(texts[ti], text_lens[ti])) return 0;
  }
  handled = true;
  struct ProfileGuard {  // Dna searchers borrow the Iupac kernels for the batch (see above)
    sassy_SearcherType* s; Profile saved;
    ~ProfileGuard() { s->profile = saved; }
  } guard{s, s->profile};
  s->profile = PROFILE_IUPAC;

  const bool all = (flags & SASSY_HIP_ALL_MINIMA) != 0;
  const bool wo = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
  const uint64_t pad = ((uint64_t)max_m + k + 1 + 15) / 16 * 16;
  const uint64_t batch_cap = 1ull << 30;  // bytes of device buffer per batch
  uint8_t* hbuf = nullptr;  // the batch in pinned host memory (s->h_stage)
  HostTexts ht, ht_rev;
  size_t t0 = 0;
  while (t0 < n_texts) {
    // ---- lay out texts t0 .. t1 ----
    size_t t1 = t0;
    uint64_t total = 0;
    ht.start.clear(); ht.len.clear();
    while (t1 < n_texts && (t1 == t0 || total + pad + text_lens[t1] <= batch_cap)) {
      if (t1 > t0) total += pad;
      ht.start.push_back(total);
      ht.len.push_back(text_lens[t1]);
      total += text_lens[t1];
      ++t1;
    }
    const size_t nt = t1 - t0;
    if (total > 0) {
      if (int rc = s->reserve_stage(total + 64)) return rc;
      hbuf = s->h_stage;
      if (ht.start[0] > 0) memset(hbuf, 'X', ht.start[0]);
      if (int rc = s->d_text.reserve(total + 64)) return rc;
      if (int rc = s->d_tables.reserve(4 * nt)) return rc;
      if (int rc = layout_and_upload(hbuf, s->d_text.p, texts + t0, text_lens + t0, ht.start.data(), nt, total, (uint8_t)'X',
                                     s->stream)) return rc;
      uint64_t* d_tab = s->d_tables.p;
      HIP_TRY(hipMemcpyAsync(d_tab, ht.start.data(), nt * 8, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(d_tab + nt, ht.len.data(), nt * 8, hipMemcpyHostToDevice, s->stream));
      TextTable tt{d_tab, d_tab + nt, (uint32_t)nt, all ? 1u : 0u}, tt_rev{};
      if (s->rc) {
        // the reversed buffer holds the texts in reverse order, each one reversed
        ht_rev.start.resize(nt); ht_rev.len.resize(nt);
        for (size_t r = 0; r < nt; ++r) {
          const size_t t = nt - 1 - r;
          ht_rev.start[r] = total - (ht.start[t] + ht.len[t]);
          ht_rev.len[r] = ht.len[t];
        }
        HIP_TRY(hipMemcpyAsync(d_tab + 2 * nt, ht_rev.start.data(), nt * 8, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipMemcpyAsync(d_tab + 3 * nt, ht_rev.len.data(), nt * 8, hipMemcpyHostToDevice, s->stream));
        tt_rev = TextTable{d_tab + 2 * nt, d_tab + 3 * nt, (uint32_t)nt, all ? 1u : 0u};
        s->rev_src = nullptr;
        if (int rc = s->d_rev.reserve(total + 64)) return rc;
        hipError_t le = launch_reverse(s->d_text.p, s->d_rev.p, total, s->stream);
        if (le != hipSuccess) return hip_fail(le, "reverse kernel launch");
      }
      std::string err;
      // Many patterns of one length: the pattern-tiled scan (tiled_kernel.hip) takes all of them over the whole
      // batch in one pass per strand -- the separators are characters that match nothing, so after m + k + 1 of
      // them the columns are fresh, as for the scans of one pattern.  SASSY_HIP_MANY_TILED=0 / 1 forces the choice.
      bool tiled_done = false;
      {
        const bool use = many_tiled_wanted(s, pattern_lens, n_patterns, total, k);
        // ... or the seeded search (seed_kernels.hip: 'X' bytes match nothing there) when the patterns and the
        // batch are plain ACGT and its estimate is the lower one (SASSY_HIP_MANY_SEEDED=0 / 1 forces the choice)
        bool seed_batch = false;
        if (use && seeded_hit_rate(pattern_lens[0], k) > 0 && total < (1ull << 36)) {
          const int env_seed = (int)s->sw.many_seeded;
          const double est_seed = seeded_estimate(pattern_lens[0], k, n_patterns, total);
          const double est_tile = (double)total * (double)((n_patterns + 63) / 64) / 2.8e10 + 1.5e-4;
          bool plain = env_seed != 0 && (env_seed > 0 || est_seed < est_tile);
          for (size_t pi = 0; plain && pi < n_patterns; ++pi) plain = acgt_only(patterns[pi], pattern_lens[pi]);
          if (plain) {
            if (int rc = s->d_ncount.reserve(4)) return rc;
            HIP_TRY(hipMemsetAsync(s->d_ncount.p, 0, 4, s->stream));
            hipError_t le = launch_acgt_check(s->d_text.p, total, s->d_ncount.p, s->stream, 1);
            if (le != hipSuccess) return hip_fail(le, "text check kernel launch");
            uint32_t bad = 1;
            HIP_TRY(hipMemcpyAsync(&bad, s->d_ncount.p, 4, hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            plain = !bad;
          }
          seed_batch = plain;
        }
        if (use) {
          const size_t batch_first = R->matches.size(), pool_first = R->pool.size();
          tiled_done = true;
          // the whole call is this one batch, traced, every report is a record: the records are put in order on the
          // device (assemble_many; SASSY_HIP_MANY_ASSEMBLE=0: by the host, as for several batches)
          const bool env_noasm = s->sw.many_assemble == 0;
          // (the device's sort key packs pattern << 33 | text << 1 | strand into bits 0 .. 58: many_keys_kernel)
          const bool on_device = !env_noasm && !wo && !all && std::isnan(s->max_n_frac) && !s->only_best && t0 == 0 &&
                                 t1 == n_texts && batch_first == 0 && pool_first == 0 && !R->pin.h &&
                                 (uint64_t)n_patterns < (1ull << 25) && (uint64_t)n_texts < (1ull << 31);
          ManyDefer defer[2];
          defer[1].lane = 1;
          for (int strand = 0; strand < (s->rc ? 2 : 1) && tiled_done; ++strand) {
            sassy_hip_Encoded tmp;
            tmp.profile = s->profile;
            tmp.rc = false;
            tmp.plen = pattern_lens[0];
            tmp.n_original = n_patterns;
            for (size_t pi = 0; pi < n_patterns; ++pi) {
              tmp.patterns.emplace_back(patterns[pi], patterns[pi] + pattern_lens[pi]);
              if (strand)
                for (uint8_t& c : tmp.patterns.back()) c = complement_char(s->profile, c);
            }
            const size_t first = R->matches.size();
            bool done = false;
            if (seed_batch)
              if (int rc = search_encoded_seeded(s, &tmp, strand ? s->d_rev.p : s->d_text.p, strand ? nullptr : hbuf, total,
                                                 (uint32_t)k, all, wo, R, &done, strand ? &tt_rev : &tt,
                                                 strand ? &ht_rev : &ht, false, on_device ? &defer[strand] : nullptr)) return rc;
            if (!done)
              if (int rc = search_encoded_tiled(s, &tmp, strand ? s->d_rev.p : s->d_text.p, strand ? nullptr : hbuf, total,
                                                (uint32_t)k, all, wo, R, &done, strand ? &tt_rev : &tt,
                                                strand ? &ht_rev : &ht, on_device ? &defer[strand] : nullptr)) return rc;
            if (!done) { tiled_done = false; break; }
            for (size_t i = first; i < R->matches.size(); ++i) {
              sassy_hip_Match& m = R->matches[i];
              if (!strand) { m.text_idx += t0; continue; }
              // reference: src/search.rs:859-873
              const size_t t = nt - 1 - (size_t)m.text_idx;
              const uint64_t len = ht.len[t], rs = m.text_start, re = m.text_end;
              m.strand = 1;
              m.text_idx = t0 + t;
              m.text_start = len - re;
              m.text_end = wo ? UINT64_MAX : len - rs;
            }
          }
          if (tiled_done && on_device) {
            if (int rc = assemble_many(s, defer[0], defer[1], (uint32_t)nt, d_tab + nt, t0, R)) return rc;
          }
          if (!tiled_done) {  // too many end positions for one list: back to one chain per pattern for this batch
            R->matches.resize(batch_first);
            R->pool.resize(pool_first);
          }
        }
        if (tiled_only && !tiled_done && t0 == 0) {  // first batch, nothing appended yet: leave it all to the caller
          handled = false;
          return 0;
        }  // (a later batch that the tiled scan cannot take runs as chains below)
      }
      // one scan per pattern and strand, several in flight (ScanQueue); tag = 2 * pattern + strand
      ScanQueue queue(s, [&](uint64_t tag, ScanOut& so, const PatternPlan& plan, const uint8_t* pat) -> int {
        const size_t pi = (size_t)(tag >> 1);
        const bool is_rc = (tag & 1) != 0;
        const HostTexts& h = is_rc ? ht_rev : ht;
        if (int rc = post_filter(s, so, plan, pat, (uint32_t)k, is_rc ? 1 : 0, is_rc ? nullptr : hbuf,
                                 is_rc ? s->d_rev.p : s->d_text.p, total, !wo, EndFilter(), &h)) return rc;
        size_t first = 0;
        if (int rc = append_matches(so, total, plan, wo, pi, R, first, &h)) return rc;
        for (size_t i = first; i < R->matches.size(); ++i) {
          sassy_hip_Match& m = R->matches[i];
          if (!is_rc) { m.text_idx += t0; continue; }
          // reference: src/search.rs:859-873
          const size_t t = nt - 1 - (size_t)m.text_idx;
          const uint64_t len = ht.len[t], rs = m.text_start, re = m.text_end;
          m.strand = 1;
          m.text_idx = t0 + t;
          m.text_start = len - re;
          m.text_end = wo ? UINT64_MAX : len - rs;
        }
        return 0;
      });
      for (size_t pi = 0; pi < (tiled_done ? 0 : n_patterns); ++pi) {
        PatternPlan plan;
        if (!make_plan(s->profile, patterns[pi], pattern_lens[pi], plan, err)) return fail(SASSY_HIP_EINVAL, err);
        ShardView sh{s->d_text.p, total, 0, 0, true, true};
        if (int rc = queue.submit(plan, patterns[pi], sh, tt, (uint32_t)k, all, !wo, total, 2 * pi)) return rc;
        if (s->rc) {
          std::vector<uint8_t> cp(pattern_lens[pi]);
          for (size_t i = 0; i < cp.size(); ++i) cp[i] = complement_char(s->profile, patterns[pi][i]);
          PatternPlan cplan;
          if (!make_plan(s->profile, cp.data(), cp.size(), cplan, err)) return fail(SASSY_HIP_EINVAL, err);
          ShardView shr{s->d_rev.p, total, 0, 0, true, true};
          if (int rc = queue.submit(cplan, cp.data(), shr, tt_rev, (uint32_t)k, all, !wo, total, 2 * pi + 1)) return rc;
        }
      }
      if (int rc = queue.drain_all()) return rc;
    }
    t0 = t1;
  }
  return 0;
}

// ---- many host texts, one lane per text ----
// The other way to run search_many over many short texts: the texts are laid out block-aligned
// (each starts at a multiple of 64 bytes, padded to whole blocks) and the list-mode DP kernel gets
// one descriptor per text, so every lane walks exactly one text from its column 0 to its end --
// what the reference's multi-text SIMD mode does with its lanes (src/search.rs:615-637).  Nothing
// has to be cut back afterwards: a lane seeds the true text-start column (or the overhang left edge),
// applies the end-of-text rule (or the overhang columns and costs) at its own text's end, and tags
// its reports with the text index.  There is no prefilter in this mode, so it is used where the
// separator layout (search_many_batched) cannot be: overhang searches, the Ascii profile, Dna text
// with other letters, and patterns whose pieces are too short to filter anyway.
// The overhang column and the virtual columns of a pattern of m rows (reference: src/search.rs:347-356, 1695-1748;
// f32 arithmetic as there): *steps = 'N' columns behind the text, *vp = the vertical deltas at the text's start (bit j =
// floor((j+1) alpha) - floor(j alpha) for j < max_overhang, else 1), *cost0 = their sum.
static void overhang_column(const sassy_SearcherType* s, uint32_t m, uint32_t k, uint32_t* steps, unsigned long long* vp, int32_t* cost0) {
  uint64_t st = m;
  if (s->alpha > 0.0f) {
    const float qf = std::ceil(((float)k + s->alpha) / s->alpha);
    if (qf < (float)st) st = (uint64_t)qf;
  }
  if (s->max_overhang >= 0) st = std::min<uint64_t>(st, (uint64_t)s->max_overhang);
  *steps = (uint32_t)st;
  const uint64_t mo = s->max_overhang >= 0 ? (uint64_t)s->max_overhang : UINT64_MAX;
  unsigned long long bits = 0;
  int32_t sum = 0;
  for (uint32_t i = 0; i < m && i < 64; ++i) {
    uint32_t d = 1;
    if (i < mo) d = (uint32_t)((uint64_t)std::floor((float)(i + 1) * s->alpha) - (uint64_t)std::floor((float)i * s->alpha));
    bits |= (unsigned long long)(d & 1u) << i;
    sum += (int32_t)(d & 1u);
  }
  *vp = bits;
  *cost0 = sum;
}

// search_encoded_patterns of an overhang searcher on ONE long text in one pass (the reference's v2 scan takes overhang in its
// tiled loop: src/pattern_tiling/search.rs:222-323; here it used to be a kernel chain per pattern).  The text is a batch of one:
// the seeded search lists the end positions (m + k, len] -- overhang cannot change them --, tiled_pertext_kernel's two edge
// segments add [0, m + k] from the overhang column and (len, len + steps], the common tail applies the report rule and traces.
// Needs what the batch path needs: an Iupac searcher, >= 4 patterns of <= 64 rows with seeds, a text of plain bases (the
// seeded search reads Dna codes).  *done = false: not this shape -- the caller runs the patterns one by one.
int search_encoded_overhang(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr, const uint8_t* h_text,
                            uint64_t text_len, uint32_t k, bool all, bool wo, sassy_hip_Result* R, bool* done) {
  *done = false;
  const size_t m = e->plen;
  if (std::isnan(s->alpha) || s->profile != PROFILE_IUPAC || e->patterns.size() < 4 || e->patterns.size() >= (1u << 24) || m > 64 ||
      2 * k + 3 > 64 || k >= m)
    return 0;
  if (s->sw.overhang_tiled == 0 || s->sw.overhang_seeded == 0 || !(seeded_hit_rate(m, k) > 0)) return 0;
  if (text_len <= m + k + 64 || text_len >= (1ull << 36) || ((uintptr_t)tptr & 15) != 0) return 0;
  if (int rc = s->d_ncount.reserve(4)) return rc;
  if (int rc = s->d_tables.reserve(2)) return rc;
  const uint64_t tab[2] = {0, text_len};
  HIP_TRY(hipMemsetAsync(s->d_ncount.p, 0, 4, s->stream));
  HIP_TRY(hipMemcpyAsync(s->d_tables.p, tab, sizeof tab, hipMemcpyHostToDevice, s->stream));
  hipError_t le = launch_acgt_check(tptr, text_len, s->d_ncount.p, s->stream);
  if (le != hipSuccess) return hip_fail(le, "text check kernel launch");
  uint32_t bad = 1;
  HIP_TRY(hipMemcpyAsync(&bad, s->d_ncount.p, 4, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (bad) return 0;
  uint32_t ov_exact = 0;
  unsigned long long ov_vp = 0;
  int32_t ov_cost0 = 0;
  overhang_column(s, (uint32_t)m, k, &ov_exact, &ov_vp, &ov_cost0);
  const uint64_t* d_tab = s->d_tables.p;
  TextTable tt{d_tab, d_tab + 1, 1u, all ? 1u : 0u, 0u, ov_exact};
  HostTexts ht;
  ht.start.push_back(0);
  ht.len.push_back(text_len);
  const TiledPerText edges{d_tab, d_tab + 1, 1u, ov_exact, s->alpha, ov_vp, ov_cost0, (uint32_t)(m + k)};
  const size_t first = R->matches.size(), pool_first = R->pool.size();
  if (int rc = search_encoded_seeded(s, e, tptr, h_text, text_len, k, all, wo, R, done, &tt, &ht, false, nullptr, &edges)) return rc;
  if (!*done) {
    R->matches.resize(first);
    R->pool.resize(pool_first);
  }
  return 0;
}

int search_many_pertext(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                               size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                               size_t k, uint32_t flags, sassy_hip_Result* R, bool& handled) {
  handled = false;
  if (n_texts < 2 || n_patterns == 0 || (flags & SASSY_HIP_TEXT_ON_DEVICE)) return 0;
  if (n_texts >= (1u << (32 - kCandTextShift))) return 0;
  const bool overhang = !std::isnan(s->alpha);
  size_t max_m = 0;
  bool filterable = true;  // every pattern has selective pieces: the separator layout + prefilter is faster
  for (size_t pi = 0; pi < n_patterns; ++pi) {
    if (!patterns[pi] || pattern_lens[pi] == 0 || k >= pattern_lens[pi]) return 0;
    max_m = std::max(max_m, pattern_lens[pi]);
    if (pattern_lens[pi] / (k + 1) < 7) filterable = false;
  }
  uint64_t longest = 0;
  for (size_t ti = 0; ti < n_texts; ++ti) {
    if (!texts[ti] && text_lens[ti]) return fail(SASSY_HIP_EINVAL, "null text");
    longest = std::max<uint64_t>(longest, text_lens[ti]);
  }
  if (longest > (1u << 20)) return 0;                      // a lane per text only pays for short texts
  if (!overhang && s->profile != PROFILE_ASCII && filterable) return 0;  // search_many_batched takes it
  handled = true;
  if (int rc = s->ensure_device()) return rc;
  const bool all = (flags & SASSY_HIP_ALL_MINIMA) != 0;
  const bool wo = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
  // virtual columns behind a text's end (overhang): at most max_m; padded with 'N' (any other profile
  // never looks at the padding: no end position lies beyond the text)
  // (+ 2 with overhang: the end positions of two texts -- the last virtual column of one, column 0 of the next -- must not
  // be neighbours in the one-pass search's list)
  const uint64_t steps = overhang ? max_m + 2 : 0;
  const uint8_t pad = overhang ? (uint8_t)'N' : (uint8_t)'X';
  // Overhang, several patterns of one length: ONE pass per strand over the batch (tiled_pertext_kernel: a pattern per
  // lane, every text from its own overhang column to its last virtual column; reference: the v2 scan takes overhang in
  // its tiled loop, src/pattern_tiling/search.rs:222-323) instead of one launch per pattern and strand -- 96 barcodes x
  // both strands were 192 launches.  SASSY_HIP_OVERHANG_TILED=0: as before.
  bool tiled_ov = false;
  uint32_t ov_exact = 0;
  unsigned long long ov_vp = 0;
  int32_t ov_cost0 = 0;
  if (overhang && s->profile == PROFILE_IUPAC && n_patterns >= 4 && n_patterns < (1u << 24) && max_m <= 64 && 2 * k + 3 <= 64) {
    const bool env_off = s->sw.overhang_tiled == 0;
    tiled_ov = !env_off;
    for (size_t pi = 0; pi < n_patterns && tiled_ov; ++pi) tiled_ov = pattern_lens[pi] == max_m;
    if (tiled_ov) overhang_column(s, (uint32_t)max_m, (uint32_t)k, &ov_exact, &ov_vp, &ov_cost0);
  }
  // ... and where the seeded search applies (plain ACGT batch, seeds long enough), IT lists the inside of the texts -- the
  // end positions (m + k, len], which overhang cannot change -- at 1.4-1.9 TB/s, and the per-text tiled scan only the two
  // edges of every text (6 % of a 1 kb read).  The batch is then padded with 'X' (the seeded search's separator: matches
  // nothing); the virtual 'N' columns are made by the kernel.  SASSY_HIP_OVERHANG_SEEDED=0: the tiled scan over everything.
  bool seed_ov = false;
  if (tiled_ov && seeded_hit_rate(max_m, k) > 0) {
    const bool env_off = s->sw.overhang_seeded == 0;
    seed_ov = !env_off;
  }
  const uint64_t batch_cap = 1ull << 30;
  uint8_t* hbuf = nullptr;  // the batch in pinned host memory (s->h_stage)
  HostTexts ht;
  std::vector<uint32_t> blk2text;
  std::vector<ChunkDesc> desc;
  std::vector<size_t> order;
  size_t t0 = 0;
  while (t0 < n_texts) {
    // ---- lay out texts t0 .. t1, each in its own whole blocks ----
    size_t t1 = t0;
    uint64_t total = 0;
    ht.start.clear(); ht.len.clear();
    while (t1 < n_texts) {
      const uint64_t slot = (text_lens[t1] + steps + 63) / 64 * 64;
      if (t1 > t0 && total + slot > batch_cap) break;
      ht.start.push_back(total);
      ht.len.push_back(text_lens[t1]);
      total += slot;
      ++t1;
    }
    const size_t nt = t1 - t0;
    if (total > 0) {
      if (int rc = s->reserve_stage(total + 64)) return rc;
      hbuf = s->h_stage;
      uint8_t pad_b = seed_ov ? (uint8_t)'X' : pad;
      // (the upload rides along: the copy of a 32 MB segment runs while the next one is laid out, and the tables below
      // are built while the last copies are in flight)
      if (int rc = s->d_text.reserve(total + 64)) return rc;
      if (int rc = layout_and_upload(hbuf, s->d_text.p, texts + t0, text_lens + t0, ht.start.data(), nt, total, pad_b, s->stream)) return rc;
      blk2text.assign(total / 64, 0u);
      for (size_t i = 0; i < nt; ++i) {
        const uint64_t b0 = ht.start[i] / 64, b1 = (i + 1 < nt ? ht.start[i + 1] : total) / 64;
        for (uint64_t b = b0; b < b1; ++b) blk2text[b] = (uint32_t)i;
      }
      // descriptors, longest texts first so that the lanes of a wave have similar work: a counting
      // sort on the length in blocks (texts are at most 2^20 bytes here), stable
      order.resize(nt);
      {
        std::vector<uint32_t> cnt((1u << 14) + 2, 0u);
        for (size_t i = 0; i < nt; ++i) ++cnt[(ht.len[i] + 63) / 64];
        uint32_t run = 0;
        for (size_t bkt = cnt.size(); bkt-- > 0;) { const uint32_t c = cnt[bkt]; cnt[bkt] = run; run += c; }
        for (size_t i = 0; i < nt; ++i) order[cnt[(ht.len[i] + 63) / 64]++] = i;
      }
      desc.clear();
      for (size_t i : order) {
        if (ht.len[i] == 0) continue;  // an empty text has no matches
        ChunkDesc d;
        d.own_lo = (uint32_t)(ht.start[i] / 64);
        d.own_hi = (uint32_t)((i + 1 < nt ? ht.start[i + 1] : total) / 64);
        d.flags = kDescWholeText;
        d.pad_ = (uint32_t)i;
        desc.push_back(d);
      }
      if (desc.empty()) { t0 = t1; continue; }
      if (int rc = s->d_tables.reserve(2 * nt + 2 * desc.size() + total / 64 / 2 + 8)) return rc;
      uint64_t* d_tab = s->d_tables.p;
      ChunkDesc* d_desc = reinterpret_cast<ChunkDesc*>(d_tab + 2 * nt);
      uint32_t* d_b2t = reinterpret_cast<uint32_t*>(d_tab + 2 * nt + 2 * desc.size());
      HIP_TRY(hipMemcpyAsync(d_tab, ht.start.data(), nt * 8, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(d_tab + nt, ht.len.data(), nt * 8, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(d_desc, desc.data(), desc.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice, s->stream));
      TextTable tt{d_tab, d_tab + nt, (uint32_t)nt, all ? 1u : 0u, 1u};
      if (s->rc) {
        HIP_TRY(hipMemcpyAsync(d_b2t, blk2text.data(), blk2text.size() * 4, hipMemcpyHostToDevice, s->stream));
        s->rev_src = nullptr;
        if (int rc = s->d_rev.reserve(total + 64)) return rc;
        hipError_t le = launch_reverse_texts(s->d_text.p, s->d_rev.p, total, d_b2t, d_tab, d_tab + nt, pad_b, s->stream);
        if (le != hipSuccess) return hip_fail(le, "reverse kernel launch");
      }
      std::string err;
      if (tiled_ov) {
        bool seed_this = seed_ov;
        if (seed_this) {  // the seeded search reads Dna codes: the batch must hold plain bases (and the padding) only
          if (int rc = s->d_ncount.reserve(4)) return rc;
          HIP_TRY(hipMemsetAsync(s->d_ncount.p, 0, 4, s->stream));
          hipError_t le = launch_acgt_check(s->d_text.p, total, s->d_ncount.p, s->stream, 1);
          if (le != hipSuccess) return hip_fail(le, "text check kernel launch");
          uint32_t bad = 1;
          HIP_TRY(hipMemcpyAsync(&bad, s->d_ncount.p, 4, hipMemcpyDeviceToHost, s->stream));
          HIP_TRY(hipStreamSynchronize(s->stream));
          seed_this = !bad;
        }
        const size_t batch_first = R->matches.size(), pool_first = R->pool.size();
        TextTable tto = tt;
        tto.per_text = 0;
        tto.ov_steps = ov_exact;
        TiledPerText pt{d_tab, d_tab + nt, (uint32_t)nt, ov_exact, s->alpha, ov_vp, ov_cost0, 0u};
        TiledPerText pt_edges = pt;
        pt_edges.edge_cols = (uint32_t)(max_m + k);
        bool ok_all = true;
        // the whole call is this one batch, traced, every report a record: both strands' records stay on the device and are
        // put in order there (assemble_many: every text was reversed in its own slot -- no index flip)
        const bool env_noasm = s->sw.many_assemble == 0;
        const bool on_device = !env_noasm && !wo && !all && std::isnan(s->max_n_frac) && !s->only_best && t0 == 0 && t1 == n_texts &&
                               batch_first == 0 && pool_first == 0 && !R->pin.h && (uint64_t)n_patterns < (1ull << 25) &&
                               (uint64_t)n_texts < (1ull << 31);
        ManyDefer defer[2];
        defer[1].lane = 1;
        for (int strand = 0; strand < (s->rc ? 2 : 1) && ok_all; ++strand) {
          sassy_hip_Encoded tmp;
          tmp.profile = s->profile;
          tmp.rc = false;
          tmp.plen = max_m;
          tmp.n_original = n_patterns;
          for (size_t pi = 0; pi < n_patterns; ++pi) {
            tmp.patterns.emplace_back(patterns[pi], patterns[pi] + pattern_lens[pi]);
            if (strand)
              for (uint8_t& c : tmp.patterns.back()) c = complement_char(s->profile, c);
          }
          const size_t first = R->matches.size();
          bool done = false;
          if (seed_this)
            if (int rc = search_encoded_seeded(s, &tmp, strand ? s->d_rev.p : s->d_text.p, strand ? nullptr : hbuf, total, (uint32_t)k, all,
                                               wo, R, &done, &tto, &ht, false, on_device ? &defer[strand] : nullptr, &pt_edges)) return rc;
          if (!done)
            if (int rc = search_encoded_tiled(s, &tmp, strand ? s->d_rev.p : s->d_text.p, strand ? nullptr : hbuf, total, (uint32_t)k, all,
                                              wo, R, &done, &tto, &ht, on_device ? &defer[strand] : nullptr, &pt)) return rc;
          if (!done) { ok_all = false; break; }
          for (size_t i = first; i < R->matches.size(); ++i) {
            sassy_hip_Match& m = R->matches[i];
            if (strand) {  // reference: src/search.rs:859-873 (each text was reversed in its own slot)
              const uint64_t len = ht.len[m.text_idx], rs = m.text_start, re = m.text_end;
              m.strand = 1;
              m.text_start = len - re;
              m.text_end = wo ? UINT64_MAX : len - rs;
            }
            m.text_idx += t0;
          }
        }
        if (ok_all && on_device)
          if (int rc = assemble_many(s, defer[0], defer[1], (uint32_t)nt, d_tab + nt, t0, R, false)) return rc;
        if (ok_all) {
          t0 = t1;
          continue;
        }
        R->matches.resize(batch_first);  // (more end positions than the list holds: the patterns one by one)
        R->pool.resize(pool_first);
        if (pad_b != pad) {  // ... whose DP reads the virtual columns from the buffer: pad it with 'N' after all
          pad_b = pad;
          layout_texts(hbuf, texts + t0, text_lens + t0, ht.start.data(), nt, total, pad_b);
          HIP_TRY(hipMemcpyAsync(s->d_text.p, hbuf, total, hipMemcpyHostToDevice, s->stream));
          if (s->rc) {
            hipError_t le = launch_reverse_texts(s->d_text.p, s->d_rev.p, total, d_b2t, d_tab, d_tab + nt, pad_b, s->stream);
            if (le != hipSuccess) return hip_fail(le, "reverse kernel launch");
          }
        }
      }
      ScanQueue queue(s, [&](uint64_t tag, ScanOut& so, const PatternPlan& plan, const uint8_t* pat) -> int {
        const size_t pi = (size_t)(tag >> 1);
        const bool is_rc = (tag & 1) != 0;
        // N counting for max_n_frac: the forward buffer has a host copy, the reversed one lives on the device
        if (int rc = post_filter(s, so, plan, pat, (uint32_t)k, is_rc ? 1 : 0, is_rc ? nullptr : hbuf,
                                 is_rc ? s->d_rev.p : s->d_text.p, total, !wo, EndFilter(), &ht)) return rc;
        size_t first = 0;
        if (int rc = append_matches(so, total, plan, wo, pi, R, first, &ht)) return rc;
        for (size_t i = first; i < R->matches.size(); ++i) {
          sassy_hip_Match& m = R->matches[i];
          if (is_rc) {  // reference: src/search.rs:859-873 (each text was reversed in its own slot)
            const uint64_t len = ht.len[m.text_idx], rs = m.text_start, re = m.text_end;
            m.strand = 1;
            m.text_start = len - re;
            m.text_end = wo ? UINT64_MAX : len - rs;
          }
          m.text_idx += t0;
        }
        return 0;
      });
      for (size_t pi = 0; pi < n_patterns; ++pi) {
        PatternPlan plan;
        if (!make_plan(s->profile, patterns[pi], pattern_lens[pi], plan, err)) return fail(SASSY_HIP_EINVAL, err);
        ShardView sh{s->d_text.p, total, 0, 0, true, true};
        if (int rc = queue.submit(plan, patterns[pi], sh, tt, (uint32_t)k, all, !wo, total, 2 * pi, nullptr, 0, nullptr,
                                  d_desc, (uint32_t)desc.size())) return rc;
        if (s->rc) {
          std::vector<uint8_t> cp(pattern_lens[pi]);
          for (size_t i = 0; i < cp.size(); ++i) cp[i] = complement_char(s->profile, patterns[pi][i]);
          PatternPlan cplan;
          if (!make_plan(s->profile, cp.data(), cp.size(), cplan, err)) return fail(SASSY_HIP_EINVAL, err);
          ShardView shr{s->d_rev.p, total, 0, 0, true, true};
          if (int rc = queue.submit(cplan, cp.data(), shr, tt, (uint32_t)k, all, !wo, total, 2 * pi + 1, nullptr, 0, nullptr,
                                    d_desc, (uint32_t)desc.size())) return rc;
        }
      }
      if (int rc = queue.drain_all()) return rc;
    }
    t0 = t1;
  }
  return 0;
}

}  // namespace sassy_hip

extern "C" {

long sassy_hip_seed_layout(const char* alphabet, const uint8_t* const* patterns, size_t n_patterns, size_t pattern_len, size_t k,
                           uint32_t* out_end, uint32_t* out_len) {
  if (!alphabet || !patterns || !out_end || !out_len || pattern_len == 0 || pattern_len > 64 || k > 7 || pattern_len / (k + 1) < 1)
    return -1;
  const std::string a(alphabet);
  const int profile = a == "dna" ? PROFILE_DNA : a == "iupac" ? PROFILE_IUPAC : a == "ascii" ? PROFILE_ASCII : -1;
  if (profile < 0) return -1;
  seed_layout(true, profile, patterns, n_patterns, (uint32_t)pattern_len, (uint32_t)k, out_end, out_len);
  return (long)(k + 1);
}

long sassy_hip_seed_test_rows(size_t pattern_len, size_t k, const uint32_t* seed_end, const uint32_t* seed_len, uint32_t* out_rows,
                              uint32_t* out_win_left) {
  if (!seed_end || !seed_len || !out_rows || !out_win_left || pattern_len == 0 || pattern_len > 32 || k > 7) return -1;
  for (size_t i = 0; i <= k; ++i)
    if (seed_len[i] == 0 || seed_len[i] > kSeedMaxLen || seed_end[i] > pattern_len || seed_end[i] < seed_len[i]) return -1;
  uint32_t max_off = 0;
  seed_test_rows((uint32_t)pattern_len, (uint32_t)k, seed_end, seed_len, out_rows, out_win_left, &max_off);
  return (long)max_off;
}

}  // extern "C"
