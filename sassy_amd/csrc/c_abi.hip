// c_abi.hip -- the C-ABI of include/sassy.h (the reference's c/sassy.h:9-63) and include/sassy_hip.h, the switch table
// (switches.h) and the synthetic inputs of the benchmarks.
#include "host_internal.h"

namespace sassy_hip {
thread_local std::string g_err;
// ---- switches.h: the one table of switches, and the library's only reader of the environment ----
namespace {
struct SwitchRow {
  const char* name;
  long Switches::*field;
  long dflt;
  const char* doc;
};
#define SASSY_HIP_SWITCH_ROW(name, dflt, doc) {#name, &Switches::name, (long)(dflt), doc},
const SwitchRow kSwitchRows[] = {SASSY_HIP_SWITCHES(SASSY_HIP_SWITCH_ROW)};
#undef SASSY_HIP_SWITCH_ROW
}  // namespace
Switches load_switches() {
  Switches sw;
  std::string var;
  auto env_of = [&](const char* name) -> const char* {
    var = "SASSY_HIP_";
    for (const char* c = name; *c; ++c) var.push_back((char)toupper((unsigned char)*c));
    return getenv(var.c_str());
  };
  for (const SwitchRow& r : kSwitchRows)
    if (const char* v = env_of(r.name))
      if (*v) sw.*(r.field) = strtol(v, nullptr, 10);
  if (const char* v = env_of("devices")) sw.devices = v;
  return sw;
}
bool set_switch(Switches& sw, const char* name, long value) {
  for (const SwitchRow& r : kSwitchRows)
    if (!strcmp(r.name, name)) { sw.*(r.field) = value; return true; }
  return false;
}
bool get_switch(const Switches& sw, const char* name, long* value) {
  for (const SwitchRow& r : kSwitchRows)
    if (!strcmp(r.name, name)) { *value = sw.*(r.field); return true; }
  return false;
}
const char* switch_table() {
  static const std::string table = [] {
    std::string t;
    for (const SwitchRow& r : kSwitchRows) t += std::string(r.name) + "\t" + std::to_string(r.dflt) + "\t" + r.doc + "\n";
    t += "devices\t\tthe drop-in search() of include/sassy.h fans a host text over these devices (\"all\", or a list like 0,1,2)\n";
    return t;
  }();
  return table.c_str();
}

// "ascii" | "dna" | "iupac" (reference: src/c.rs:52-70)
bool parse_alphabet(const char* alphabet, Profile& pr) {
  std::string a(alphabet ? alphabet : "");
  for (char& c : a) c = (char)tolower((unsigned char)c);
  if (a == "dna") pr = PROFILE_DNA;
  else if (a == "iupac") pr = PROFILE_IUPAC;
  else if (a == "ascii") pr = PROFILE_ASCII;
  else return false;
  return true;
}
}  // namespace sassy_hip

using namespace sassy_hip;

// ======================================================================== C-ABI
extern "C" {

const char* sassy_hip_last_error(void) { return g_err.c_str(); }
const char* sassy_hip_version(void) { return "sassy-hip 0.1 (gfx950)"; }

int sassy_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}


sassy_SearcherType* sassy_hip_searcher_new(const char* alphabet, bool rc, float alpha) {
  if (!alphabet) { fail(SASSY_HIP_EINVAL, "Alphabet pointer must not be null"); return nullptr; }
  Profile pr;
  if (!parse_alphabet(alphabet, pr)) {
    fail(SASSY_HIP_EINVAL, std::string("Unsupported alphabet: ") + alphabet);
    return nullptr;
  }
  if (!std::isnan(alpha)) {  // reference: Searcher::_overhang_check (src/search.rs:373-383)
    if (pr != PROFILE_IUPAC) {
      fail(SASSY_HIP_EUNSUPPORTED, "Overhang is not supported for this alphabet (iupac only)");
      return nullptr;
    }
    if (!(alpha >= 0.0f && alpha <= 1.0f)) {
      fail(SASSY_HIP_EINVAL, "Alpha must be in range 0.0 <= alpha <= 1.0");
      return nullptr;
    }
  }
  sassy_SearcherType* s = new sassy_SearcherType();
  s->profile = pr;
  s->rc = rc;
  s->alpha = alpha;
  return s;
}

int sassy_hip_set_max_overhang(sassy_SearcherType* s, long max_overhang) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  s->max_overhang = max_overhang < 0 ? -1 : max_overhang;
  return 0;
}

[[noreturn]] static void die(const char* msg) {
  std::fprintf(stderr, "sassy (hip): %s\n", msg);
  std::abort();
}

sassy_SearcherType* sassy_searcher(const char* alphabet, bool rc, float alpha) {
  sassy_SearcherType* s = sassy_hip_searcher_new(alphabet, rc, alpha);
  if (!s) die(g_err.c_str());  // the reference panics (src/c.rs:57,66)
  return s;
}

void sassy_searcher_free(sassy_SearcherType* ptr) {
  if (!ptr) die("Pointer to SearcherType must not be null");  // src/c.rs:75-77
  DeviceGuard on_device(ptr);
  delete ptr;
}

int sassy_hip_set_device(sassy_SearcherType* s, int device) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  if ((s->device_ready || s->bound) && s->device != device)
    return fail(SASSY_HIP_EINVAL, "the searcher already works on another device");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
    (void)hipGetLastError();
    return fail(SASSY_HIP_EINVAL, "no such HIP device");
  }
  s->device = device;
  return 0;
}
int sassy_hip_get_device(const sassy_SearcherType* s) { return s ? s->device : -1; }

int sassy_hip_set_stream(sassy_SearcherType* s, void* hip_stream) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  SASSY_NO_TICKETS(s);
  DeviceGuard on_device(s);
  ScanLane& l0 = s->lanes[0];
  if (l0.own_stream && l0.stream) (void)hipStreamDestroy(l0.stream);
  s->user_stream = reinterpret_cast<hipStream_t>(hip_stream);
  l0.stream = s->user_stream;
  l0.own_stream = false;
  if (!hip_stream && s->device_ready) {
    HIP_TRY(hipStreamCreateWithFlags(&l0.stream, hipStreamNonBlocking));
    l0.own_stream = true;
  }
  s->stream = l0.stream;
  return 0;
}

int sassy_hip_get_stats(const sassy_SearcherType* s, sassy_hip_Stats* out) {
  if (!s || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  *out = s->stats;
  return 0;
}

int sassy_hip_enable_counters(sassy_SearcherType* s, int on) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  s->want_counters = on != 0;
  return 0;
}

int sassy_hip_set_prefilter(sassy_SearcherType* s, int mode) {
  if (!s || mode < -1 || mode > 1) return fail(SASSY_HIP_EINVAL, "prefilter mode must be -1, 0 or 1");
  s->prefilter = mode;
  return 0;
}

int sassy_hip_set_fused(sassy_SearcherType* s, int on) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  s->fuse = on != 0;
  s->sw.fused = on != 0;
  return 0;
}

int sassy_hip_set_reference_lanes(sassy_SearcherType* s, int lanes) {
  if (!s || (lanes != 0 && lanes != 4 && lanes != 8)) return fail(SASSY_HIP_EINVAL, "reference lanes must be 0, 4 or 8");
  s->ref_lanes = (uint32_t)lanes;
  s->sw.ref_lanes = lanes;
  return 0;
}

int sassy_hip_set_geometry_tuner(sassy_SearcherType* s, int on) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  s->tune = on != 0;
  s->sw.tune = on != 0;
  return 0;
}

int sassy_hip_set_pipe_depth(sassy_SearcherType* s, int depth) {
  if (!s || depth < 1 || depth > kMaxLanes) return fail(SASSY_HIP_EINVAL, "pipe depth must be 1 .. 4");
  for (sassy_hip_Ticket* t : s->lane_ticket)
    if (t) return fail(SASSY_HIP_EINVAL, "searches are in flight");
  s->pipe_depth = depth;
  s->sw.pipe_depth = depth;
  s->last_begun_lane = -1;
  return 0;
}

int sassy_hip_set_timing(sassy_SearcherType* s, int level) {
  if (!s || level < 0 || level > 2) return fail(SASSY_HIP_EINVAL, "timing level must be 0, 1 or 2");
  s->timing = level;
  s->sw.timing = level;
  return 0;
}

// One entry of the searcher's switch table (switches.h) by name -- lower case, without the SASSY_HIP_ prefix.  The table
// was filled from its defaults and the environment when the searcher was made; this is how a test or a timing tool
// forces another kernel path on a searcher that already exists.  Refused while searches are in flight.
int sassy_hip_set_option(sassy_SearcherType* s, const char* name, long value) {
  if (!s || !name) return fail(SASSY_HIP_EINVAL, "null argument");
  for (sassy_hip_Ticket* t : s->lane_ticket)
    if (t) return fail(SASSY_HIP_EINVAL, "searches are in flight");
  if (!set_switch(s->sw, name, value)) return fail(SASSY_HIP_EINVAL, std::string("no such option: ") + name);
  s->apply_switches();
  s->last_begun_lane = -1;
  return 0;
}
int sassy_hip_get_option(const sassy_SearcherType* s, const char* name, long* value) {
  if (!s || !name || !value) return fail(SASSY_HIP_EINVAL, "null argument");
  if (!get_switch(s->sw, name, value)) return fail(SASSY_HIP_EINVAL, std::string("no such option: ") + name);
  return 0;
}
const char* sassy_hip_option_table(void) { return switch_table(); }

int sassy_hip_set_only_best_match(sassy_SearcherType* s, int on) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  s->only_best = on != 0;
  return 0;
}

int sassy_hip_set_max_n_frac(sassy_SearcherType* s, float max_n_frac) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  // the reference treats 1.0 as "no filter" (src/search.rs:454-460)
  s->max_n_frac = (std::isnan(max_n_frac) || max_n_frac == 1.0f) ? NAN : max_n_frac;
  return 0;
}

int sassy_hip_search_with_fn(sassy_SearcherType* s, const uint8_t* pattern, size_t pattern_len,
                             const uint8_t* text, size_t text_len, size_t k, uint32_t flags,
                             sassy_hip_end_filter fn, void* user, sassy_hip_Result** out) {
  if (!s || !pattern || (!text && text_len) || !out || !fn) return fail(SASSY_HIP_EINVAL, "Pointers in search() must not be null");
  SASSY_NO_TICKETS(s);
  DeviceGuard on_device(s);
  if (flags & SASSY_HIP_TEXT_ON_DEVICE) return fail(SASSY_HIP_EINVAL, "search_with_fn needs the text in host memory");
  const double t0 = now_ms();
  reset_stats(s);
  EndFilter ef;
  ef.fn = fn;
  ef.user = user;
  sassy_hip_Result* R = new sassy_hip_Result();
  if (int rc = search_text(s, pattern, pattern_len, text, text_len, k, flags, 0, true, s->rc, R, ef)) { delete R; return rc; }
  if (R->pool.empty()) R->pool.push_back('\0');
  s->stats.total_ms = now_ms() - t0;
  s->stats.host_post_ms = s->stats.total_ms - s->stats.host_enqueue_ms - s->stats.host_wait_ms;
  *out = R;
  return 0;
}

int sassy_hip_search_many(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                          size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                          size_t k, uint32_t flags, sassy_hip_Result** out) {
  if (!s || !out || (n_patterns && (!patterns || !pattern_lens)) || (n_texts && (!texts || !text_lens)))
    return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_NO_TICKETS(s);
  DeviceGuard on_device(s);
  if (s->rc && s->profile == PROFILE_ASCII && n_patterns && n_texts)  // as in search_text: the reference panics here
    return fail(SASSY_HIP_EUNSUPPORTED, "reverse complement is not defined for the ascii alphabet");
  const double t0 = now_ms();
  reset_stats(s);
  if (int rc = s->ensure_device()) return rc;
  std::unique_ptr<sassy_hip_Result> R(new sassy_hip_Result());
  bool handled = false;
  {  // patterns of one length over many host texts: the pattern-tiled scan over the separator layout, if it pays
    uint64_t sum = 0;
    for (size_t ti = 0; ti < n_texts; ++ti) sum += text_lens[ti];
    if (n_texts >= 2 && many_tiled_wanted(s, pattern_lens, n_patterns, sum, k))
      if (int rc = search_many_batched(s, patterns, pattern_lens, n_patterns, texts, text_lens, n_texts, k, flags, R.get(),
                                       handled, true)) return rc;
  }
  if (!handled)
    if (int rc = search_many_pertext(s, patterns, pattern_lens, n_patterns, texts, text_lens, n_texts, k, flags, R.get(), handled))
      return rc;
  if (!handled)
    if (int rc = search_many_batched(s, patterns, pattern_lens, n_patterns, texts, text_lens, n_texts, k, flags, R.get(), handled))
      return rc;
  // Device-resident texts, forward strand: every (pattern, text) pair is one scan job; several are in flight
  // on the searcher's lanes (ScanQueue), so the latency-bound tail of one pair runs next to the filter of the
  // next instead of the host waiting for each pair in turn (reference: search_many spreads the pairs over
  // threads, src/search.rs:531-603).  Both-strand searchers keep the pair loop below (its one-pass two-strand
  // path already uses two lanes per pair).
  if (!handled && (flags & SASSY_HIP_TEXT_ON_DEVICE) && !s->rc && n_patterns * n_texts > 1) {
    handled = true;
    const bool all = (flags & SASSY_HIP_ALL_MINIMA) != 0;
    const bool wo = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
    if (k > 0x7FFFFFFFu) return fail(SASSY_HIP_EINVAL, "k too large");
    for (size_t ti = 0; ti < n_texts; ++ti) {
      if (!texts[ti] && text_lens[ti]) return fail(SASSY_HIP_EINVAL, "null text");
      if (text_lens[ti] && ((uintptr_t)texts[ti] & 15) != 0) return fail(SASSY_HIP_EINVAL, "device text pointer must be 16-byte aligned");
    }
    sassy_hip_Result* Rp = R.get();
    ScanQueue queue(s, [&](uint64_t tag, ScanOut& so, const PatternPlan& plan, const uint8_t* pat) -> int {
      const size_t pi = (size_t)(tag / n_texts), ti = (size_t)(tag % n_texts);
      if (int rc = post_filter(s, so, plan, pat, (uint32_t)k, 0, nullptr, texts[ti], text_lens[ti], !wo, EndFilter())) return rc;
      size_t first = 0;
      if (int rc = append_matches(so, text_lens[ti], plan, wo, pi, Rp, first)) return rc;
      for (size_t i = first; i < Rp->matches.size(); ++i) Rp->matches[i].text_idx = ti;
      return 0;
    });
    std::string err;
    for (size_t ti = 0; ti < n_texts; ++ti) {
      if (text_lens[ti] == 0) continue;  // no reports for an empty text (src/search.rs:1314-1316)
      for (size_t pi = 0; pi < n_patterns; ++pi) {
        if (!patterns[pi]) return fail(SASSY_HIP_EINVAL, "null pattern");
        PatternPlan plan;
        if (!make_plan(s->profile, patterns[pi], pattern_lens[pi], plan, err)) return fail(SASSY_HIP_EINVAL, err);
        ShardView sh{texts[ti], text_lens[ti], 0, 0, true, true};
        if (int rc = queue.submit(plan, patterns[pi], sh, TextTable{}, (uint32_t)k, all, !wo, text_lens[ti],
                                  (uint64_t)pi * n_texts + ti)) return rc;
      }
    }
    if (int rc = queue.drain_all()) return rc;
  }
  // otherwise: text-major (each host text is uploaded once), pattern-major in the result
  for (size_t ti = 0; !handled && ti < n_texts; ++ti) {
    const uint8_t* tptr = texts[ti];
    uint32_t f = flags;
    if (!tptr && text_lens[ti]) return fail(SASSY_HIP_EINVAL, "null text");
    if (!(flags & SASSY_HIP_TEXT_ON_DEVICE) && text_lens[ti]) {
      if (int rc = s->d_text.reserve(text_lens[ti] + 64)) return rc;
      HIP_TRY(hipMemcpyAsync(s->d_text.p, tptr, text_lens[ti], hipMemcpyHostToDevice, s->stream));
    }
    for (size_t pi = 0; pi < n_patterns; ++pi) {
      if (!patterns[pi]) return fail(SASSY_HIP_EINVAL, "null pattern");
      const size_t first = R->matches.size();
      // host texts: the scans read the uploaded copy, the host-side filters (if any) the original
      if (int rc = search_text(s, patterns[pi], pattern_lens[pi], tptr, text_lens[ti], k, f, pi, true, s->rc, R.get(),
                               EndFilter(), !(flags & SASSY_HIP_TEXT_ON_DEVICE))) return rc;
      for (size_t i = first; i < R->matches.size(); ++i) R->matches[i].text_idx = ti;
    }
  }
  std::stable_sort(R->matches.begin(), R->matches.end(), [](const sassy_hip_Match& a, const sassy_hip_Match& b) {
    if (a.pattern_idx != b.pattern_idx) return a.pattern_idx < b.pattern_idx;
    return a.text_idx < b.text_idx;
  });
  if (R->pool.empty()) R->pool.push_back('\0');
  s->stats.total_ms = now_ms() - t0;
  s->stats.host_post_ms = s->stats.total_ms - s->stats.host_enqueue_ms - s->stats.host_wait_ms;
  *out = R.release();
  return 0;
}

const char* sassy_hip_tsv_header(void) {
  return "pat_id\ttext_id\tcost\tstrand\tstart\tend\tmatch_region\tcigar\n";  // bin/grep.rs:465-470
}

long sassy_hip_format_tsv(const sassy_SearcherType* s, const sassy_hip_Match* mp, const char* cigar, const char* pat_id,
                          const char* text_id, const uint8_t* text, size_t text_len, int sam, char* buf, size_t cap) {
  if (!s || !mp || !cigar || !pat_id || !text_id || (!text && text_len) || (!buf && cap))
    return -(long)fail(SASSY_HIP_EINVAL, "null argument");
  const sassy_hip_Match& m = *mp;
  if (m.text_start > m.text_end || m.text_end > text_len)
    return -(long)fail(SASSY_HIP_EINVAL, "match has no text span (searched without trace?) or exceeds the text");
  std::string row;
  row.reserve(64 + (m.text_end - m.text_start) + strlen(cigar));
  row += pat_id; row += '\t'; row += text_id; row += '\t';
  row += std::to_string(m.cost); row += '\t';
  row += m.strand ? '-' : '+'; row += '\t';
  row += std::to_string(m.text_start); row += '\t';
  row += std::to_string(m.text_end); row += '\t';
  if (m.strand && !sam) {  // pattern direction: reverse complement (bin/grep.rs:738-747)
    for (uint64_t i = m.text_end; i > m.text_start; --i) row += (char)complement_char(s->profile, text[i - 1]);
  } else {
    row.append(reinterpret_cast<const char*>(text) + m.text_start, m.text_end - m.text_start);
  }
  row += '\t';
  const char* cig = cigar;
  if (m.strand && sam) {  // text direction: reverse the run list (bin/grep.rs:749-757)
    std::vector<std::string> runs;
    for (const char* p = cig; *p;) {
      const char* q = p;
      while (*q >= '0' && *q <= '9') ++q;
      runs.emplace_back(p, q + 1);
      p = q + 1;
    }
    for (size_t i = runs.size(); i > 0; --i) row += runs[i - 1];
  } else {
    row += cig;
  }
  row += '\n';
  if (cap) {
    const size_t ncopy = std::min(row.size(), cap - 1);
    memcpy(buf, row.data(), ncopy);
    buf[ncopy] = 0;
  }
  return (long)row.size();
}

int sassy_hip_search(sassy_SearcherType* s, const uint8_t* pattern, size_t pattern_len,
                     const uint8_t* text, size_t text_len, size_t k, uint32_t flags,
                     sassy_hip_Result** out) {
  if (!s || !pattern || (!text && text_len) || !out) return fail(SASSY_HIP_EINVAL, "Pointers in search() must not be null");
  SASSY_NO_TICKETS(s);
  DeviceGuard on_device(s);
  const double t0 = now_ms();
  reset_stats(s);
  sassy_hip_Result* R = new sassy_hip_Result();
  if (int rc = search_text(s, pattern, pattern_len, text, text_len, k, flags, 0, true, s->rc, R)) { delete R; return rc; }
  if (R->pool.empty()) R->pool.push_back('\0');
  s->stats.total_ms = now_ms() - t0;
  s->stats.host_post_ms = s->stats.total_ms - s->stats.host_enqueue_ms - s->stats.host_wait_ms;
  *out = R;
  return 0;
}

uint64_t sassy_hip_required_halo(size_t pattern_len, size_t k) {
  // warm-up blocks + the traceback window, rounded up to whole 128-byte lines
  // warm-up blocks, the blocks the prefilter looks back into, and the traceback window
  const uint64_t wb = warmup_blocks((uint32_t)pattern_len, (uint32_t)k);
  uint64_t h = std::max<uint64_t>(64 * (wb + 4), pattern_len + k);
  return (h + 127) / 128 * 128;
}

int sassy_hip_search_shard(sassy_SearcherType* s, const uint8_t* pattern, size_t pattern_len,
                           const uint8_t* d_text, uint64_t halo_len, uint64_t shard_len,
                           uint64_t global_offset, uint64_t total_len, size_t k, uint32_t flags,
                           sassy_hip_Result** out) {
  if (!s || !pattern || !d_text || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_NO_TICKETS(s);
  DeviceGuard on_device(s);
  if (halo_len % 64 || global_offset % 64) return fail(SASSY_HIP_EINVAL, "halo_len and global_offset must be multiples of 64");
  if (global_offset < halo_len) return fail(SASSY_HIP_EINVAL, "halo reaches left of the text start");
  if (global_offset + shard_len > total_len) return fail(SASSY_HIP_EINVAL, "shard exceeds the text");
  const bool is_first = global_offset == 0, is_last = global_offset + shard_len == total_len;
  if (!is_last && shard_len % 64) return fail(SASSY_HIP_EINVAL, "inner shard lengths must be multiples of 64");
  // (a halo that reaches byte 0 of the text is as long as a halo can be: a short text cut into many shards)
  if (!is_first && halo_len < sassy_hip_required_halo(pattern_len, k) && halo_len != global_offset)
    return fail(SASSY_HIP_EINVAL, "halo too short");
  if (((uintptr_t)d_text & 15) != 0) return fail(SASSY_HIP_EINVAL, "device text pointer must be 16-byte aligned");
  const double t0 = now_ms();
  reset_stats(s);
  PatternPlan plan;
  std::string err;
  if (!make_plan(s->profile, pattern, pattern_len, plan, err)) return fail(SASSY_HIP_EINVAL, err);
  if (int rc = s->ensure_device()) return rc;
  sassy_hip_Result* R = new sassy_hip_Result();
  if (shard_len > 0) {
    ShardView sh{d_text, halo_len + shard_len, halo_len, global_offset - halo_len,
                 global_offset == halo_len, is_last};  // (text_start: buffer byte 0 is column 0 of the text)
    sh.adopt_ok = true;  // (search_shard applies no reporting modes: the records are final as the kernels write them)
    ScanOut so;
    const bool wo = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
    if (int rc = run_scan(s, sh, plan, (uint32_t)k, (flags & SASSY_HIP_ALL_MINIMA) != 0, pattern, !wo,
                          total_len, so)) { delete R; return rc; }
    size_t first = 0;
    if (int rc = append_matches(so, total_len, plan, wo, 0, R, first)) { delete R; return rc; }
    R->exit_state = so.exit_state;
    R->conditional_index = so.conditional_index;
  }
  s->stats.total_ms = now_ms() - t0;
  s->stats.host_post_ms = s->stats.total_ms - s->stats.host_enqueue_ms - s->stats.host_wait_ms;
  *out = R;
  return 0;
}

// ---- several shards -> one result (the chain of DESIGN.md "seams", one level up) ----
int sassy_hip_merge_shards(const sassy_hip_Result* const* results, size_t n, int incoming_state, sassy_hip_Result** out) {
  if ((!results && n) || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  if (incoming_state < 0 || incoming_state > 2) return fail(SASSY_HIP_EINVAL, "incoming_state must be 0 (FALSE), 1 (TRUE) or 2 (PASS)");
  std::unique_ptr<sassy_hip_Result> R(new sassy_hip_Result());
  size_t total = 0, pool_total = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!results[i]) return fail(SASSY_HIP_EINVAL, "null shard result");
    total += results[i]->size();
    pool_total += results[i]->pool_size();
  }
  R->matches.reserve(total);
  int incoming = incoming_state;  // decreasing-state arriving at the left edge of shard i
  for (size_t i = 0; i < n; ++i) {
    const sassy_hip_Result* r = results[i];
    const sassy_hip_Match* m = r->data();
    const char* pool = r->pool_data();
    for (size_t j = 0; j < r->size(); ++j) {
      if ((int64_t)j == r->conditional_index) {
        // this report's plateau began left of the shard: it stands iff the plateau was entered by a decrease
        if (incoming == kStateDecFalse) continue;
        if (incoming == kStatePass) {  // nobody to the left of results[0] could tell: still conditional in the merged result
          if (R->conditional_index >= 0) return fail(SASSY_HIP_EINVAL, "two reports depend on the shard in front of the first one");
          R->conditional_index = (int64_t)R->matches.size();
        }
      }
      sassy_hip_Match x = m[j];
      const size_t off = R->pool.size();
      if (off + x.cigar_len + 1 > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
      R->pool.append(pool + x.cigar_off, x.cigar_len);
      R->pool.push_back('\0');
      x.cigar_off = (uint32_t)off;
      R->matches.push_back(x);
    }
    if (r->exit_state != kStatePass) incoming = r->exit_state;
  }
  (void)pool_total;
  if (R->pool.empty()) R->pool.push_back('\0');
  R->exit_state = incoming;
  *out = R.release();
  return 0;
}

}  // extern "C"

extern "C" {

// ---- searches in flight: begin / finish ----
// A stream of searches over a resident text (many patterns against one genome) is pipelined on the device:
// up to SASSY_HIP_PIPE_DEPTH (default 2, at most 4) searches are in flight, each on a lane (stream + buffers)
// of its own.  begin() queues the whole kernel chain of one search and returns at once; the filter of
// search i+1 starts when the filter of search i is done, so that the short, latency-bound tail of search i
// (chunk list, chunk DP, traceback) runs underneath the bandwidth-bound filter of search i+1.
int sassy_hip_search_shard_begin(sassy_SearcherType* s, const uint8_t* pattern, size_t pattern_len,
                                 const uint8_t* d_text, uint64_t halo_len, uint64_t shard_len,
                                 uint64_t global_offset, uint64_t total_len, size_t k, uint32_t flags,
                                 sassy_hip_Ticket** out) {
  if (!s || !pattern || !d_text || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  DeviceGuard on_device(s);
  if (halo_len % 64 || global_offset % 64) return fail(SASSY_HIP_EINVAL, "halo_len and global_offset must be multiples of 64");
  if (global_offset < halo_len) return fail(SASSY_HIP_EINVAL, "halo reaches left of the text start");
  if (global_offset + shard_len > total_len) return fail(SASSY_HIP_EINVAL, "shard exceeds the text");
  const bool is_first = global_offset == 0, is_last = global_offset + shard_len == total_len;
  if (!is_last && shard_len % 64) return fail(SASSY_HIP_EINVAL, "inner shard lengths must be multiples of 64");
  // (a halo that reaches byte 0 of the text is as long as a halo can be: a short text cut into many shards)
  if (!is_first && halo_len < sassy_hip_required_halo(pattern_len, k) && halo_len != global_offset)
    return fail(SASSY_HIP_EINVAL, "halo too short");
  if (((uintptr_t)d_text & 15) != 0) return fail(SASSY_HIP_EINVAL, "device text pointer must be 16-byte aligned");
  if (k > 0x7FFFFFFFu) return fail(SASSY_HIP_EINVAL, "k too large");
  const int depth = s->pipe_depth;
  int lane = -1;
  for (int l = 0; l < depth; ++l)
    if (!s->lane_ticket[(s->last_begun_lane + 1 + l) % depth]) { lane = (s->last_begun_lane + 1 + l) % depth; break; }
  if (lane < 0) return fail(SASSY_HIP_EINVAL, "too many searches in flight: finish one first (SASSY_HIP_PIPE_DEPTH)");
  std::unique_ptr<sassy_hip_Ticket> t(new sassy_hip_Ticket());
  std::string err;
  if (!make_plan(s->profile, pattern, pattern_len, t->plan, err)) return fail(SASSY_HIP_EINVAL, err);
  if (int rc = s->ensure_device()) return rc;
  t->owner = s;
  t->lane = lane;
  t->pat.assign(pattern, pattern + pattern_len);
  t->total_len = total_len;
  t->without_trace = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
  t->t0 = now_ms();
  t->empty_shard = shard_len == 0;
  if (!t->empty_shard) {
    ShardView sh{d_text, halo_len + shard_len, halo_len, global_offset - halo_len, global_offset == halo_len, is_last};
    sh.adopt_ok = true;
    auto job = std::make_shared<ScanJob>(s, s->lanes[lane], sh, t->plan, (uint32_t)k, (flags & SASSY_HIP_ALL_MINIMA) != 0,
                                         t->pat.data(), !t->without_trace, total_len);
    job->pipelined = depth > 1;
    job->signal_filter_done = true;
    // The searches in flight run freely side by side.  Two alternatives were measured and dropped (3 GB, two
    // searches in flight, 0.585 ms per search as it is): every filter waiting for the END of the previous one (0.64 --
    // the filters of two searches fill each other's ramp-up and drain, a strict sequence leaves those bubbles), and
    // the filter as two half launches with the next search waiting for the event in between (0.65).
    int rc = job->prepare();
    if (rc == 0 && !job->empty) rc = job->enqueue(0);
    if (rc != 0) {
      (void)hipStreamSynchronize(s->lanes[lane].stream);
      return rc;
    }
    t->job = job;
  }
  s->lane_ticket[lane] = t.get();
  s->last_begun_lane = lane;
  *out = t.release();
  return 0;
}

int sassy_hip_search_finish(sassy_SearcherType* s, sassy_hip_Ticket* t, sassy_hip_Result** out) {
  if (!s || !t || t->owner != s) return fail(SASSY_HIP_EINVAL, "not a ticket of this searcher");
  DeviceGuard on_device(s);
  std::unique_ptr<sassy_hip_Ticket> guard(t);
  s->lane_ticket[t->lane] = nullptr;
  reset_stats(s);
  std::unique_ptr<sassy_hip_Result> R(new sassy_hip_Result());
  if (t->job) {
    ScanJob* job = static_cast<ScanJob*>(t->job.get());
    ScanOut so;
    if (int rc = job->finish(so)) return rc;
    if (out) {
      size_t first = 0;
      if (int rc = append_matches(so, t->total_len, t->plan, t->without_trace, 0, R.get(), first)) return rc;
      R->exit_state = so.exit_state;
      R->conditional_index = so.conditional_index;
    }
  }
  if (R->pool.empty()) R->pool.push_back('\0');
  s->stats.total_ms = now_ms() - t->t0;
  s->stats.host_post_ms = std::max(0.0, s->stats.total_ms - s->stats.host_enqueue_ms - s->stats.host_wait_ms);
  if (out) *out = R.release();
  return 0;
}

size_t sassy_hip_result_len(const sassy_hip_Result* r) { return r ? r->size() : 0; }
const sassy_hip_Match* sassy_hip_result_matches(const sassy_hip_Result* r) { return r ? r->data() : nullptr; }
const char* sassy_hip_result_cigars(const sassy_hip_Result* r) { return r ? r->pool_data() : nullptr; }
size_t sassy_hip_result_cigars_len(const sassy_hip_Result* r) { return r ? r->pool_size() : 0; }

int sassy_hip_pack_rows(const sassy_hip_Match* matches, size_t n, const char* cigars, size_t cigars_len, int64_t* rows,
                        size_t cigar_bytes) {
  if ((n && (!matches || !rows)) || cigar_bytes % 8 != 0) return fail(SASSY_HIP_EINVAL, "bad argument");
  const size_t cols = 7 + cigar_bytes / 8;
  for (size_t i = 0; i < n; ++i) {
    const sassy_hip_Match& m = matches[i];
    int64_t* r = rows + i * cols;
    r[0] = (int64_t)m.pattern_idx;
    r[1] = (int64_t)m.text_start;
    r[2] = (int64_t)m.text_end;
    r[3] = (int64_t)m.pattern_start;
    r[4] = (int64_t)m.pattern_end;
    r[5] = m.cost;
    r[6] = m.strand;
    if (m.cigar_len > cigar_bytes || (m.cigar_len && (!cigars || (size_t)m.cigar_off + m.cigar_len > cigars_len)))
      return fail(SASSY_HIP_EINVAL, "cigar longer than the fixed gather field");
    char* c = reinterpret_cast<char*>(r + 7);
    if (m.cigar_len) memcpy(c, cigars + m.cigar_off, m.cigar_len);
    memset(c + m.cigar_len, 0, cigar_bytes - m.cigar_len);
  }
  return 0;
}
int sassy_hip_result_exit_state(const sassy_hip_Result* r) { return r ? r->exit_state : -1; }
int64_t sassy_hip_result_conditional_index(const sassy_hip_Result* r) { return r ? r->conditional_index : -1; }
void sassy_hip_result_free(sassy_hip_Result* r) { delete r; }

// ---- drop-in `search` (reference: c/sassy.h:52-58, src/c.rs:89-122) ----
// SASSY_HIP_DEVICES = "all" | "0,1,2,...": the drop-in search() cuts a host text into one shard per named device (a
// device may be named more than once), uploads the shards over all PCIe links at once, searches them at once and merges
// (sassy_hip_multi_*).  Unset, or a text of less than 4 MiB per device: the searcher's own device does it all.
static std::vector<int> drop_in_devices(const std::string& names) {
  std::vector<int> devs;
  const char* e = names.c_str();
  if (!*e) return devs;
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess) { (void)hipGetLastError(); return devs; }
  if (!strcmp(e, "all")) {
    for (int d = 0; d < visible; ++d) devs.push_back(d);
    if (devs.size() < 2) devs.clear();
    return devs;
  }
  for (const char* q = e; *q;) {
    char* end = nullptr;
    const long d = strtol(q, &end, 10);
    if (end == q || d < 0 || d >= visible) { devs.clear(); return devs; }
    devs.push_back((int)d);
    q = *end == ',' ? end + 1 : end;
    if (*end && *end != ',') { devs.clear(); return devs; }
  }
  return devs;
}

uintptr_t search(sassy_SearcherType* searcher, const uint8_t* pattern, uintptr_t pattern_len,
                 const uint8_t* text, uintptr_t text_len, uintptr_t k, sassy_Match** out_matches) {
  if (!searcher || !pattern || !text || !out_matches) die("Pointers in search() must not be null");
  sassy_hip_Result* R = nullptr;
  const std::vector<int> devs = searcher->sw.devices.empty() ? std::vector<int>() : drop_in_devices(searcher->sw.devices);
  const bool plain_modes = std::isnan(searcher->alpha) && std::isnan(searcher->max_n_frac) && !searcher->only_best &&
                           searcher->ref_lanes == 0 && !(searcher->rc && searcher->profile == PROFILE_ASCII);
  if (!devs.empty() && plain_modes && text_len >= devs.size() * (size_t)(4u << 20)) {
    if (!searcher->multi) {
      const char* names[] = {"ascii", "dna", "iupac"};
      sassy_hip_Multi* mm = sassy_hip_multi_new(names[(int)searcher->profile], NAN, devs.data(), devs.size());
      if (!mm) die(g_err.c_str());
      if (sassy_hip_multi_set_rc(mm, searcher->rc ? 1 : 0) != 0) die(g_err.c_str());
      searcher->multi = std::shared_ptr<void>(mm, [](void* q) { sassy_hip_multi_free(static_cast<sassy_hip_Multi*>(q)); });
    }
    sassy_hip_Multi* mm = static_cast<sassy_hip_Multi*>(searcher->multi.get());
    if (sassy_hip_multi_set_text(mm, text, text_len, pattern_len, k) != 0) die(g_err.c_str());
    if (sassy_hip_multi_search(mm, pattern, pattern_len, k, 0, &R) != 0) die(g_err.c_str());
  } else if (sassy_hip_search(searcher, pattern, pattern_len, text, text_len, k, 0, &R) != 0) die(g_err.c_str());
  const size_t n = R->matches.size();
  // never null, also for zero matches (the reference hands out a dangling non-null pointer and
  // sassy_matches_free asserts non-null: src/c.rs:112-127)
  sassy_Match* arr = static_cast<sassy_Match*>(std::malloc(std::max<size_t>(1, n) * sizeof(sassy_Match)));
  if (!arr) die("out of memory");
  for (size_t i = 0; i < n; ++i) {
    const sassy_hip_Match& m = R->matches[i];
    arr[i].text_start = (uintptr_t)m.text_start;
    arr[i].text_end = (uintptr_t)m.text_end;
    arr[i].pattern_start = (uintptr_t)m.pattern_start;
    arr[i].pattern_end = (uintptr_t)m.pattern_end;
    arr[i].cost = m.cost;
    arr[i].strand = m.strand;
  }
  sassy_hip_result_free(R);
  *out_matches = arr;
  return n;
}

void sassy_matches_free(sassy_Match* ptr, uintptr_t len) {
  (void)len;
  if (!ptr) die("Pointer to matches must not be null");  // src/c.rs:127
  std::free(ptr);
}

// ---- encoded patterns (reference: src/search.rs:404-423) ----
sassy_hip_Encoded* sassy_hip_encode_patterns(sassy_SearcherType* s, const uint8_t* patterns,
                                             size_t npat, size_t plen) {
  if (!s || !patterns) { fail(SASSY_HIP_EINVAL, "null argument"); return nullptr; }
  if (npat == 0) { fail(SASSY_HIP_EINVAL, "No queries provided"); return nullptr; }  // general.rs:250-252
  if (plen == 0 || plen > 64) {  // tqueries.rs:60-65, general.rs:285-291
    fail(SASSY_HIP_EINVAL, "Invalid pattern length (must be 1..=64)");
    return nullptr;
  }
  if (s->rc && s->profile == PROFILE_ASCII) {
    fail(SASSY_HIP_EUNSUPPORTED, "reverse complement is not defined for the ascii alphabet");
    return nullptr;
  }
  sassy_hip_Encoded* e = new sassy_hip_Encoded();
  e->profile = s->profile;
  e->rc = s->rc;
  e->plen = plen;
  e->n_original = npat;
  for (size_t p = 0; p < npat; ++p) e->patterns.emplace_back(patterns + p * plen, patterns + (p + 1) * plen);
  if (s->rc) {  // RC of every pattern appended as patterns P..2P (tqueries.rs:74-80)
    for (size_t p = 0; p < npat; ++p) {
      std::vector<uint8_t> r(plen);
      for (size_t i = 0; i < plen; ++i) r[i] = complement_char(PROFILE_IUPAC, patterns[p * plen + plen - 1 - i]);
      e->patterns.push_back(std::move(r));
    }
  }
  return e;
}
void sassy_hip_encoded_free(sassy_hip_Encoded* e) { delete e; }

int sassy_hip_search_encoded(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* text,
                             size_t text_len, size_t k, uint32_t flags, sassy_hip_Result** out) {
  if (!s || !e || (!text && text_len) || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_NO_TICKETS(s);
  DeviceGuard on_device(s);
  const double t0 = now_ms();
  reset_stats(s);
  if (int rc = s->ensure_device()) return rc;
  sassy_hip_Result* R = new sassy_hip_Result();
  std::unique_ptr<sassy_hip_Result> guard(R);
  // One forward scan per (rc-expanded) pattern over the device-resident text: the text goes to
  // the device once, every pattern reuses it.
  const uint8_t* d_text = text;
  uint32_t f = flags;
  if (!(flags & SASSY_HIP_TEXT_ON_DEVICE) && text_len) {
    if (int rc = s->d_text.reserve(text_len + 64)) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_text.p, text, text_len, hipMemcpyHostToDevice, s->stream));
  }
  const uint8_t* tptr = (flags & SASSY_HIP_TEXT_ON_DEVICE) ? d_text : s->d_text.p;
  if ((flags & SASSY_HIP_TEXT_ON_DEVICE) && text_len && ((uintptr_t)tptr & 15) != 0)
    return fail(SASSY_HIP_EINVAL, "device text pointer must be 16-byte aligned");  // the kernels load 16-byte chunks
  f |= SASSY_HIP_TEXT_ON_DEVICE;  // search_text must not upload the text again per pattern
  // Many plain-ACGT patterns on an Iupac searcher (the CRISPR-guide case): if the text is plain
  // ACGT as well, the Dna kernels give identical results and are cheaper -- test the text once.
  struct ProfileGuard {
    sassy_SearcherType* s; Profile saved;
    ~ProfileGuard() { s->profile = saved; }
  } pguard{s, s->profile};
  // (patterns with ambiguity letters -- guides with their NGG -- stay Iupac, but on a plain text the seeded search
  // takes them too: its seeds and masks are built from the letters' base sets)
  bool text_plain = false, text_checked = false;
  if (s->profile == PROFILE_IUPAC && std::isnan(s->alpha) && e->patterns.size() >= 4 && text_len >= 16 &&
      ((uintptr_t)tptr & 15) == 0) {
    bool plain = true;
    for (const auto& p : e->patterns) plain = plain && acgt_only(p.data(), p.size());
    if (plain || seeded_hit_rate(e->plen, k) > 0) {
      if (int rc = s->d_ncount.reserve(4)) return rc;
      HIP_TRY(hipMemsetAsync(s->d_ncount.p, 0, 4, s->stream));
      hipError_t le = launch_acgt_check(tptr, text_len, s->d_ncount.p, s->stream);
      if (le != hipSuccess) return hip_fail(le, "text check kernel launch");
      uint32_t bad = 1;
      HIP_TRY(hipMemcpyAsync(&bad, s->d_ncount.p, 4, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      text_plain = !bad;
      text_checked = true;
      if (plain && text_plain) s->profile = PROFILE_DNA;
    }
  }
  (void)f;
  if (text_len > 0) {
    if (k > 0x7FFFFFFFu) return fail(SASSY_HIP_EINVAL, "k too large");
    const bool all = (flags & SASSY_HIP_ALL_MINIMA) != 0;
    const bool wo = (flags & SASSY_HIP_WITHOUT_TRACE) != 0;
    const uint8_t* h_text = (flags & SASSY_HIP_TEXT_ON_DEVICE) ? nullptr : text;
    // one forward scan per pattern, several in flight (ScanQueue); results in pattern order
    ScanQueue queue(s, [&](uint64_t p, ScanOut& so, const PatternPlan& plan, const uint8_t* pat) -> int {
      if (int rc = post_filter(s, so, plan, pat, (uint32_t)k, 0, h_text, tptr, text_len, !wo, EndFilter())) return rc;
      size_t first = 0;
      if (int rc = append_matches(so, text_len, plan, wo, p % e->n_original, R, first)) return rc;
      for (size_t i = first; i < R->matches.size(); ++i) R->matches[i].strand = p >= e->n_original ? 1 : 0;
      return 0;
    });
    std::string err;
    // Many Dna patterns over a long text: one multi-pattern prefilter pass per batch of patterns
    // (filter_dna_multi_kernel), then chunk list -> DP -> rank -> traceback per pattern.
    const uint64_t multi_min = (uint64_t)std::max<long>(1, s->sw.multi_min_text);
    const uint32_t mq = (uint32_t)std::min<size_t>(e->plen / (k + 1), 12);
    const bool multi = s->profile == PROFILE_DNA && std::isnan(s->alpha) && e->patterns.size() >= 8 && k + 1 <= 8 &&
                       mq >= 6 && text_len >= multi_min && (((uintptr_t)tptr) & 15) == 0;
    // Many patterns: the pattern-tiled scan does all of them in one pass, where one scan per pattern pays a kernel
    // chain each.  Measured (tools/bench_encoded.py, config 4's shape; profiles/r02_encoded_paths.txt): the tiled
    // kernel advances 3.8e10 (character x group of 64 patterns) per second; a chain costs ~60 us, or 40 us +
    // 7.8e-14 s per text byte behind the multi-pattern prefilter (long plain-ACGT texts), or 60 us + 5.5e-13 s
    // per byte with its own filter pass (texts with other letters).  SASSY_HIP_TILED=0 / 1 forces the choice.
    const int env_tiled = (int)s->sw.tiled;
    const bool tiled_ok = s->profile != PROFILE_ASCII && std::isnan(s->alpha) && 2 * k + 3 <= 64 &&
                          e->patterns.size() < (1u << 24) && text_len < (1ull << 40);
    const uint64_t tiled_groups = (e->patterns.size() + 63) / 64;
    const double est_tiled = (double)text_len * (double)tiled_groups / 3.8e10 + 1e-4;
    const double est_chains = (double)e->patterns.size() *
        (multi ? 40e-6 + 7.8e-14 * (double)text_len
               : 60e-6 + (text_len >= multi_min ? 5.5e-13 * (double)text_len : 0.0));
    bool tiled = tiled_ok && e->patterns.size() >= 2 && est_tiled <= est_chains;
    if (env_tiled >= 0) tiled = tiled_ok && env_tiled != 0;
    // Many patterns, long text, selective pieces: seed -> verify -> report (seed_kernels.hip) reads the text
    // once for all patterns.  Expected cost per (character, pattern): hit rate x window x ~24 operations, against
    // 17 for the pattern-tiled scan (SASSY_HIP_SEEDED=0 / 1 forces the choice).
    const int env_seeded = (int)s->sw.seeded;
    bool seeded = false;
    // (an Iupac searcher's text with other letters: the seeded search plus the pattern-tiled scan around those letters,
    // search_encoded_seeded / seeded_dirty_zones; SASSY_HIP_SEEDED_DIRTY=0: not for such texts)
    const bool env_dirty = true;
    const bool dirty_text = s->profile == PROFILE_IUPAC && text_checked && !text_plain;
    if ((s->profile == PROFILE_DNA || (s->profile == PROFILE_IUPAC && text_checked && (text_plain || env_dirty))) &&
        std::isnan(s->alpha) && k + 1 <= 8 &&
        e->plen / (k + 1) >= 5 &&
        e->plen + 3 * k + 1 <= 4 * kSeedWindowDwords && e->patterns.size() < (1u << 24) && text_len < (1ull << 36) &&
        (((uintptr_t)tptr) & 15) == 0) {
      const double est_seeded = seeded_estimate(e->plen, k, e->patterns.size(), text_len);
      seeded = est_seeded < est_chains && (!tiled || est_seeded < est_tiled);
      if (env_seeded >= 0) seeded = env_seeded != 0;
    }
    bool tiled_done = false;
    // an overhang searcher: one pass where the batch path's kernels apply to this text as a batch of one
    if (!std::isnan(s->alpha))
      if (int rc = search_encoded_overhang(s, e, tptr, h_text, text_len, (uint32_t)k, all, wo, R, &tiled_done)) return rc;
    if (seeded) {
      if (int rc = search_encoded_seeded(s, e, tptr, h_text, text_len, (uint32_t)k, all, wo, R, &tiled_done, nullptr, nullptr,
                                         dirty_text)) return rc;
      if (tiled_done) tiled = false;
    }
    if (tiled && !tiled_done)
      if (int rc = search_encoded_tiled(s, e, tptr, h_text, text_len, (uint32_t)k, all, wo, R, &tiled_done)) return rc;
    const size_t batch = multi ? 64 : 1;
    for (size_t p0 = 0; p0 < (tiled_done ? 0 : e->patterns.size()); p0 += batch) {
      const size_t nb = std::min(batch, e->patterns.size() - p0);
      unsigned long long* bm_base = nullptr;
      uint64_t bm_stride = 0;
      if (multi) {
        const uint64_t n_blocks = (text_len + 63) / 64;
        bm_stride = ((n_blocks + 63) / 64 + 2 + 7) / 8 * 8;
        if (int rc = s->d_multi_bitmap.reserve(nb * bm_stride)) return rc;
        if (int rc = s->d_multi_bits.reserve(16 * nb)) return rc;
        bm_base = s->d_multi_bitmap.p;
        // piece p covers rows [start, start + len): len = q + 1 for the first m mod (k+1) pieces (the spare
        // rows make those pieces more selective), q otherwise; bit d of a piece word = code bit of the
        // row at distance d from the piece's end
        const uint32_t spare = (uint32_t)(e->plen - (size_t)mq * (k + 1));
        uint32_t p_start[8], p_len[8], long_mask = 0;
        for (uint32_t pp = 0; pp < 8; ++pp) {
          const uint32_t pc = std::min<uint32_t>(pp, (uint32_t)k);
          p_len[pp] = mq + (pc < spare ? 1u : 0u);
          p_start[pp] = pc * mq + std::min(pc, spare);
          if (pp <= k && pc < spare) long_mask |= 1u << pp;
        }
        std::vector<uint32_t> bits(16 * nb, 0u);
        for (size_t i = 0; i < nb; ++i) {
          const uint8_t* pt = e->patterns[p0 + i].data();
          for (uint32_t pp = 0; pp < k + 1; ++pp)
            for (uint32_t d = 0; d < p_len[pp]; ++d) {
              const uint32_t code = (pt[p_start[pp] + p_len[pp] - 1 - d] >> 1) & 3u;  // src/profiles/dna.rs:19-40
              bits[16 * i + 2 * pp] |= (code & 1u) << d;
              bits[16 * i + 2 * pp + 1] |= (code >> 1) << d;
            }
        }
        ScanParams F{};
        F.text = tptr;
        F.text_len = text_len;
        F.n_blocks = n_blocks;
        F.first_owned_block = 0;
        F.m = (uint32_t)e->plen;
        F.k = (uint32_t)k;
        F.n_pieces = (uint32_t)k + 1;
        F.piece_len = mq;
        for (uint32_t pp = 0; pp < 8; ++pp) F.piece_rem[pp] = (uint32_t)e->plen - (p_start[pp] + p_len[pp]);
        F.multi_long = long_mask;
        F.stage_blocks = 2;
        F.lds_per_wave = 4096u * 2;
        F.hit_bitmap = bm_base;
        F.multi_bits = s->d_multi_bits.p;
        F.multi_n = (uint32_t)nb;
        F.multi_stride = bm_stride;
        uint32_t fgrid = 0;
        // register-heavy kernel (two blocks of shifted planes): 3 resident waves per SIMD up to q = 8, 2 above
        if (int rc = stream_geometry(F, n_blocks, 1, &fgrid, mq <= 8 ? 12 : 8)) return rc;
        HIP_TRY(hipMemsetAsync(bm_base, 0, nb * bm_stride * 8, s->stream));
        HIP_TRY(hipMemcpyAsync(s->d_multi_bits.p, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipEventRecord(s->ev_a_multi(), s->stream));
        hipError_t le = launch_filter_dna_multi(F, fgrid, s->stream);
        if (le != hipSuccess) return hip_fail(le, "multi-pattern filter launch");
        HIP_TRY(hipEventRecord(s->ev_multi, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));  // `bits` must outlive the upload; also times the pass
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_a_multi(), s->ev_multi));
        s->stats.filter_ms += ms;
      }
      for (size_t i = 0; i < nb; ++i) {
        const size_t p = p0 + i;
        PatternPlan plan;
        if (!make_plan(s->profile, e->patterns[p].data(), e->plen, plan, err)) return fail(SASSY_HIP_EINVAL, err);
        ShardView sh{tptr, text_len, 0, 0, true, true};
        if (int rc = queue.submit(plan, e->patterns[p].data(), sh, TextTable{}, (uint32_t)k, all, !wo, text_len, p,
                                  multi ? bm_base + i * bm_stride : nullptr, multi ? mq : 0, nullptr)) return rc;
      }
      if (multi)  // the bitmaps are reused by the next batch
        if (int rc = queue.drain_all()) return rc;
    }
    if (int rc = queue.drain_all()) return rc;
  }
  // The reference's order is an artefact of its range bookkeeping; its own differential test
  // sorts by this key before comparing (pattern_tiling/search.rs:748-757).
  if (R->pool.empty()) R->pool.push_back('\0');
  const char* pool = R->pool.c_str();
  auto before = [pool](const sassy_hip_Match& a, const sassy_hip_Match& b) {
    if (a.pattern_idx != b.pattern_idx) return a.pattern_idx < b.pattern_idx;
    if (a.text_start != b.text_start) return a.text_start < b.text_start;
    if (a.text_end != b.text_end) return a.text_end < b.text_end;
    if (a.cost != b.cost) return a.cost < b.cost;
    if (a.strand != b.strand) return a.strand < b.strand;
    return strcmp(pool + a.cigar_off, pool + b.cigar_off) < 0;
  };
  // (the one-pass paths deliver the records pattern by pattern in position order, the Rc strand's behind the forward
  // strand's: already in this order, or two runs that are -- one linear merge instead of a sort of millions of records)
  {
    auto mid = std::is_sorted_until(R->matches.begin(), R->matches.end(), before);
    if (mid != R->matches.end()) {
      if (std::is_sorted(mid, R->matches.end(), before)) std::inplace_merge(R->matches.begin(), mid, R->matches.end(), before);
      else std::sort(R->matches.begin(), R->matches.end(), before);
    }
  }
  guard.release();
  s->stats.total_ms = now_ms() - t0;
  s->stats.host_post_ms = s->stats.total_ms - s->stats.host_enqueue_ms - s->stats.host_wait_ms;
  *out = R;
  return 0;
}

// ---- synthetic inputs ----
int sassy_hip_generate_dna(uint8_t* d_text, uint64_t n, uint64_t seed, uint64_t first, void* hip_stream) {
  if (!d_text && n) return fail(SASSY_HIP_EINVAL, "null argument");
  hipError_t e = launch_generate_dna(d_text, n, seed, first, reinterpret_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "generate kernel launch");
  HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream)));
  return 0;
}

int sassy_hip_generate_genome_like(uint8_t* d_text, uint64_t n, uint64_t seed, uint64_t first, int with_n, void* hip_stream) {
  if (!d_text && n) return fail(SASSY_HIP_EINVAL, "null argument");
  hipError_t e = launch_generate_genome_like(d_text, n, seed, first, with_n, reinterpret_cast<hipStream_t>(hip_stream));
  if (e != hipSuccess) return hip_fail(e, "generate kernel launch");
  HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream)));
  return 0;
}

static inline uint64_t splitmix64_host(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline uint64_t hash_host(uint64_t seed, uint64_t idx) { return splitmix64_host(seed * 0x9E3779B97F4A7C15ull + idx); }

// Plant q = the pattern with (q mod (k+1)) edits drawn from the counter-based hash:
// r = hash(seed ^ "plant", 64*q + t); type = r % 3 (0 sub, 1 ins, 2 del); pos = (r >> 8) % len;
// base = (r >> 40) & 3.  The CPU twin used by the tests is oracle/sassy_oracle.c:orc_make_plant.
static std::vector<uint8_t> make_plant(uint64_t seed, uint64_t q, const uint8_t* pat, size_t m, int edits) {
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  std::vector<uint8_t> s(pat, pat + m);
  for (int t = 0; t < edits; ++t) {
    const uint64_t r = hash_host(seed ^ 0x706c616e74ull, 64 * q + (uint64_t)t);
    const int type = (int)(r % 3);
    const size_t pos = (size_t)((r >> 8) % s.size());
    const int b = (int)((r >> 40) & 3);
    if (type == 0) {
      int idx = 0;
      for (int a = 0; a < 4; ++a)
        if (s[pos] == (uint8_t)acgt[a]) idx = a;
      s[pos] = (uint8_t)acgt[(idx + 1 + (b % 3)) & 3];
    } else if (type == 1) {
      s.insert(s.begin() + (long)pos, (uint8_t)acgt[b]);
    } else if (s.size() > 1) {
      s.erase(s.begin() + (long)pos);
    }
  }
  return s;
}

int sassy_hip_plant(uint8_t* d_text, uint64_t n, uint64_t first, uint64_t total_n, uint64_t seed,
                    const uint8_t* pattern, size_t pattern_len, size_t k, uint64_t stride,
                    void* hip_stream, uint64_t* planted) {
  return sassy_hip_plant_phase(d_text, n, first, total_n, seed, pattern, pattern_len, k, stride, 0, hip_stream, planted);
}
// the same with the plants `phase` bytes further on (q * stride + stride / 2 + phase): several patterns planted into one text
// (bench.py: the searches in flight look for different patterns)
int sassy_hip_plant_phase(uint8_t* d_text, uint64_t n, uint64_t first, uint64_t total_n, uint64_t seed,
                          const uint8_t* pattern, size_t pattern_len, size_t k, uint64_t stride, uint64_t phase,
                          void* hip_stream, uint64_t* planted) {
  if (!d_text || !pattern || stride == 0) return fail(SASSY_HIP_EINVAL, "bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  std::vector<uint64_t> pos;
  std::vector<uint8_t> val;
  uint64_t cnt = 0;
  for (uint64_t q = 0;; ++q) {
    const uint64_t p = q * stride + stride / 2 + phase;
    if (p + pattern_len + k > total_n) break;
    if (p >= first + n) break;
    std::vector<uint8_t> s = make_plant(seed, q, pattern, pattern_len, (int)(q % (k + 1)));
    if (p + s.size() <= first) continue;
    for (size_t i = 0; i < s.size(); ++i) {
      const uint64_t g = p + i;
      if (g >= first && g < first + n) { pos.push_back(g); val.push_back(s[i]); }
    }
    cnt++;
  }
  if (planted) *planted = cnt;
  if (pos.empty()) return 0;
  uint64_t* d_pos = nullptr;
  uint8_t* d_val = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_pos), pos.size() * sizeof(uint64_t)));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_val), val.size());
  if (e != hipSuccess) { (void)hipFree(d_pos); return hip_fail(e, "hipMalloc"); }
  int rc = 0;
  do {
    if ((e = hipMemcpyAsync(d_pos, pos.data(), pos.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(d_val, val.data(), val.size(), hipMemcpyHostToDevice, st)) != hipSuccess) break;
    if ((e = launch_scatter_bytes(d_text, n, first, d_pos, d_val, pos.size(), st)) != hipSuccess) break;
    e = hipStreamSynchronize(st);
  } while (0);
  if (e != hipSuccess) rc = hip_fail(e, "plant");
  (void)hipFree(d_pos);
  (void)hipFree(d_val);
  return rc;
}

// ---- plain device memory helpers ----
void* sassy_hip_malloc(size_t bytes) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) { hip_fail(e, "hipMalloc"); return nullptr; }
  return p;
}
void sassy_hip_free(void* d_ptr) { if (d_ptr) (void)hipFree(d_ptr); }
int sassy_hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
  HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int sassy_hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
  HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"

