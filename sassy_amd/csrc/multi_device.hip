// multi_device.hip -- sassy_hip_multi_*: one text over several devices inside one process (a worker thread, a bound
// searcher and a resident shard per device; reference: the thread fan-out of bin/grep.rs:476-537).
#include "host_internal.h"

using namespace sassy_hip;

// ---- one text over several devices, inside one process (reference: the thread fan-out of bin/grep.rs:476-503) ----
// A worker thread per device, alive as long as the multi-searcher: HIP's current device is per thread, and a search
// of a resident text takes less time than starting a thread.
struct MultiWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, quit = false;
  void start(int device) {
    th = std::thread([this, device] {
      (void)hipSetDevice(device);
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        j();
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

struct sassy_hip_Multi {
  struct Part {
    int device = 0;
    sassy_SearcherType* searcher = nullptr;
    sassy_SearcherType* searcher_rc = nullptr;  // (both strands in one call: search_encoded / search_many with rc)
    uint8_t* d_text = nullptr;       // halo first
    size_t d_cap = 0;
    uint8_t* d_rev = nullptr;        // Rc strand: the reversed view of this part's share of the reversed text
    size_t d_rev_cap = 0;
    bool rev_valid = false;          // d_rev holds the reverse of the resident bytes as they are now (built once per text)
    sassy_SearcherType* searcher_rs = nullptr;  // searches in flight: the Rc strand's shard search has lanes of its own
    uint64_t halo = 0, len = 0, offset = 0;  // bytes in front of the shard, shard length, its global offset
    uint64_t halo_r = 0;             // bytes of text kept behind the shard (the Rc strand's halo lies on that side)
    std::unique_ptr<MultiWorker> worker;
    int open_tickets = 0;            // searches begun and not yet finished (sassy_hip_multi_search_begin)
    int rc = 0;
    std::string err;
    sassy_hip_Result* result = nullptr;
    sassy_hip_Result* result_rc = nullptr;
  };
  std::vector<Part> parts;
  std::string alphabet;
  float alpha = NAN;
  uint64_t total_len = 0;
  uint64_t halo_for = 0;  // the resident shards carry halos good for searches with required_halo(m, k) <= this
  bool have_text = false;
  int pipe_depth = 3;       // searches in flight per device (sassy_hip_multi_set_pipe_depth)
  bool rc = false;          // sassy_hip_multi_set_rc: searches return both strands
  bool replicate = false;   // sassy_hip_multi_set_replicated: every device holds the WHOLE text (patterns are sharded)
  ~sassy_hip_Multi() {
    for (Part& p : parts) {
      if (p.worker) p.worker->stop();
      int prev = 0;
      (void)hipGetDevice(&prev);
      (void)hipSetDevice(p.device);
      if (p.d_text) (void)hipFree(p.d_text);
      if (p.d_rev) (void)hipFree(p.d_rev);
      if (p.searcher) delete p.searcher;
      if (p.searcher_rc) delete p.searcher_rc;
      if (p.searcher_rs) delete p.searcher_rs;
      (void)hipSetDevice(prev);
    }
  }
  // runs f(part) on every part's worker thread (on its device) and waits for all of them; first error wins
  int on_all(const std::function<int(Part&)>& f) {
    for (Part& p : parts) {
      Part* pp = &p;
      p.worker->submit([pp, &f] {
        pp->rc = f(*pp);
        pp->err = pp->rc ? g_err : std::string();  // (g_err is thread-local: carry it over to the caller's thread)
      });
    }
    int first = 0;
    for (Part& p : parts) {
      p.worker->wait();
      if (p.rc && !first) { first = p.rc; g_err = "device " + std::to_string(p.device) + ": " + p.err; }
    }
    return first;
  }
  // [a, b) of shard i: equal shares of whole 64-byte blocks (sassy_amd/multigpu.py: shard_bounds)
  void bounds(size_t i, uint64_t& a, uint64_t& b) const { multi_bounds(total_len, eff_parts(), i, a, b); }
  // how many of the parts get a share: all of them, unless the text is so short that a share would be smaller than the
  // slack between forward and reversed shard borders (a trailing part without bytes cannot hold its share of the
  // reversed text) -- such a text is one device's
  size_t eff_parts() const { return multi_eff_parts(total_len, parts.size()); }
  static size_t multi_eff_parts(uint64_t len, size_t n) { return (n <= 1 || len < 64ull * n * (n + 2)) ? 1 : n; }
  static void multi_bounds(uint64_t len, size_t n, size_t i, uint64_t& a, uint64_t& b) {
    if (i >= n) { a = b = len; return; }
    uint64_t per = (len + n - 1) / n;
    per = (per + 63) / 64 * 64;
    a = std::min<uint64_t>(i * per, len);
    b = std::min<uint64_t>((i + 1) * per, len);
  }
  // The Rc strand is complement(pattern) against the REVERSED text (src/search.rs:813-878), sharded like the forward
  // one but in reversed coordinates: reversed shard j owns the reversed end positions (A, B] with A = j * per -- the
  // forward bytes [n - B, n - A), whose borders differ from the forward shards' by up to 64 * parts bytes unless n is a
  // multiple of 64 * parts.  Part i keeps reversed shard parts - 1 - i: it needs a few bytes more of text on either side.
  uint64_t slack() const { return 64ull * (eff_parts() + 1); }
  int layout(uint64_t len, size_t max_m, size_t max_k) {
    total_len = len;
    halo_for = sassy_hip_required_halo(max_m, max_k);
    for (size_t i = 0; i < parts.size(); ++i) {
      uint64_t a, b;
      bounds(i, a, b);
      if (replicate) { a = 0; b = len; }
      parts[i].offset = a;
      parts[i].len = b - a;
      parts[i].halo = (i == 0 || a == 0) ? 0 : std::min<uint64_t>(halo_for + slack(), a);
      parts[i].halo = parts[i].halo / 64 * 64;
      parts[i].halo_r = std::min<uint64_t>(halo_for + slack(), len - b);
      parts[i].rev_valid = false;
    }
    return 0;
  }
  // reversed shard `j` in forward coordinates: [fa, fb) and its halo [fb, fb + hrev) (hrev a multiple of 64)
  void rev_bounds(size_t j, uint64_t& fa, uint64_t& fb, uint64_t& hrev) const {
    uint64_t A, B;
    bounds(j, A, B);
    fa = total_len - B;
    fb = total_len - A;
    hrev = std::min<uint64_t>(halo_for, A);  // (A = n - fb bytes lie behind fb; A and halo_for are multiples of 64)
  }
  static int reserve(Part& p, size_t bytes) {
    if (bytes <= p.d_cap) return 0;
    if (p.d_text) (void)hipFree(p.d_text);
    p.d_text = nullptr;
    p.d_cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p.d_text), bytes + 256);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (text shard)");
    p.d_cap = bytes;
    return 0;
  }
};

// Searches begun with sassy_hip_multi_search_begin read the parts' resident shards (and their cached reversed copies) and
// use the parts' searchers until they are finished: nothing that rewrites, reallocates or re-lays-out those buffers, and
// no synchronous search on those searchers, may run in between.
static bool multi_tickets_open(const sassy_hip_Multi* m) {
  for (const sassy_hip_Multi::Part& p : m->parts)
    if (p.open_tickets) return true;
  return false;
}
#define SASSY_MULTI_NO_TICKETS(m)                                                                                        \
  do {                                                                                                                   \
    if (multi_tickets_open(m))                                                                                           \
      return fail(SASSY_HIP_EINVAL, "searches are in flight on this multi-searcher (sassy_hip_multi_search_begin): finish them first"); \
  } while (0)

extern "C" {

sassy_hip_Multi* sassy_hip_multi_new(const char* alphabet, float alpha, const int* devices, size_t n_devices) {
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
    (void)hipGetLastError();
    fail(SASSY_HIP_ENODEVICE, "no usable HIP device (libsassy_hip has no CPU fallback)");
    return nullptr;
  }
  std::vector<int> devs;
  if (devices && n_devices) devs.assign(devices, devices + n_devices);
  else for (int d = 0; d < visible; ++d) devs.push_back(d);
  for (int d : devs)
    if (d < 0 || d >= visible) { fail(SASSY_HIP_EINVAL, "no such HIP device"); return nullptr; }
  std::unique_ptr<sassy_hip_Multi> M(new sassy_hip_Multi());
  M->alphabet = alphabet ? alphabet : "";
  M->alpha = alpha;
  M->parts.resize(devs.size());
  for (size_t i = 0; i < devs.size(); ++i) {
    sassy_hip_Multi::Part& p = M->parts[i];
    p.device = devs[i];
    p.searcher = sassy_hip_searcher_new(alphabet, false, alpha);  // (shards are forward searches, like sassy_hip_search_shard)
    if (!p.searcher) return nullptr;
    p.searcher->device = devs[i];
    p.worker.reset(new MultiWorker());
    p.worker->start(devs[i]);
  }
  return M.release();
}

size_t sassy_hip_multi_shards(const sassy_hip_Multi* m) { return m ? m->parts.size() : 0; }
int sassy_hip_multi_device(const sassy_hip_Multi* m, size_t shard) { return (m && shard < m->parts.size()) ? m->parts[shard].device : -1; }
sassy_SearcherType* sassy_hip_multi_searcher(sassy_hip_Multi* m, size_t shard) {
  return (m && shard < m->parts.size()) ? m->parts[shard].searcher : nullptr;
}

int sassy_hip_multi_set_text(sassy_hip_Multi* m, const uint8_t* text, size_t len, size_t max_pattern_len, size_t max_k) {
  if (!m || (!text && len)) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_MULTI_NO_TICKETS(m);
  m->layout(len, max_pattern_len, max_k);
  m->have_text = false;
  // every device fetches its own shard (halo included) over its own PCIe link, all at the same time
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t bytes = (size_t)(p.halo + p.len + p.halo_r);
    if (int r = sassy_hip_Multi::reserve(p, bytes)) return r;
    if (bytes) HIP_TRY(hipMemcpy(p.d_text, text + (p.offset - p.halo), bytes, hipMemcpyHostToDevice));
    return 0;
  });
  if (rc) return rc;
  m->have_text = true;
  return 0;
}

int sassy_hip_multi_generate_dna(sassy_hip_Multi* m, uint64_t len, uint64_t seed, size_t max_pattern_len, size_t max_k) {
  if (!m) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_MULTI_NO_TICKETS(m);
  m->layout(len, max_pattern_len, max_k);
  m->have_text = false;
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t bytes = (size_t)(p.halo + p.len + p.halo_r);
    if (int r = sassy_hip_Multi::reserve(p, bytes)) return r;
    if (bytes)
      if (int r = sassy_hip_generate_dna(p.d_text, bytes, seed, p.offset - p.halo, nullptr)) return r;
    HIP_TRY(hipDeviceSynchronize());
    return 0;
  });
  if (rc) return rc;
  m->have_text = true;
  return 0;
}

int sassy_hip_multi_plant(sassy_hip_Multi* m, uint64_t seed, const uint8_t* pattern, size_t pattern_len, size_t k, uint64_t stride,
                          uint64_t* planted) {
  if (!m || !pattern || !m->have_text) return fail(SASSY_HIP_EINVAL, "no resident text");
  SASSY_MULTI_NO_TICKETS(m);
  std::vector<uint64_t> cnt(m->parts.size(), 0);
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t i = (size_t)(&p - m->parts.data());
    const size_t bytes = (size_t)(p.halo + p.len + p.halo_r);
    if (!bytes) return 0;
    p.rev_valid = false;  // (the text changes under the cached reversed copy)
    if (int r = sassy_hip_plant(p.d_text, bytes, p.offset - p.halo, m->total_len, seed, pattern, pattern_len, k, stride, nullptr, &cnt[i]))
      return r;
    HIP_TRY(hipDeviceSynchronize());
    return 0;
  });
  if (rc) return rc;
  if (planted) {
    // sassy_hip_plant's own rule over the whole text (the per-shard counts overlap in the halos): plant q exists while
    // q * stride + stride / 2 + m + k <= total_len
    const uint64_t need = stride / 2 + pattern_len + k;
    *planted = m->total_len < need ? 0 : (m->total_len - need) / stride + 1;
  }
  return 0;
}

int sassy_hip_multi_set_rc(sassy_hip_Multi* m, int rc) {
  if (!m) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_MULTI_NO_TICKETS(m);
  Profile pr;
  if (rc && parse_alphabet(m->alphabet.c_str(), pr) && pr == PROFILE_ASCII)
    return fail(SASSY_HIP_EUNSUPPORTED, "reverse complement is not defined for the ascii alphabet");
  m->rc = rc != 0;
  return 0;
}
int sassy_hip_multi_set_replicated(sassy_hip_Multi* m, int on) {
  if (!m) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_MULTI_NO_TICKETS(m);
  if (m->replicate != (on != 0)) m->have_text = false;  // (the resident buffers were laid out the other way)
  m->replicate = on != 0;
  return 0;
}

// The Rc strand of one part: reversed shard j = E - 1 - i of the reversed text (E = the parts that hold a share), read off
// the part's resident forward bytes by the reverse kernel ONCE per resident text (rev_valid; the copy is made on the
// searcher's stream and waited for there -- not on the null stream, which every blocking stream of the device would
// wait behind), searched with complement(pattern) like any shard (reversed coordinates).
// Returns 1 when the part has no share of the reversed text.
static int multi_rc_prepare(sassy_hip_Multi* m, sassy_hip_Multi::Part& p, uint64_t* A_out, uint64_t* B_out, uint64_t* hrev_out) {
  const size_t i = (size_t)(&p - m->parts.data());
  const size_t E = m->eff_parts();
  if (i >= E) return 1;
  const size_t j = E - 1 - i;
  uint64_t fa, fb, hrev;
  m->rev_bounds(j, fa, fb, hrev);
  if (fb <= fa) return 1;
  m->bounds(j, *A_out, *B_out);
  *hrev_out = hrev;
  if (p.rev_valid) return 0;
  const uint64_t buf0 = p.offset - p.halo, buf1 = p.offset + p.len + p.halo_r;  // the resident bytes [buf0, buf1)
  const uint64_t fa16 = fa / 16 * 16, end = fb + hrev;
  if (fa16 < buf0 || end > buf1)
    return fail(SASSY_HIP_EINVAL, "internal: the part's resident text does not cover its share of the reversed text");
  const size_t nrev = (size_t)(end - fa16);
  if (nrev + 256 > p.d_rev_cap) {
    if (p.d_rev) (void)hipFree(p.d_rev);
    p.d_rev = nullptr;
    p.d_rev_cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p.d_rev), nrev + 512);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (reversed shard)");
    p.d_rev_cap = nrev + 256;
  }
  // reverse(text[fa16, end)): its first (end - fa) bytes are the reversed shard with its halo in front; the up to 15
  // bytes behind them (the alignment the reverse kernel wants) are only there
  if (int rc = p.searcher->ensure_device()) return rc;
  hipError_t le = launch_reverse(p.d_text + (fa16 - buf0), p.d_rev, nrev, p.searcher->stream);
  if (le != hipSuccess) return hip_fail(le, "reverse kernel launch");
  HIP_TRY(hipStreamSynchronize(p.searcher->stream));
  p.rev_valid = true;
  return 0;
}
static int multi_rc_shard(sassy_hip_Multi* m, sassy_hip_Multi::Part& p, const uint8_t* cpat, size_t plen, size_t k, uint32_t f) {
  uint64_t A = 0, B = 0, hrev = 0;
  p.result_rc = nullptr;
  const int pr = multi_rc_prepare(m, p, &A, &B, &hrev);
  if (pr == 1) { p.result_rc = new sassy_hip_Result(); return 0; }
  if (pr) return pr;
  return sassy_hip_search_shard(p.searcher, cpat, plen, p.d_rev, hrev, B - A, A, m->total_len, k, f, &p.result_rc);
}

// the parts' shard results (p.result, and p.result_rc with both strands) -> one result; the parts' results are freed
static int multi_merge(sassy_hip_Multi* m, uint32_t f, int rc, sassy_hip_Result** out) {
  int mrc = rc;
  sassy_hip_Result* fwd = nullptr;
  sassy_hip_Result* rev = nullptr;
  if (!mrc) {
    std::vector<const sassy_hip_Result*> rs;
    for (sassy_hip_Multi::Part& p : m->parts) rs.push_back(p.result);
    mrc = sassy_hip_merge_shards(rs.data(), rs.size(), kStateDecTrue, &fwd);
  }
  if (!mrc && m->rc) {  // reversed shard j lives on part E - 1 - j
    std::vector<const sassy_hip_Result*> rs;
    const size_t E = m->eff_parts();
    for (size_t j = 0; j < E; ++j) rs.push_back(m->parts[E - 1 - j].result_rc);
    mrc = sassy_hip_merge_shards(rs.data(), rs.size(), kStateDecTrue, &rev);
  }
  for (sassy_hip_Multi::Part& p : m->parts) {
    delete p.result; p.result = nullptr;
    delete p.result_rc; p.result_rc = nullptr;
  }
  if (!mrc && rev) {
    // the Rc strand's matches behind the forward ones, mapped back to forward coordinates (src/search.rs:868-873)
    const bool wo = (f & SASSY_HIP_WITHOUT_TRACE) != 0;
    const size_t base = fwd->pool.size();
    if (base + rev->pool_size() > 0xFFFFFFFFull) mrc = fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB");
    else {
      fwd->pool.append(rev->pool_data(), rev->pool_size());
      const sassy_hip_Match* rm = rev->data();
      for (size_t i = 0; i < rev->size(); ++i) {
        sassy_hip_Match r = rm[i];
        const uint64_t rs_ = r.text_start, re = r.text_end;
        r.strand = 1;
        r.text_start = m->total_len - re;
        r.text_end = wo ? UINT64_MAX : m->total_len - rs_;
        r.cigar_off = (uint32_t)(r.cigar_off + base);
        fwd->matches.push_back(r);
      }
    }
  }
  delete rev;
  if (mrc) { delete fwd; return mrc; }
  *out = fwd;
  return 0;
}

static int multi_check_search(sassy_hip_Multi* m, const uint8_t* pattern, size_t pattern_len, size_t k, const void* out,
                              std::vector<uint8_t>& cp) {
  if (!m || !pattern || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  if (!m->have_text) return fail(SASSY_HIP_EINVAL, "no resident text (sassy_hip_multi_set_text)");
  if (m->replicate) return fail(SASSY_HIP_EINVAL, "the devices hold whole copies of the text (sassy_hip_multi_set_replicated): search_encoded only");
  if (sassy_hip_required_halo(pattern_len, k) > m->halo_for && m->eff_parts() > 1)
    return fail(SASSY_HIP_EINVAL, "the resident shards' halos are too short for this pattern length and k");
  if (m->rc) {  // complement(pattern) for the Rc strand (src/search.rs:813-820)
    Profile pr;
    if (!parse_alphabet(m->alphabet.c_str(), pr)) return fail(SASSY_HIP_EINVAL, "unknown alphabet");
    cp.resize(pattern_len);
    for (size_t i = 0; i < pattern_len; ++i) cp[i] = complement_char(pr, pattern[i]);
  }
  return 0;
}

int sassy_hip_multi_search(sassy_hip_Multi* m, const uint8_t* pattern, size_t pattern_len, size_t k, uint32_t flags,
                           sassy_hip_Result** out) {
  std::vector<uint8_t> cp;
  if (int rc = multi_check_search(m, pattern, pattern_len, k, out, cp)) return rc;
  SASSY_MULTI_NO_TICKETS(m);
  const uint32_t f = flags & (SASSY_HIP_ALL_MINIMA | SASSY_HIP_WITHOUT_TRACE);
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    p.result = nullptr;
    p.result_rc = nullptr;
    if (p.len == 0) p.result = new sassy_hip_Result();
    else if (int r = sassy_hip_search_shard(p.searcher, pattern, pattern_len, p.d_text, p.halo / 64 * 64, p.len, p.offset, m->total_len, k, f, &p.result))
      return r;
    if (m->rc) return multi_rc_shard(m, p, cp.data(), pattern_len, k, f);
    return 0;
  });
  return multi_merge(m, f, rc, out);
}

// ---- searches in flight over several devices (the reference's workers never idle between tasks: bin/grep.rs:516-537) ----
// begin() queues one shard search per device and strand (sassy_hip_search_shard_begin on the part's worker thread) and
// returns; finish() waits for them, in any order of tickets, and merges.  Up to depth (sassy_hip_multi_set_pipe_depth,
// 1 .. 4, default 3) searches per multi-searcher: the tail of search i -- chunk DP, traceback, the host's merge -- runs
// under the text stream of search i + 1 on every device.
struct sassy_hip_MultiTicket {
  sassy_hip_Multi* owner = nullptr;
  std::vector<uint8_t> pat, cpat;
  size_t k = 0;
  uint32_t f = 0;
  std::vector<sassy_hip_Ticket*> fwd, rcs;  // per part; nullptr: the part has no share
};

int sassy_hip_multi_set_pipe_depth(sassy_hip_Multi* m, int depth) {
  if (!m || depth < 1 || depth > kMaxLanes) return fail(SASSY_HIP_EINVAL, "pipe depth must be 1 .. 4");
  for (sassy_hip_Multi::Part& p : m->parts)
    if (p.open_tickets) return fail(SASSY_HIP_EINVAL, "searches are in flight");
  m->pipe_depth = depth;
  return 0;
}

int sassy_hip_multi_search_begin(sassy_hip_Multi* m, const uint8_t* pattern, size_t pattern_len, size_t k, uint32_t flags,
                                 sassy_hip_MultiTicket** out) {
  std::unique_ptr<sassy_hip_MultiTicket> T(new sassy_hip_MultiTicket());
  if (int rc = multi_check_search(m, pattern, pattern_len, k, out, T->cpat)) return rc;
  for (sassy_hip_Multi::Part& p : m->parts)
    if (p.open_tickets >= m->pipe_depth) return fail(SASSY_HIP_EINVAL, "too many searches in flight: finish one first (sassy_hip_multi_set_pipe_depth)");
  T->owner = m;
  T->pat.assign(pattern, pattern + pattern_len);
  T->k = k;
  T->f = flags & (SASSY_HIP_ALL_MINIMA | SASSY_HIP_WITHOUT_TRACE);
  T->fwd.assign(m->parts.size(), nullptr);
  T->rcs.assign(m->parts.size(), nullptr);
  sassy_hip_MultiTicket* t = T.get();
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t i = (size_t)(&p - m->parts.data());
    if (p.searcher->pipe_depth != m->pipe_depth)
      if (int r = sassy_hip_set_pipe_depth(p.searcher, m->pipe_depth)) return r;
    if (p.len != 0)
      if (int r = sassy_hip_search_shard_begin(p.searcher, t->pat.data(), pattern_len, p.d_text, p.halo / 64 * 64, p.len, p.offset,
                                               m->total_len, k, t->f, &t->fwd[i])) return r;
    if (m->rc) {
      uint64_t A = 0, B = 0, hrev = 0;
      const int pr = multi_rc_prepare(m, p, &A, &B, &hrev);
      if (pr == 1) return 0;
      if (pr) return pr;
      if (!p.searcher_rs) {
        p.searcher_rs = sassy_hip_searcher_new(m->alphabet.c_str(), false, m->alpha);
        if (!p.searcher_rs) return SASSY_HIP_EINVAL;
        p.searcher_rs->device = p.device;
      }
      if (p.searcher_rs->pipe_depth != m->pipe_depth)
        if (int r = sassy_hip_set_pipe_depth(p.searcher_rs, m->pipe_depth)) return r;
      if (int r = sassy_hip_search_shard_begin(p.searcher_rs, t->cpat.data(), pattern_len, p.d_rev, hrev, B - A, A, m->total_len, k,
                                               t->f, &t->rcs[i])) return r;
    }
    return 0;
  });
  if (rc) {  // what was begun on the other devices is waited for and dropped
    const std::string err = g_err;
    (void)m->on_all([&](sassy_hip_Multi::Part& p) -> int {
      const size_t i = (size_t)(&p - m->parts.data());
      if (t->fwd[i]) (void)sassy_hip_search_finish(p.searcher, t->fwd[i], nullptr);
      if (t->rcs[i]) (void)sassy_hip_search_finish(p.searcher_rs, t->rcs[i], nullptr);
      return 0;
    });
    g_err = err;
    return rc;
  }
  for (sassy_hip_Multi::Part& p : m->parts) ++p.open_tickets;
  *out = T.release();
  return 0;
}

int sassy_hip_multi_search_finish(sassy_hip_Multi* m, sassy_hip_MultiTicket* t, sassy_hip_Result** out) {
  if (!m || !t || t->owner != m || !out) return fail(SASSY_HIP_EINVAL, "not a ticket of this multi-searcher");
  std::unique_ptr<sassy_hip_MultiTicket> guard(t);
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t i = (size_t)(&p - m->parts.data());
    p.result = nullptr;
    p.result_rc = nullptr;
    int first = 0;
    if (t->fwd[i]) first = sassy_hip_search_finish(p.searcher, t->fwd[i], &p.result);
    else p.result = new sassy_hip_Result();
    if (m->rc) {
      int r2 = 0;
      if (t->rcs[i]) r2 = sassy_hip_search_finish(p.searcher_rs, t->rcs[i], &p.result_rc);
      else p.result_rc = new sassy_hip_Result();
      if (!first) first = r2;
    }
    return first;
  });
  for (sassy_hip_Multi::Part& p : m->parts)
    if (p.open_tickets) --p.open_tickets;
  if (rc)
    for (sassy_hip_Multi::Part& p : m->parts) {  // (a part that failed may have left no result at all)
      if (!p.result) p.result = new sassy_hip_Result();
      if (m->rc && !p.result_rc) p.result_rc = new sassy_hip_Result();
    }
  return multi_merge(m, t->f, rc, out);
}

// The layout arithmetic of a multi-searcher, without any device (tests; drivers that want to know a shard's bytes before
// they allocate): for a text of `len` bytes over `n_parts` devices with halos good for (max_pattern_len, max_k), part i's
// {offset, len, halo in front, bytes kept behind, first forward byte of its share of the REVERSED text, one past its
// last, that share's halo} go to out[7 i .. 7 i + 6]; returns the number of parts that hold a share, or -1 when some
// part's resident bytes would not cover its share of the reversed text (never, by construction).
long sassy_hip_multi_layout(uint64_t len, size_t n_parts, size_t max_pattern_len, size_t max_k, uint64_t* out) {
  if (n_parts == 0) return -1;
  const size_t E = sassy_hip_Multi::multi_eff_parts(len, n_parts);
  const uint64_t halo_for = sassy_hip_required_halo(max_pattern_len, max_k), slack = 64ull * (E + 1);
  long ok = (long)E;
  for (size_t i = 0; i < n_parts; ++i) {
    uint64_t a, b;
    sassy_hip_Multi::multi_bounds(len, E, i, a, b);
    uint64_t halo = (i == 0 || a == 0) ? 0 : std::min<uint64_t>(halo_for + slack, a);
    halo = halo / 64 * 64;
    const uint64_t halo_r = std::min<uint64_t>(halo_for + slack, len - b);
    uint64_t fa = 0, fb = 0, hrev = 0;
    if (i < E) {
      uint64_t A, B;
      sassy_hip_Multi::multi_bounds(len, E, E - 1 - i, A, B);
      fa = len - B; fb = len - A; hrev = std::min<uint64_t>(halo_for, A);
      if (fb > fa && (fa / 16 * 16 < a - halo || fb + hrev > b + halo_r)) ok = -1;
    }
    if (out) {
      uint64_t* o = out + 7 * i;
      o[0] = a; o[1] = b - a; o[2] = halo; o[3] = halo_r; o[4] = fa; o[5] = fb; o[6] = hrev;
    }
  }
  return ok;
}

// search_encoded_patterns over several devices: the PATTERNS are sharded (every device scans the whole text
// for its share of the patterns -- no halo, no seam), which needs the whole text on every device
// (sassy_hip_multi_set_replicated before the text is set).  pattern_idx of the result refers to the caller's list.
int sassy_hip_multi_search_encoded(sassy_hip_Multi* m, const uint8_t* patterns, size_t n_patterns, size_t pattern_len, size_t k,
                                   uint32_t flags, sassy_hip_Result** out) {
  if (!m || !patterns || !out) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_MULTI_NO_TICKETS(m);
  if (!m->have_text || !m->replicate)
    return fail(SASSY_HIP_EINVAL, "search_encoded over several devices shards the patterns: every device needs the whole text "
                                  "(sassy_hip_multi_set_replicated(m, 1), then set the text)");
  if (n_patterns == 0) return fail(SASSY_HIP_EINVAL, "No queries provided");
  const size_t G = m->parts.size();
  const uint32_t f = (flags & (SASSY_HIP_ALL_MINIMA | SASSY_HIP_WITHOUT_TRACE)) | SASSY_HIP_TEXT_ON_DEVICE;
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t i = (size_t)(&p - m->parts.data());
    const size_t p0 = n_patterns * i / G, p1 = n_patterns * (i + 1) / G;
    p.result = nullptr;
    if (p1 == p0) { p.result = new sassy_hip_Result(); return 0; }
    sassy_SearcherType* s = p.searcher;
    if (m->rc) {
      if (!p.searcher_rc) {
        p.searcher_rc = sassy_hip_searcher_new(m->alphabet.c_str(), true, m->alpha);
        if (!p.searcher_rc) return SASSY_HIP_EINVAL;
        p.searcher_rc->device = p.device;
      }
      s = p.searcher_rc;
    }
    sassy_hip_Encoded* e = sassy_hip_encode_patterns(s, patterns + p0 * pattern_len, p1 - p0, pattern_len);
    if (!e) return SASSY_HIP_EINVAL;
    const int r = sassy_hip_search_encoded(s, e, p.d_text, (size_t)m->total_len, k, f, &p.result);
    sassy_hip_encoded_free(e);
    return r;
  });
  std::unique_ptr<sassy_hip_Result> R(new sassy_hip_Result());
  int mrc = rc;
  for (size_t i = 0; i < G && !mrc; ++i) {
    const sassy_hip_Result* r = m->parts[i].result;
    const size_t p0 = n_patterns * i / G, base = R->pool.size();
    if (base + r->pool_size() > 0xFFFFFFFFull) { mrc = fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB"); break; }
    R->pool.append(r->pool_data(), r->pool_size());
    const sassy_hip_Match* rm = r->data();
    for (size_t x = 0; x < r->size(); ++x) {
      sassy_hip_Match q = rm[x];
      q.pattern_idx += p0;
      q.cigar_off = (uint32_t)(q.cigar_off + base);
      R->matches.push_back(q);
    }
  }
  for (sassy_hip_Multi::Part& p : m->parts) { delete p.result; p.result = nullptr; }
  if (mrc) return mrc;
  if (R->pool.empty()) R->pool.push_back('\0');
  *out = R.release();
  return 0;
}

// search_many over several devices: the TEXTS are sharded (whole texts; contiguous runs of about equal total length),
// every device searches all patterns in its texts; text_idx of the result refers to the caller's list.  Host texts;
// nothing resident is needed.  Order: device by device, each in sassy_hip_search_many's order.
int sassy_hip_multi_search_many(sassy_hip_Multi* m, const uint8_t* const* patterns, const size_t* pattern_lens, size_t n_patterns,
                                const uint8_t* const* texts, const size_t* text_lens, size_t n_texts, size_t k, uint32_t flags,
                                sassy_hip_Result** out) {
  if (!m || !out || (!patterns && n_patterns) || (!texts && n_texts)) return fail(SASSY_HIP_EINVAL, "null argument");
  SASSY_MULTI_NO_TICKETS(m);
  const size_t G = m->parts.size();
  const uint32_t f = flags & (SASSY_HIP_ALL_MINIMA | SASSY_HIP_WITHOUT_TRACE);
  // cut points: text t goes to part floor(G * (bytes in front of t) / total)
  std::vector<size_t> first(G + 1, n_texts);
  {
    uint64_t total = 0;
    for (size_t t = 0; t < n_texts; ++t) total += text_lens[t] + 64;
    uint64_t before = 0;
    size_t g = 0;
    first[0] = 0;
    for (size_t t = 0; t < n_texts; ++t) {
      const size_t want = total ? (size_t)((unsigned __int128)before * G / total) : 0;
      while (g < want && g + 1 < G) first[++g] = t;
      before += text_lens[t] + 64;
    }
    while (g + 1 <= G - 1) first[++g] = n_texts;
    first[G] = n_texts;
  }
  const int rc = m->on_all([&](sassy_hip_Multi::Part& p) -> int {
    const size_t i = (size_t)(&p - m->parts.data());
    const size_t t0 = first[i], t1 = first[i + 1];
    p.result = nullptr;
    if (t1 <= t0 || n_patterns == 0) { p.result = new sassy_hip_Result(); return 0; }
    sassy_SearcherType* s = p.searcher;
    if (m->rc) {
      if (!p.searcher_rc) {
        p.searcher_rc = sassy_hip_searcher_new(m->alphabet.c_str(), true, m->alpha);
        if (!p.searcher_rc) return SASSY_HIP_EINVAL;
        p.searcher_rc->device = p.device;
      }
      s = p.searcher_rc;
    }
    return sassy_hip_search_many(s, patterns, pattern_lens, n_patterns, texts + t0, text_lens + t0, t1 - t0, k, f, &p.result);
  });
  std::unique_ptr<sassy_hip_Result> R(new sassy_hip_Result());
  int mrc = rc;
  for (size_t i = 0; i < G && !mrc; ++i) {
    const sassy_hip_Result* r = m->parts[i].result;
    const size_t base = R->pool.size();
    if (base + r->pool_size() > 0xFFFFFFFFull) { mrc = fail(SASSY_HIP_EUNSUPPORTED, "cigar pool of one result exceeds 4 GiB"); break; }
    R->pool.append(r->pool_data(), r->pool_size());
    const sassy_hip_Match* rm = r->data();
    for (size_t x = 0; x < r->size(); ++x) {
      sassy_hip_Match q = rm[x];
      q.text_idx += first[i];
      q.cigar_off = (uint32_t)(q.cigar_off + base);
      R->matches.push_back(q);
    }
  }
  for (sassy_hip_Multi::Part& p : m->parts) { delete p.result; p.result = nullptr; }
  if (mrc) return mrc;
  if (R->pool.empty()) R->pool.push_back('\0');
  *out = R.release();
  return 0;
}

void sassy_hip_multi_free(sassy_hip_Multi* m) { delete m; }

}  // extern "C"
