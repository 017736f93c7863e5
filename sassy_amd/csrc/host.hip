// host.hip -- host side of libsassy_hip.so: the Searcher (mirror of the reference's
// Searcher<P>, reference: src/search.rs:227-937) and the C-ABI of include/sassy.h + sassy_hip.h.
//
// Division of labour: the HIP kernels find every reported (end position, cost); the host sorts
// the (rare) records, resolves the plateau-entry chain across lane chunks, and turns each record
// into a Match with the reference's traceback rules.  There is no CPU scan path: if the device
// is unusable every search entry point fails (additive API) or aborts (drop-in API).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <iterator>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sassy_hip.h"
#include "common.h"
#include "profiles.h"

namespace sassy_hip {
thread_local LaunchEvents g_launch_events;


// kernel launchers (scan_kernel.hip is compiled once per profile; aux_kernels.hip)
hipError_t launch_scan_dna(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_scan_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_scan_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_filter_dna(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_filter_table(const ScanParams& P, uint32_t grid, hipStream_t stream);
hipError_t launch_filter_dna_multi(const ScanParams& P, uint32_t grid, hipStream_t stream);
hipError_t launch_filter_count(const ScanParams& P, uint32_t grid, hipStream_t stream);
hipError_t launch_count_n(const uint8_t* d_text, const uint64_t* d_range, uint32_t n, uint32_t* d_count,
                          hipStream_t stream);
hipError_t launch_acgt_check(const uint8_t* d_text, uint64_t n, uint32_t* d_flag, hipStream_t stream, int allow_x = 0);
hipError_t launch_reverse_texts(const uint8_t* d_src, uint8_t* d_dst, uint64_t n, const uint32_t* d_blk2text,
                                const uint64_t* d_start, const uint64_t* d_len, uint32_t pad, hipStream_t stream);
hipError_t launch_filter_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_filter_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_list_dna(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_list_iupac(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_list_ascii(const ScanParams& P, uint32_t grid, size_t smem, hipStream_t stream);
hipError_t launch_build_chunks(const unsigned long long* d_hit, uint64_t n_words, uint64_t n_blocks,
                               uint64_t first_owned, uint32_t wb, uint32_t L, uint32_t maxlen,
                               ChunkDesc* d_desc, uint32_t* d_desc_count, uint32_t desc_cap,
                               unsigned long long* d_hit_count, hipStream_t stream);
hipError_t launch_generate_dna(uint8_t* d_text, uint64_t n, uint64_t seed, uint64_t first, hipStream_t stream);
hipError_t launch_generate_genome_like(uint8_t* d_text, uint64_t n, uint64_t seed, uint64_t first, int with_n,
                                       hipStream_t stream);
hipError_t launch_scatter_bytes(uint8_t* d_text, uint64_t n, uint64_t first, const uint64_t* d_pos,
                                const uint8_t* d_val, uint64_t count, hipStream_t stream);
hipError_t launch_reverse(const uint8_t* d_in, uint8_t* d_out, uint64_t n, hipStream_t stream);
hipError_t launch_trace(const TraceParams& P, uint32_t nblocks, hipStream_t stream);
hipError_t launch_rank(const Candidate* d_cand, const uint32_t* d_count, uint32_t cap, uint32_t* d_rank,
                       Candidate* d_sorted, Candidate* h_sorted, uint32_t host_cap, void* h_ctl,
                       const TextTable& texts, hipStream_t stream);

size_t sort_scratch_bytes(uint32_t count);
hipError_t launch_sort_candidates(const Candidate* d_cand, Candidate* d_sorted, uint32_t count, void* d_scratch,
                                  size_t scratch_bytes, hipStream_t stream, int by_tag = 0, int key_bits = 64);
hipError_t launch_report_flags(const Candidate* d_list, uint32_t max_count, const uint32_t* d_count, uint64_t min_pos,
                               uint32_t* d_flags, hipStream_t stream);
size_t unique_scratch_bytes(uint32_t count);
hipError_t launch_unique_reports(Candidate* d_sorted, uint32_t count, uint64_t min_pos, Candidate* d_out, uint32_t* d_out_count,
                                 void* d_scratch, size_t scratch_bytes, hipStream_t stream);
size_t select_scratch_bytes(uint32_t count);
hipError_t launch_select_reports(const Candidate* d_sorted, uint32_t count, Candidate* d_sel, uint32_t* d_sel_count,
                                 void* d_scratch, size_t scratch_bytes, hipStream_t stream, int all_minima = 0);
hipError_t launch_seed_search(const SeedParams& P, uint32_t grid, hipStream_t stream);
hipError_t launch_pack_text(const uint8_t* d_text, uint64_t n, uint32_t* d_packed, hipStream_t stream);
hipError_t launch_dirty_scan(const uint8_t* d_text, uint64_t n, unsigned long long* d_starts, unsigned long long* d_ends,
                             unsigned long long* d_hard, uint32_t cap, uint32_t* d_counts, hipStream_t stream);
hipError_t launch_gather_zones(const uint8_t* d_text, uint8_t* d_dst, const unsigned long long* d_seg, uint32_t n_seg,
                               hipStream_t stream);
hipError_t launch_map_zone_list(const Candidate* d_in, uint32_t count, const unsigned long long* d_zone, uint32_t n_zones,
                                Candidate* d_out, uint32_t* d_out_count, uint32_t out_cap, hipStream_t stream);
hipError_t launch_drop_excluded(const Candidate* d_in, uint32_t count, const unsigned long long* d_excl, uint32_t n_excl,
                                unsigned char* d_keep, hipStream_t stream);
hipError_t launch_compact_candidates(const Candidate* d_in, uint32_t count, const unsigned char* d_keep, Candidate* d_out,
                                     uint32_t* d_out_count, void* d_scratch, size_t scratch_bytes, hipStream_t stream);
hipError_t launch_tiled_scan(const TiledParams& P, hipStream_t stream);
hipError_t launch_tiled_pertext(const TiledParams& P, hipStream_t stream);
size_t many_scratch_bytes(uint32_t count);
hipError_t launch_assemble_many(const ManyPart& a, const ManyPart& b, uint32_t n_texts, const uint64_t* d_text_len,
                                uint64_t first_text, uint32_t str_stride, MatchOut* d_rows, char* d_strs, uint32_t* d_flags,
                                void* d_scratch, size_t scratch_bytes, hipStream_t stream, int flip = 1);
hipError_t launch_assign_texts(Candidate* d_rep, uint32_t count, const TextTable& texts, uint32_t* d_report_text,
                               hipStream_t stream);
hipError_t launch_keep_interior(const Candidate* d_rec, uint32_t count, const TextTable& texts, uint32_t edge, unsigned char* d_keep,
                                hipStream_t stream);
size_t compact_scratch_bytes(uint32_t max_count, uint32_t str_stride);
hipError_t launch_compact_cigars(MatchOut* d_rows, const char* d_strs, uint32_t max_count, const uint32_t* d_count, uint32_t str_stride,
                                 uint32_t* d_total, void* d_scratch, size_t scratch_bytes, const char** d_out_strs, hipStream_t stream);
size_t encoded_scratch_bytes(uint32_t count);
hipError_t launch_assemble_encoded(const MatchOut* d_rows_in, const char* d_strs_in, uint32_t count, uint64_t n_original, uint32_t str_stride,
                                   int key_bits, MatchOut* d_rows, char* d_strs, uint32_t* d_flags, void* d_scratch, size_t scratch_bytes,
                                   hipStream_t stream);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int hip_fail(hipError_t e, const char* what) {
  return fail(SASSY_HIP_ENODEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

// A growable device buffer.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  int reserve(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace sassy_hip

using namespace sassy_hip;

// Pinned, device-mapped host blocks: the kernels write a search's control block, reports, match rows and cigar
// strings straight into one (ScanLane::h_pin).  A result that needs no host-side editing ADOPTS the block instead of
// copying 0.5 MB out of it (17 us of a 0.64 ms search); the lane takes another block from this pool, and
// sassy_hip_result_free puts the adopted one back.
struct PinBlock {
  unsigned char* h = nullptr;
  unsigned char* d = nullptr;  // device address of h
  size_t cap = 0;
  int dev = -1;
};
struct PinPool {
  std::mutex mu;
  std::vector<PinBlock> blocks;
  static constexpr size_t kKeep = 12;
  static constexpr size_t kKeepBytes = (size_t)5 << 30;  // idle blocks kept for reuse (dense results hold 100 MB and more each; a
                                                         // guide set's 17 M matches 2.2 GB -- pinning that again costs more than the search)
  // blocks that results hold right now: a caller who keeps every result alive must not pin memory without bound (and
  // pay a hipHostMalloc per search) -- beyond kMaxAdopted outstanding blocks (or kMaxAdoptedBytes) results are copied
  // out as before
  static constexpr int kMaxAdopted = 16;
  static constexpr size_t kMaxAdoptedBytes = (size_t)8 << 30;
  int adopted = 0;
  size_t adopted_bytes = 0;
  bool may_adopt(size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    if (adopted >= kMaxAdopted || adopted_bytes + bytes > kMaxAdoptedBytes) return false;
    ++adopted;
    adopted_bytes += bytes;
    return true;
  }
  void adopted_back(size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    if (adopted > 0) --adopted;
    adopted_bytes -= std::min(adopted_bytes, bytes);
  }
  bool take(size_t bytes, int dev, PinBlock& out) {
    std::lock_guard<std::mutex> g(mu);
    for (size_t i = 0; i < blocks.size(); ++i)
      if (blocks[i].dev == dev && blocks[i].cap >= bytes && blocks[i].cap <= 2 * bytes + (1u << 20)) {
        out = blocks[i];
        blocks.erase(blocks.begin() + (long)i);
        return true;
      }
    return false;
  }
  // newest block in, oldest blocks out: after a few dense searches the pool holds their 100 MB blocks, and a
  // following stream of small results must still find its own block sizes kept (dropping the NEW block instead
  // cost every such search a hipHostFree + hipHostMalloc, about 0.9 ms)
  void give(const PinBlock& b) {
    if (!b.h) return;
    std::vector<PinBlock> drop;
    {
      std::lock_guard<std::mutex> g(mu);
      if (b.cap > kKeepBytes) drop.push_back(b);
      else {
        blocks.push_back(b);
        size_t kept = 0;
        for (const PinBlock& x : blocks) kept += x.cap;
        while (blocks.size() > kKeep || kept > kKeepBytes) {
          kept -= blocks.front().cap;
          drop.push_back(blocks.front());
          blocks.erase(blocks.begin());
        }
      }
    }
    for (const PinBlock& x : drop) (void)hipHostFree(x.h);
  }
};
static PinPool g_pin_pool;

struct sassy_hip_Result {
  std::vector<sassy_hip_Match> matches;
  std::string pool;
  int exit_state = kStateDecTrue;
  int64_t conditional_index = -1;
  // adopted pinned block (pin.h != nullptr): the rows and the cigar pool live in it, the vectors above are empty
  PinBlock pin;
  const sassy_hip_Match* ext_matches = nullptr;
  size_t ext_n = 0;
  const char* ext_pool = nullptr;
  size_t ext_pool_len = 0;
  size_t size() const { return pin.h ? ext_n : matches.size(); }
  const sassy_hip_Match* data() const { return pin.h ? ext_matches : matches.data(); }
  const char* pool_data() const { return pin.h ? ext_pool : pool.c_str(); }
  size_t pool_size() const { return pin.h ? ext_pool_len : pool.size(); }
  ~sassy_hip_Result() {
    if (pin.h) g_pin_pool.adopted_back(pin.cap);
    g_pin_pool.give(pin);
  }
};

// One search in flight: everything its ScanJob refers to lives here until sassy_hip_search_finish.
struct sassy_hip_Ticket {
  sassy_SearcherType* owner = nullptr;
  int lane = -1;
  sassy_hip::PatternPlan plan;
  std::vector<uint8_t> pat;
  uint64_t total_len = 0;
  bool without_trace = false;
  bool empty_shard = false;
  double t0 = 0;
  std::shared_ptr<void> job;     // the ScanJob (defined below)
};

struct sassy_hip_Encoded {
  Profile profile;
  bool rc;
  size_t plen;
  std::vector<std::vector<uint8_t>> patterns;  // originals, then (if rc) their reverse complements
  size_t n_original;
};

// Everything one scan pipeline (filter -> chunk list -> DP -> rank -> traceback) needs for itself:
// a stream, its timing events, its device work buffers and the pinned, device-mapped host buffer
// its kernels write the results into.  A searcher owns several lanes so that a long text can be
// cut into sub-shards whose pipelines overlap (the next sub-shard's bandwidth-bound filter runs
// while the previous one's latency-bound DP / rank / traceback kernels finish).
constexpr int kMaxLanes = 4;
struct ScanLane {
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_f = nullptr, ev_filter_done = nullptr;
  DevBuf<uint8_t> d_state, d_scratch, d_str, d_ctl, d_sort, d_scratch2;
  DevBuf<uint32_t> d_flags;     // dense results: "does any record need the host's attention" (report_flags_kernel)
  DevBuf<Candidate> d_cand, d_sorted;
  DevBuf<MatchOut> d_trace;
  DevBuf<ChunkDesc> d_desc;
  // pattern-dependent device data of the scan that runs on this lane, and what it currently holds
  // (uploads are skipped when the pattern repeats); per lane, so that scans of different patterns
  // can be in flight on different lanes
  DevBuf<uint8_t> d_pattern, d_table;
  DevBuf<TextStash> d_stash;   // fused filter: the text under the reports, for the traceback
  DevBuf<unsigned long long> d_probe;  // SASSY_HIP_TRACE_PROBE
  DevBuf<uint32_t> d_rowoff, d_ovtab;
  std::vector<uint8_t> up_pattern, h_table, table_pattern;
  std::vector<uint32_t> up_rowtab, up_ovtab;
  int up_profile = -1, table_profile = -1;
  uint32_t table_q = 0, table_k = 0, table_r = 0;   // table_r: 0 = piece bit table, else the counting table's R
  bool table_rc = false;                            // the counting table also holds the Rc strand's q-grams
  uint32_t fuse_backoff = 0;                        // searches this lane still runs unfused after a fused one overflowed
  double table_density = 0;
  // pinned host staging area: control block and the first kSpec reports of a scan are written into
  // it by the kernels themselves; one stream synchronisation makes them readable
  unsigned char* h_pin = nullptr;
  unsigned char* h_pin_dev = nullptr;  // device address of h_pin
  size_t h_pin_cap = 0;
  bool ready = false;
  // small host -> device uploads (pattern, row table, tables) go through pinned memory: a copy from
  // pageable memory makes the host wait for the device, which serialises the lanes of a ScanQueue
  uint8_t* h_up = nullptr;
  size_t h_up_cap = 0, h_up_used = 0;
  int upload(void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return 0;
    const size_t need = h_up_used + ((bytes + 63) & ~(size_t)63);
    if (need > h_up_cap) {
      if (h_up_used != 0) {  // no room left behind the copies already queued: an ordinary copy
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
        return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpyAsync");
      }
      if (h_up) (void)hipHostFree(h_up);
      h_up = nullptr;
      h_up_cap = 0;
      const size_t want = std::max<size_t>(need * 2, 256 * 1024);
      hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h_up), want, hipHostMallocDefault);
      if (e != hipSuccess) {  // no pinned memory to be had: an ordinary (host-blocking) copy
        (void)hipGetLastError();
        h_up = nullptr;
        e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
        return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpyAsync");
      }
      h_up_cap = want;
    }
    memcpy(h_up + h_up_used, src, bytes);
    hipError_t e = hipMemcpyAsync(dst, h_up + h_up_used, bytes, hipMemcpyHostToDevice, stream);
    h_up_used = need;
    return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpyAsync");
  }
  // Results too large for the device-mapped staging area (dense matches: 10^5 .. 10^6 records) come back with
  // ordinary copies.  A copy into pageable memory runs at ~5 GB/s; through two pinned 8 MiB buffers, the next
  // piece in flight while the previous one is moved to its final place, the transfer runs at the speed of the
  // host memcpy.
  static constexpr size_t kBulk = 8u << 20;
  unsigned char* h_bulk[2] = {nullptr, nullptr};
  int download(void* dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return 0;
    for (unsigned char*& b : h_bulk)
      if (!b && hipHostMalloc(reinterpret_cast<void**>(&b), kBulk, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        b = nullptr;
      }
    if (!h_bulk[0] || !h_bulk[1] || bytes < (1u << 20)) {  // small, or no pinned memory to be had
      hipError_t e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpy (results)");
    }
    unsigned char* out = static_cast<unsigned char*>(dst);
    const unsigned char* src = static_cast<const unsigned char*>(d_src);
    size_t issued = 0, done = 0;
    int slot = 0;
    size_t len[2] = {0, 0};
    // prime one piece, then: wait for piece i, issue piece i+1, move piece i
    len[0] = std::min(kBulk, bytes);
    hipError_t e = hipMemcpyAsync(h_bulk[0], src, len[0], hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync (results)");
    issued = len[0];
    while (done < bytes) {
      e = hipStreamSynchronize(stream);
      if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize (results)");
      const int cur = slot;
      slot ^= 1;
      if (issued < bytes) {
        len[slot] = std::min(kBulk, bytes - issued);
        e = hipMemcpyAsync(h_bulk[slot], src + issued, len[slot], hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync (results)");
        issued += len[slot];
      }
      memcpy(out + done, h_bulk[cur], len[cur]);
      done += len[cur];
    }
    return 0;
  }
  int h_pin_device = -1;
  int reserve_pinned(size_t bytes) {
    if (bytes <= h_pin_cap) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (h_pin) g_pin_pool.give(PinBlock{h_pin, h_pin_dev, h_pin_cap, h_pin_device});
    h_pin = nullptr;
    h_pin_cap = 0;
    PinBlock b;
    if (g_pin_pool.take(bytes, dev, b)) {
      h_pin = b.h; h_pin_dev = b.d; h_pin_cap = b.cap; h_pin_device = b.dev;
      return 0;
    }
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h_pin), bytes, hipHostMallocMapped);
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h_pin_dev), h_pin, 0);
    if (e != hipSuccess) return hip_fail(e, "hipHostGetDevicePointer");
    h_pin_cap = bytes;
    h_pin_device = dev;
    return 0;
  }
  // hands the block to a result; the next search reserves another one (from the pool)
  PinBlock take_pin() {
    PinBlock b{h_pin, h_pin_dev, h_pin_cap, h_pin_device};
    h_pin = h_pin_dev = nullptr;
    h_pin_cap = 0;
    return b;
  }
  int init() {
    if (ready) return 0;
    if (!stream) {
      HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      own_stream = true;
    }
    HIP_TRY(hipEventCreate(&ev_a));
    HIP_TRY(hipEventCreate(&ev_b));
    HIP_TRY(hipEventCreate(&ev_c));
    HIP_TRY(hipEventCreate(&ev_f));
    HIP_TRY(hipEventCreateWithFlags(&ev_filter_done, hipEventDisableTiming));
    ready = true;
    return 0;
  }
  void destroy() {
    if (h_up) (void)hipHostFree(h_up);
    h_up = nullptr; h_up_cap = h_up_used = 0;
    d_state.release(); d_scratch.release(); d_str.release(); d_ctl.release(); d_sort.release(); d_scratch2.release(); d_flags.release();
    for (unsigned char*& b : h_bulk) { if (b) (void)hipHostFree(b); b = nullptr; }
    d_cand.release(); d_sorted.release(); d_trace.release(); d_desc.release();
    d_pattern.release(); d_table.release(); d_rowoff.release(); d_ovtab.release(); d_stash.release();
    if (h_pin) g_pin_pool.give(PinBlock{h_pin, h_pin_dev, h_pin_cap, h_pin_device});
    h_pin = nullptr;
    for (hipEvent_t e : {ev_a, ev_b, ev_c, ev_f, ev_filter_done})
      if (e) (void)hipEventDestroy(e);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

// The searcher.  Mirrors the configuration surface of the reference's Searcher<P>
// (rc, alpha; reference: src/search.rs:227-256, 486-503) and caches device buffers the way the
// reference caches its host buffers.
// On-line choice of the lane-chunk length of the prefilter for a resident text.  The kernel time depends
// chaotically on it (HBM channel mapping against the lane stride and the lane count: +-10 % between
// neighbouring even values, see stream_geometry), so the first searches of a (text, filter kind) try
// the even values around the default, two calls each, and the rest use the fastest.  Every trial is a
// complete, correct search; only its prefilter geometry differs.
struct GeoTuner {
  const void* text = nullptr;
  uint64_t len = 0, owned = 0;
  uint32_t kind = 0, extra = 0;
  std::vector<uint32_t> cand;
  std::vector<float> best;
  uint32_t trials = 0, chosen = 0;
  // the chunk length to use for this call (0: the default)
  uint32_t next(const void* t, uint64_t l, uint64_t own, uint32_t k, uint32_t ex, uint32_t dflt, uint32_t min_bpl) {
    if (t != text || l != len || own != owned || k != kind || ex != extra) {
      text = t; len = l; owned = own; kind = k; extra = ex;
      cand.clear();
      for (int d = 0; d <= 14; d += 2) {
        for (int sgn = (d ? -1 : 1); sgn <= 1; sgn += 2) {
          const int64_t v = (int64_t)dflt + sgn * d;
          if (v >= (int64_t)min_bpl && v >= 4) cand.push_back((uint32_t)v);
        }
      }
      // ... and around two thirds of it (half again as many lanes: the other basin seen in the sweeps)
      for (int d = -2; d <= 2; d += 2) {
        const int64_t v = ((int64_t)dflt * 2 / 3) / 2 * 2 + d;
        if (v >= (int64_t)min_bpl && v >= 4 && std::find(cand.begin(), cand.end(), (uint32_t)v) == cand.end())
          cand.push_back((uint32_t)v);
      }
      best.assign(cand.size(), 1e30f);
      trials = 0;
      chosen = 0;
    }
    if (chosen) return chosen;
    if (trials < 2 * cand.size()) return cand[trials % cand.size()];
    size_t b = 0;
    for (size_t i = 1; i < cand.size(); ++i)
      if (best[i] < best[b]) b = i;
    chosen = cand[b];
    return chosen;
  }
  void report(uint32_t bpl, float ms) {
    if (chosen) return;
    for (size_t i = 0; i < cand.size(); ++i)
      if (cand[i] == bpl) { best[i] = std::min(best[i], ms); break; }
    ++trials;
  }
};

struct sassy_SearcherType {
  Profile profile = PROFILE_DNA;
  bool rc = false;
  // lanes[0].stream doubles as the searcher's stream: text / pattern uploads and everything that
  // is not split into sub-shards run on it (sassy_hip_set_stream replaces it)
  ScanLane lanes[kMaxLanes];
  hipStream_t stream = nullptr;
  hipStream_t user_stream = nullptr;
  hipEvent_t ev_inputs = nullptr;  // "uploads of this call are queued" (other lanes wait for it)
  bool device_ready = false;
  // The HIP device all of this searcher's streams, buffers and launches live on: the calling thread's current device at
  // the searcher's first search (HIP's current device is per host thread), or sassy_hip_set_device before it.  Every
  // entry point switches to it for the duration of the call (DeviceGuard), whatever thread it is called from.
  int device = -1;
  bool bound = false;  // an entry point has run on `device` (streams / buffers / events may exist there): it stays
  DevBuf<uint8_t> d_text, d_rev;
  DevBuf<unsigned long long> d_rc_bitmap;  // the Rc strand's candidate blocks, marked by the forward pass
  // search_many lays its texts out in pinned host memory (no zero fill, H2D at the PCIe rate, reused
  // across calls)
  uint8_t* h_stage = nullptr;
  size_t h_stage_cap = 0;
  bool h_stage_pinned = false;
  void free_stage() {
    if (h_stage) {
      if (h_stage_pinned) (void)hipHostFree(h_stage);
      else free(h_stage);
    }
    h_stage = nullptr;
    h_stage_cap = 0;
  }
  int reserve_stage(size_t bytes) {
    if (bytes <= h_stage_cap) return 0;
    free_stage();
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h_stage), want, hipHostMallocDefault);
    h_stage_pinned = e == hipSuccess;
    if (!h_stage_pinned) {  // no pinned memory to be had (locked-memory limit): ordinary memory, slower upload
      (void)hipGetLastError();
      h_stage = static_cast<uint8_t*>(malloc(want));
      if (!h_stage) return fail(SASSY_HIP_ENOMEM, "out of host memory (text staging)");
    }
    h_stage_cap = want;
    return 0;
  }
  const uint8_t* rev_src = nullptr;  // d_rev holds reverse(rev_src[0 .. rev_len)) (SASSY_HIP_TEXT_UNCHANGED)
  uint64_t rev_len = 0;

  bool want_counters = false;
  int prefilter = -1;            // sassy_hip_set_prefilter: -1 process default, 0 never, 1 also with short pieces
  // sassy_hip_set_fused / SASSY_HIP_FUSED: the bit-plane filter runs the chunk DP of what it finds itself (one launch
  // instead of filter -> chunk list -> list kernel); 0 = always the classic chain
  bool fuse = !(getenv("SASSY_HIP_FUSED") && atoi(getenv("SASSY_HIP_FUSED")) == 0);
  // sassy_hip_set_reference_lanes: 0 = the definition (one pass), 4 / 8 = the reference binary's lane reports
  // (anything but 4 or 8 in the environment is ignored, as the setter rejects it)
  uint32_t ref_lanes = (getenv("SASSY_HIP_REF_LANES") && (atoi(getenv("SASSY_HIP_REF_LANES")) == 4 || atoi(getenv("SASSY_HIP_REF_LANES")) == 8))
                           ? (uint32_t)atoi(getenv("SASSY_HIP_REF_LANES")) : 0u;
  // searches in flight (sassy_hip_search_shard_begin / sassy_hip_search_finish): the ticket that owns each lane
  struct sassy_hip_Ticket* lane_ticket[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  int last_begun_lane = -1;
  int pipe_depth = getenv("SASSY_HIP_PIPE_DEPTH") ? std::max(1, std::min(atoi(getenv("SASSY_HIP_PIPE_DEPTH")), kMaxLanes)) : 2;
  // reporting modes of the reference's Searcher (src/search.rs:442-475)
  float alpha = NAN;             // overhang cost per pattern character (NaN = no overhang), Iupac only
  long max_overhang = -1;        // with_max_overhang(): -1 = none
  bool only_best = false;        // only_best_match(): one match per strand, minimal cost, rightmost end
  float max_n_frac = NAN;        // with_max_n_frac(): NaN = off (the reference's None)
  GeoTuner tuner, tuner_scan;    // prefilter / streaming-DP geometry per resident text
  // the on-line geometry tuner is opt-in (sassy_hip_set_geometry_tuner, SASSY_HIP_TUNE=1): see stream_geometry
  bool tune = getenv("SASSY_HIP_TUNE") != nullptr && atoi(getenv("SASSY_HIP_TUNE")) != 0;
  DevBuf<uint64_t> d_tables;     // multi-text buffers: start / len tables (both strands)
  DevBuf<unsigned long long> d_multi_bitmap;  // multi-pattern prefilter: one hit bitmap per pattern of the batch
  DevBuf<uint32_t> d_multi_bits;
  // pattern-tiled search (search_encoded_tiled): match masks, the patterns' bytes, counters, the selected reports
  DevBuf<unsigned long long> d_tiled_peq;
  DevBuf<uint8_t> d_tiled_pat;
  DevBuf<uint32_t> d_tiled_cnt, d_tiled_rtext;
  DevBuf<Candidate> d_tiled_sel, d_tiled_list;  // (the list is not a lane's d_cand: its size must not leak into single searches)
  // seeded search (search_encoded_seeded): the piece tables; sub-piece table, packed text and patterns
  DevBuf<uint32_t> d_seed_start[2], d_seed_entries[2], d_seed_sub, d_seed_packed, d_seed_bits, d_seed_e16;
  // ... on texts with other letters (seeded_dirty_zones): run lists / tables, the gathered neighbourhoods, their scan
  DevBuf<unsigned long long> d_zone_u64, d_zone_tab, d_zone_peq;
  DevBuf<uint8_t> d_zone_text;
  DevBuf<Candidate> d_zone_list;
  hipEvent_t ev_multi = nullptr, ev_multi_a = nullptr;
  hipEvent_t ev_a_multi() { return ev_multi_a; }
  DevBuf<uint64_t> d_range;      // N counting on device-resident text
  DevBuf<uint32_t> d_ncount;
  // HIP-event timing of the call's phases: 0 none, 1 the dominant kernel only (filter / streaming
  // scan; default), 2 every phase.  Each event record costs a few microseconds of stream idle time.
  int timing = getenv("SASSY_HIP_TIMING") ? atoi(getenv("SASSY_HIP_TIMING")) : 1;
  sassy_hip_Stats stats{};
  // the drop-in search() over several devices (SASSY_HIP_DEVICES): a multi-device searcher of this searcher's alphabet
  // and strands, made at the first such call (sassy_hip_Multi is defined further down: owned through its deleter)
  std::shared_ptr<void> multi;

  ~sassy_SearcherType() {
    for (ScanLane& l : lanes)  // searches still in flight (tickets never finished): let their kernels drain
      if (l.stream) (void)hipStreamSynchronize(l.stream);
    for (sassy_hip_Ticket*& t : lane_ticket) { delete t; t = nullptr; }
    d_text.release(); d_rev.release(); d_rc_bitmap.release();
    free_stage();
    d_range.release(); d_ncount.release(); d_tables.release(); d_multi_bitmap.release(); d_multi_bits.release();
    d_tiled_peq.release(); d_tiled_pat.release(); d_tiled_cnt.release(); d_tiled_sel.release(); d_tiled_list.release(); d_tiled_rtext.release();
    for (int t = 0; t < 2; ++t) { d_seed_start[t].release(); d_seed_entries[t].release(); }
    d_seed_sub.release(); d_seed_packed.release(); d_seed_bits.release(); d_seed_e16.release();
    d_zone_u64.release(); d_zone_tab.release(); d_zone_peq.release(); d_zone_text.release(); d_zone_list.release();
    if (ev_multi) (void)hipEventDestroy(ev_multi);
    if (ev_multi_a) (void)hipEventDestroy(ev_multi_a);
    for (ScanLane& l : lanes) l.destroy();
    if (ev_inputs) (void)hipEventDestroy(ev_inputs);
  }

  int ensure_device() {
    if (device_ready) return 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
      return fail(SASSY_HIP_ENODEVICE,
                  "no usable HIP device (libsassy_hip has no CPU fallback; the scan runs on gfx950 only)");
    lanes[0].stream = user_stream;  // null: the lane creates its own
    for (ScanLane& l : lanes) {
      if (getenv("SASSY_HIP_PIPE_ONESTREAM") && &l != &lanes[0]) l.stream = lanes[0].stream;
      if (int rc = l.init()) return rc;
    }
    stream = lanes[0].stream;
    HIP_TRY(hipEventCreateWithFlags(&ev_inputs, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&ev_multi));
    HIP_TRY(hipEventCreate(&ev_multi_a));
    device_ready = true;
    return 0;
  }
};

namespace sassy_hip {

// pa_types::Cigar::to_string: run-length encoded "<count><op>" (SURVEY 8c).  `ops` holds one
// char per alignment column in end -> start order (as the device traceback writes them).
static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// SASSY_HIP_DEBUG_TIMING=1: print host-side phase marks of every call to stderr
struct PhaseMarks {
  bool on = getenv("SASSY_HIP_DEBUG_TIMING") != nullptr;
  double last = 0;
  void start() { if (on) last = now_ms(); }
  void mark(const char* what) {
    if (!on) return;
    const double t = now_ms();
    fprintf(stderr, "[sassy-hip] %-18s %8.1f us\n", what, (t - last) * 1e3);
    last = t;
  }
};
static thread_local PhaseMarks g_marks;  // (per host thread: the multi-device searcher runs one worker per device)

// ------------------------------------------------------------------ scan driver
struct ShardView {
  const uint8_t* d_text;   // device buffer (halo first)
  uint64_t text_len;       // bytes in the buffer
  uint64_t halo_len;       // bytes before the first owned block
  uint64_t global_offset;  // global position of d_text[0]
  bool text_start;         // buffer byte 0 is column 0 of the whole text
  bool text_end;           // buffer end is the end of the whole text
  bool adopt_ok = false;   // the caller takes the result as it comes (no reporting modes, no strand merge): it may
                           // stay in the pinned block the kernels wrote it into (ScanOut::pin)
};

struct ScanOut {
  std::vector<Candidate> cands;  // sorted by pos, unconditional (COND resolved) ...
  int64_t conditional_index = -1; // ... except this one, which depends on the previous shard
  int exit_state = kStateDecTrue;
  uint64_t cond_seen = 0;
  // device traceback results (empty without trace): one finished record per candidate, in the
  // same order, whose cigar_off points into `pool`
  std::vector<sassy_hip_Match> matches;
  std::string pool;
  // ... or, adopted (pin.h != nullptr; cands / matches / pool above stay empty), in the pinned block itself
  PinBlock pin;
  ScanOut() = default;
  ScanOut(const ScanOut&) = delete;
  ScanOut& operator=(const ScanOut&) = delete;
  ScanOut(ScanOut&& o) noexcept { *this = std::move(o); }
  ScanOut& operator=(ScanOut&& o) noexcept {
    if (this != &o) {
      if (pin.h) g_pin_pool.adopted_back(pin.cap);
      g_pin_pool.give(pin);
      cands = std::move(o.cands); conditional_index = o.conditional_index; exit_state = o.exit_state; cond_seen = o.cond_seen;
      matches = std::move(o.matches); pool = std::move(o.pool);
      pin = o.pin; o.pin = PinBlock{};
      ext_matches = o.ext_matches; ext_n = o.ext_n; ext_pool = o.ext_pool; ext_pool_len = o.ext_pool_len;
    }
    return *this;
  }
  ~ScanOut() {
    if (pin.h) g_pin_pool.adopted_back(pin.cap);
    g_pin_pool.give(pin);
  }
  const sassy_hip_Match* ext_matches = nullptr;
  size_t ext_n = 0;
  const char* ext_pool = nullptr;
  size_t ext_pool_len = 0;
};
static_assert(sizeof(MatchOut) == sizeof(sassy_hip_Match) && sizeof(MatchOut) == 64, "record layout");

static uint32_t warmup_blocks(uint32_t m, uint32_t k) { return (m + k + 1 + 63) / 64; }

// Prefilter geometry: k+1 disjoint pattern pieces of q rows.  Enabled when the pieces are long
// enough to be selective (expected hit blocks on random DNA: 64*(k+1)/4^q of all blocks).
// mode: the searcher's own setting (sassy_hip_set_prefilter), -1 = the process default (SASSY_HIP_PREFILTER)
static int prefilter_mode(int mode) {
  static const int env = getenv("SASSY_HIP_PREFILTER") ? atoi(getenv("SASSY_HIP_PREFILTER")) : -1;
  return mode >= 0 ? mode : env;
}
static uint32_t filter_piece_len(const PatternPlan& plan, uint32_t k, int mode) {
  const int env = prefilter_mode(mode);
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
static int pair_env() {
  static const int v = getenv("SASSY_HIP_PAIR") ? atoi(getenv("SASSY_HIP_PAIR")) : 1;
  return v;
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

// The three prefilter kernels (scan_kernel.hip): which one evaluates the pieces.
enum FilterKind : uint32_t {
  kFilterGeneric = 1,  // filter_kernel: slot masks in LDS, any profile, <= 255 piece rows
  kFilterPlanes = 2,   // filter_dna_kernel: Dna, <= 8 pieces
  kFilterTable = 3,    // filter_table_kernel: q-gram bit table, Dna / Iupac, 7 <= q <= 9
  kFilterCount = 4,    // filter_count_kernel: q-gram lemma (count the pattern's q-grams per window), Dna / Iupac
};

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
static int stream_geometry(ScanParams& P, uint64_t owned, uint32_t extra_front, uint32_t* grid, int wpc = 16,
                           GeoTuner* tuner = nullptr, const void* tune_text = nullptr, uint64_t tune_len = 0,
                           uint32_t tune_kind = 0) {
  static const int env_wpc = getenv("SASSY_HIP_WAVES_PER_CU") ? atoi(getenv("SASSY_HIP_WAVES_PER_CU")) : 0;
  const uint64_t target_lanes = 256ull * (env_wpc > 0 ? env_wpc : wpc) * 64 * 2;
  uint64_t bpl = (owned + target_lanes - 1) / target_lanes;
  const uint64_t min_bpl = std::max<uint64_t>(8, 6ull * extra_front);
  if (bpl < min_bpl) bpl = min_bpl;
  bpl += bpl & 1;  // even: a staged pair of blocks is then always one aligned 128-byte line
  // The lanes of a wave read addresses bpl * 64 bytes apart and the chip holds ~1.6 rounds of the grid:
  // both the stride and the lane count decide how evenly the HBM channels are loaded, and the kernel
  // time is sensitive to it (bit-plane filter, % of the 8 TB/s roofline: 3.0 GB bpl 88 / 90 / 92 / 94 ->
  // 51 / 64 / 64 / 53; 2.7 GB 78 / 82 / 84 -> 61 / 53 / 58; 2.0 GB 58 / 60 / 64 -> 54 / 63 / 47).  No static
  // rule fits every size (multiples of 6 blocks are never bad but not always best): GeoTuner tries the
  // neighbouring even values during the first searches of a resident text and keeps the fastest;
  // SASSY_HIP_BPL=<n> fixes the value, SASSY_HIP_TUNE=0 keeps the default.
  static const char* env_bpl = getenv("SASSY_HIP_BPL");
  // (opt-in since round 2: with two searches in flight -- the way a stream of searches runs -- the geometry
  // moves the time per search by 0-2 %; it still matters for the latency of a lone search at unlucky sizes,
  // 2.7 GB: 0.80 -> 0.72 ms, which is what SASSY_HIP_TUNE=1 is for; profiles/r02_geometry_sweep.txt)
  // (the callers pass a tuner only when the searcher asks for one: sassy_hip_set_geometry_tuner / SASSY_HIP_TUNE=1)
  if (env_bpl != nullptr && atoll(env_bpl) > 0) bpl = (uint64_t)atoll(env_bpl);
  else if (tuner != nullptr && env_wpc == 0 && owned * 64 >= (256ull << 20)) {
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

// One scan of one buffer (shard or sub-shard) on one lane, in three phases so that several can be
// in flight: prepare() sizes everything and uploads what the pattern needs, enqueue() queues the
// whole kernel pipeline on the lane's stream without waiting, finish() waits for it, grows buffers
// and re-runs on overflow, and turns the device output into resolved reports.
// Two exact paths: the streaming DP over every block, or -- when the pattern splits into k+1
// selective pieces -- prefilter (K0) -> chunk list (K0b) -> DP over the listed chunks (K1-list).
struct ScanJob {
  sassy_SearcherType* S;
  ScanLane& L;
  ShardView sh;
  const PatternPlan& plan;
  uint32_t k;
  bool all_minima;
  const uint8_t* pat;
  bool do_trace;
  uint64_t total_len;
  TextTable texts{};                 // several texts in the buffer (n = 0: one text)
  // multi-pattern search: the hit bitmap was filled by filter_dna_multi_kernel (piece length ext_q);
  // this job only waits for it (ext_wait) and runs chunk list -> DP -> rank -> traceback
  unsigned long long* ext_bitmap = nullptr;
  uint32_t ext_q = 0;
  hipEvent_t ext_wait = nullptr;
  // per-text mode: the chunk descriptors come from the caller (one per text of a block-aligned
  // multi-text buffer); no prefilter, no chunk builder -- list DP -> rank -> traceback
  const ChunkDesc* ext_desc = nullptr;
  uint32_t ext_ndesc = 0;
  // both strands from one pass (whole texts): the prefilter of this, the forward strand's, job also
  // evaluates the Rc strand's pattern rc_pat (= complement(pattern)) on the forward text and marks
  // rc_bitmap in the coordinates of the reversed text; rc_marked tells whether the chosen filter did
  // (bit-plane and counting filters do).  The Rc job then takes that bitmap as ext_bitmap and reads
  // the forward buffer backwards (rev_n = its length): no reversed copy exists.
  unsigned long long* rc_bitmap = nullptr;
  const uint8_t* rc_pat = nullptr;
  bool rc_marked = false;
  bool rc_second_pass = false;       // more than 4 pieces: the Rc pieces get their own filter launch
  ScanParams F2{};
  uint64_t rev_n = 0;
  hipEvent_t wait_for = nullptr;     // pipelining: the previous sub-shard's "filter done"
  bool signal_filter_done = false;   // pipelining: record L.ev_filter_done behind this filter
  bool pipelined = false;            // one of several searches in flight (sassy_hip_search_shard_begin): the
                                     // bit-plane filter takes only half of a CU's wave slots, so that the previous
                                     // search's small tail kernels find room next to it

  static constexpr size_t kCtlHead = 64 + 4 * (size_t)kRankLimit;
  static constexpr uint32_t kTraceWaveMax = 8192;
  // list mode: longest chunk in blocks.  Long runs of candidate blocks (N runs under the Iupac profile,
  // low-complexity stretches) are cut there; every cut costs the next chunk wb warm-up blocks, every uncut run
  // is one lane walking it alone.  8 * wb blocks keep the warm-up at an eighth of the work (m = 32: 16-block
  // chunks, a 4 KiB N run is shared by four lanes instead of one).
  uint32_t maxlen = 128;
  static constexpr uint32_t kSpec = 4096;  // reports the kernels also write into the host buffer
  static constexpr size_t kPinCounts = 0, kPinCounters = 16, kPinFlags = 64;
  static constexpr size_t kPinFlags2 = 68, kPinCount2 = 72;  // dense results: the device's flag word and the count behind the dedup
  static constexpr size_t pin_cands = 128;
  static constexpr size_t pin_recs = pin_cands + (size_t)kSpec * sizeof(Candidate);
  static constexpr size_t pin_ops = pin_recs + (size_t)kSpec * sizeof(MatchOut);

  bool empty = false;
  double t_enter = 0, t_mark = 0;
  uint64_t n_blocks = 0, first_owned = 0, owned = 0, n_words = 0;
  ScanParams P{}, F{};
  uint32_t bucket = 4, q = 0;
  bool filtered = false;
  FilterKind fkind = kFilterGeneric;
  uint32_t count_r = 0, count_w = 0, count_t = 0, count_wpg = 4;  // counting filter: R, window blocks, threshold, waves per workgroup
  uint32_t pair = 0;                                   // paired filter: super-pieces (0: not taken)
  double count_tail = 0;                           // ... and the expected fraction of candidate blocks
  unsigned long long* d_bitmap = nullptr;
  uint32_t* d_counts = nullptr;
  unsigned long long* d_counters = nullptr;
  TraceParams T{}, Tw{};
  uint32_t trace_blocks = 0, wave_blocks = 0, grid = 0, fgrid = 0, desc_cap = 0;
  bool use_wave = false, use_thread = false, ev_scan = false, self_rank = false, tuned = false;
  // fused: the bit-plane filter also runs the chunk DP (filter_dna_kernel<.., FUSED>): no bitmap, no chunk list, no
  // list kernel; no_fuse: this job already fell back to the classic chain
  bool fused = false, no_fuse = false;
  uint32_t counts[2] = {0, 0};  // reports, chunk descriptors
  int timing = 1;

  ScanJob(sassy_SearcherType* S_, ScanLane& L_, const ShardView& sh_, const PatternPlan& plan_, uint32_t k_,
          bool all_, const uint8_t* pat_, bool do_trace_, uint64_t total_len_)
      : S(S_), L(L_), sh(sh_), plan(plan_), k(k_), all_minima(all_), pat(pat_), do_trace(do_trace_),
        total_len(total_len_) {}
  int prepare();
  int enqueue(int attempt);
  int finish(ScanOut& out);
  int finish_once(ScanOut& out, bool& redo);
};

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
  static const bool env_nocut = getenv("SASSY_HIP_ROW_CUT") && atoi(getenv("SASSY_HIP_ROW_CUT")) == 0;
  if (env_nocut) P.flags |= kScanNoRowCut;
  P.alpha = overhang ? S->alpha : 0.0f;
  P.ov_steps = ov_steps;
  P.rev_n = rev_n;
  bucket = plan.nslots <= 4 ? 4 : plan.nslots <= 8 ? 8 : plan.nslots <= 16 ? 16 : plan.nslots <= 32 ? 32 : 64;
  static const int env_sb = getenv("SASSY_HIP_STAGE_BLOCKS") ? atoi(getenv("SASSY_HIP_STAGE_BLOCKS")) : 0;
  P.stage_blocks = env_sb == 1 || env_sb == 2 ? (uint32_t)env_sb : 1u;
  for (int s = 0; s < kMaxSlots; ++s) P.slot_val[s] = plan.slot_val[s];
  q = filter_piece_len(plan, k, S->prefilter);
  // a match that hangs over an end of the text contains only part of the pattern: the pigeonhole
  // argument of the prefilter does not cover it, so overhang searches stream the full DP
  if (overhang) q = 0;
  // Ascii patterns with more than 16 distinct bytes: only the DP kernels carry that many slot masks (or, byte mode,
  // compare bytes instead of looking slots up)
  if (plan.nslots > 16 || plan.bytes) q = 0;
  if (ext_bitmap) q = ext_q;
  if (ext_desc) q = 1;  // list mode without a filter
  // which prefilter kernel (SASSY_HIP_FILTER_KIND=1|2|3|4 forces one where it applies)
  static const int env_kind = getenv("SASSY_HIP_FILTER_KIND") ? atoi(getenv("SASSY_HIP_FILTER_KIND")) : 0;
  const int env_pre = prefilter_mode(S->prefilter);
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
  static const int env_selfrank0 = getenv("SASSY_HIP_SELF_RANK") ? atoi(getenv("SASSY_HIP_SELF_RANK")) : 1;
  static const int env_lin0 = getenv("SASSY_HIP_FILTER_LINEAR") ? atoi(getenv("SASSY_HIP_FILTER_LINEAR")) : 0;
  static const int env_wave0 = getenv("SASSY_HIP_TRACE_WAVE") ? atoi(getenv("SASSY_HIP_TRACE_WAVE")) : 1;
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
  static const int env_iupac_planes = getenv("SASSY_HIP_IUPAC_PLANES") ? atoi(getenv("SASSY_HIP_IUPAC_PLANES")) : 1;
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
  static const bool env_short = !(getenv("SASSY_HIP_SHORT_PIECES") && atoi(getenv("SASSY_HIP_SHORT_PIECES")) == 0);
  const bool short_ok = q == 0 && env_pre < 0 && env_short && fuse_ok && !overhang && !ext_bitmap && !ext_desc && plan.nslots <= 16 &&
                        !plan.bytes && (S->profile == PROFILE_DNA || plain_pattern) && pieces <= 4 && plan.m / pieces == 6;
  // (5-row pieces lose everywhere: m = 11, k = 1 takes 2.6 ms against 1.7 on the streaming DP, m = 15, k = 2 2.2 against 1.2)
  // The paired filter (filter_dna_kernel<.., PAIR>): S = ceil((k+1)/2) super-pieces of 2 Q rows, each with at most one of
  // the k edits -- one half exact, the other half with <= 1 edit right next to it, tested on the bit planes the lane
  // holds.  For the shapes whose k+1 pigeonhole pieces are 5 or 6 rows (m = 23, k = 3; m = 32, k = 4, 5; ...): the fused
  // launch, and only it (what it cannot finish goes to the paths below, as before).  SASSY_HIP_PAIR=0: never; 2: the
  // q-gram counting filter keeps the shapes it is selective for.
  const int env_pair = pair_env();
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
  n_words = filtered ? (n_blocks + 63) / 64 : 0;
  if (int rc = L.d_ctl.reserve(kCtlHead + (filtered && !ext_bitmap && !ext_desc ? (n_words + 2) * 8 : 0))) return rc;
  d_bitmap = ext_bitmap ? ext_bitmap : reinterpret_cast<unsigned long long*>(L.d_ctl.p + kCtlHead);
  if (L.d_cand.cap == 0)
    if (int rc = L.d_cand.reserve(1u << 16)) return rc;
  d_counts = reinterpret_cast<uint32_t*>(L.d_ctl.p);
  d_counters = reinterpret_cast<unsigned long long*>(L.d_ctl.p + 16);
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
    static const int env_wave = getenv("SASSY_HIP_TRACE_WAVE") ? atoi(getenv("SASSY_HIP_TRACE_WAVE")) : 1;
    const uint64_t wstride = (raw + 15) / 16 * 16;
    use_wave = env_wave != 0 && 2ull * k + 3 <= 64 && 4 * pat_bytes + 4 * wstride <= 160 * 1024;
    use_thread = !use_wave || (k <= 6 && !overhang);  // overhang: wave shape or the generic thread shape
    uint64_t stride = raw;
    if ((stride / 4) % 2 == 0) stride += 4;  // odd number of LDS words: conflict-free slices
    if (stride > 0xFFFFFFFFull) return fail(SASSY_HIP_EUNSUPPORTED, "pattern/k too large for the traceback band");
    if (use_thread) {
      // Threads of the thread-per-report launch.  With the slices in LDS (64 per workgroup) as many workgroups as the chip
      // holds at once -- a dense result (10^5 .. 10^6 reports) is bound by how many reports are in flight: 16 384 threads
      // were one wave on a quarter of the SIMDs, 2.3 ms for 743 000 reports.  Slices in global memory: 256 MB of them.
      static const int env_tt = getenv("SASSY_HIP_TRACE_THREADS") ? atoi(getenv("SASSY_HIP_TRACE_THREADS")) : 0;
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
    if (int rc = stream_geometry(P, owned, P.wb, &grid, 16, tuned ? &S->tuner_scan : nullptr, sh.d_text, sh.text_len,
                                 1000u + plan.nwords)) return rc;
    P.lds_per_wave = 4096u * P.stage_blocks + bucket * 512u + plan.nwords * 512u;
    // long patterns: the per-row carries (64 bytes per 32 rows and lane) of four waves no longer fit a workgroup's
    // 160 KiB of LDS -- fewer waves per workgroup then (m <= ~9 800 with one)
    P.waves_per_group = (uint32_t)std::min<size_t>(kWavesPerGroup, (160 * 1024) / P.lds_per_wave);
    if (P.waves_per_group == 0)
      return fail(SASSY_HIP_EUNSUPPORTED, "pattern too long for the LDS carry store (about 9 800 rows)");
    grid = (uint32_t)((P.n_chunks + 64ull * P.waves_per_group - 1) / (64ull * P.waves_per_group));
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
    static const int env_fsb = getenv("SASSY_HIP_FILTER_STAGE_BLOCKS") ? atoi(getenv("SASSY_HIP_FILTER_STAGE_BLOCKS")) : 0;
    F.stage_blocks = env_fsb == 1 || env_fsb == 2 ? (uint32_t)env_fsb : 2u;
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
      static const int env_csb = getenv("SASSY_HIP_COUNT_STAGE_BLOCKS") ? atoi(getenv("SASSY_HIP_COUNT_STAGE_BLOCKS")) : 0;
      F.stage_blocks = env_csb == 1 ? 1u : 2u;  // (whole 128-byte lines per lane and step: read with non-temporal loads)
      const uint32_t per_wave = 4096u * F.stage_blocks + 64u * count_w, table = 1u << (2 * (q + count_r - 1));
      // the table is per workgroup: sixteen waves around one copy where that fits a CU's LDS, else four
      static const int env_wpg = getenv("SASSY_HIP_COUNT_WPG") ? atoi(getenv("SASSY_HIP_COUNT_WPG")) : 0;
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
      static const int env_probe = getenv("SASSY_HIP_FUSED_PROBE") ? atoi(getenv("SASSY_HIP_FUSED_PROBE")) : 0;
      F.fused = 1u | (env_probe == 1 ? 2u : env_probe == 2 ? 6u : 0u);
      F.dp_first_owned = first_owned;
      static const int env_qcap = getenv("SASSY_HIP_FUSED_QCAP") ? atoi(getenv("SASSY_HIP_FUSED_QCAP")) : 0;
      // chunks per wave between two chunk-DP passes: a wave runs a pass when more than cap - 128 are queued (a full
      // batch of 64 lanes), so the queue never overflows; + the count (16 bytes);
      // 4 workgroups per CU still fit the LDS: 4 x 4 x (8192 + 1536 + 16) = 155 904 bytes
      F.fuse_queue_cap = env_qcap > 128 ? (uint32_t)env_qcap : 192u;
      F.lds_per_wave += F.fuse_queue_cap * 8u + 16u;
      static const int env_press = getenv("SASSY_HIP_FUSED_PRESS") ? atoi(getenv("SASSY_HIP_FUSED_PRESS")) : 0;
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
      // SASSY_HIP_FILTER_LDS_PAD=<bytes per workgroup> overrides (0 = none).
      static const int env_pad = getenv("SASSY_HIP_FILTER_LDS_PAD") ? atoi(getenv("SASSY_HIP_FILTER_LDS_PAD")) : -1;
      const uint32_t pad = env_pad >= 0 ? (uint32_t)env_pad : (pipelined ? 24u * 1024u : 0u);
      if (fkind == kFilterPlanes && pad) F.lds_per_wave += pad / 4u / 16u * 16u;
    }
    // the bit-plane filter as a linear stream (filter_dna_linear_kernel): every wave owns one contiguous
    // range of 128-block steps; SASSY_HIP_FILTER_LINEAR=<waves> sets how many waves the text is cut into
    F.lin_steps = 0;
    static const int env_lin = getenv("SASSY_HIP_FILTER_LINEAR") ? atoi(getenv("SASSY_HIP_FILTER_LINEAR")) : 0;
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
    P.lds_per_wave = bucket * 512u + plan.nwords * 512u;
    P.waves_per_group = (uint32_t)std::min<size_t>(kWavesPerGroup, (160 * 1024) / P.lds_per_wave);
    if (P.waves_per_group == 0)
      return fail(SASSY_HIP_EUNSUPPORTED, "pattern too long for the LDS carry store (about 10 000 rows)");
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
  HIP_TRY(hipMemsetAsync(L.d_ctl.p, 0, fused ? 64 : kCtlHead + (filtered && !ext_bitmap && !ext_desc && attempt == 0 ? (n_words + 2) * 8 : 0), L.stream));
  if (ext_wait && attempt == 0) HIP_TRY(hipStreamWaitEvent(L.stream, ext_wait, 0));
  // pipelined sub-shards: this lane's filter starts when the previous sub-shard's filter is done,
  // so that the previous lane's DP / rank / traceback kernels overlap this bandwidth-bound one
  if (wait_for && attempt == 0) HIP_TRY(hipStreamWaitEvent(L.stream, wait_for, 0));
  // (a job that only consumes a bitmap has no filter to time: no events at level 1, each costs ~6 us of stream idle)
  const bool time_head = timing >= 2 || (timing == 1 && !ext_bitmap);
  // (the fused launch carries its events itself: LaunchEvents)
  static const bool env_ext_ev = !(getenv("SASSY_HIP_EXT_EVENTS") && atoi(getenv("SASSY_HIP_EXT_EVENTS")) == 0);
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
    } else if (attempt == 0 && !ext_bitmap && !ext_desc) {  // the hit bitmap does not depend on buffer sizes: build it once
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
    if (!ext_desc) {
      le = launch_build_chunks(d_bitmap, n_words, n_blocks, first_owned, P.wb, fkind == kFilterPlanes || fkind == kFilterCount ? 0u : P.wb, maxlen, L.d_desc.p,
                               d_counts + 1, desc_cap, d_counters + 2, L.stream);
      if (le != hipSuccess) return hip_fail(le, "chunk builder launch");
    }
    P.desc = ext_desc ? ext_desc : L.d_desc.p;
    P.desc_count = d_counts + 1;
    P.desc_cap = desc_cap;
    // multi-word patterns with few chunks: one lane per pattern word instead of one lane per chunk
    // (up to 8192 waves' worth of chunks; beyond that the lane-per-chunk kernel fills the chip anyway)
    P.list_words_max = 0;
    P.list_group_log = 0;
    static const int env_words = getenv("SASSY_HIP_LIST_WORDS") ? atoi(getenv("SASSY_HIP_LIST_WORDS")) : 1;
    if (env_words && !ext_desc && plan.nwords >= 2 && plan.nwords <= 64 && !(P.flags & kScanOverhang)) {
      uint32_t glog = 1;
      while ((1u << glog) < plan.nwords) ++glog;
      P.list_group_log = glog;
      P.list_words_max = (8192u * 64u) >> glog;
    }
    // the descriptor count lives on the device: launch for the capacity, idle waves exit at once
    const uint32_t lgrid = (desc_cap + 64u * P.waves_per_group - 1) / (64u * P.waves_per_group);
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
  static const int env_selfrank = getenv("SASSY_HIP_SELF_RANK") ? atoi(getenv("SASSY_HIP_SELF_RANK")) : 1;
  self_rank = env_selfrank != 0 && do_trace && use_wave && texts.n == 0;
  if (!self_rank) {
    le = launch_rank(L.d_cand.p, d_counts, P.cand_cap, reinterpret_cast<uint32_t*>(L.d_ctl.p + 64),
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
    static const bool env_tprobe = getenv("SASSY_HIP_TRACE_PROBE") != nullptr;
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
    HIP_TRY(hipStreamSynchronize(L.stream));
    const double t_sync1 = now_ms();
    S->stats.host_enqueue_ms += t_sync0 - t_mark;
    S->stats.host_wait_ms += t_sync1 - t_sync0;
    t_mark = t_sync1;
    g_marks.start();
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
    g_marks.mark("event times");
    if (getenv("SASSY_HIP_TRACE_PROBE") && do_trace) {
      std::vector<unsigned long long> all((size_t)wave_blocks * 4 * 8);
      HIP_TRY(hipMemcpy(all.data(), L.d_probe.p, all.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t i = 0; i < all.size(); ++i) pr[i & 7] += all[i];
      const double nrep = (double)std::max<unsigned long long>(1, pr[7]);
      fprintf(stderr, "[sassy-hip] trace waves (us per report): rank %.2f window %.2f fill %.2f walk %.2f out %.2f (%llu reports)\n",
              pr[0] / nrep / 100, pr[1] / nrep / 100, pr[2] / nrep / 100, pr[3] / nrep / 100, pr[4] / nrep / 100, pr[7]);
    }
    if (fused) {
      uint32_t fw = 0;
      memcpy(&fw, L.h_pin + kPinCounts + 4 * kCtlFuseWord, sizeof fw);
      if (fw != 0) {
        if (getenv("SASSY_HIP_DEBUG_FUSED")) fprintf(stderr, "[sassy-hip] fused launch: queue overflow / out-of-order marks (flag %u)\n", fw);
        redo = true;
        return 0;
      }
    }
    bool again = false;
    if (filtered && !fused && counts[1] > desc_cap) {  // more chunks than descriptors fit: grow, rebuild
      if (int rc = L.d_desc.reserve((size_t)counts[1] + 1024)) return rc;
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
          static const bool env_nobig = getenv("SASSY_HIP_BIG_PIN") && atoi(getenv("SASSY_HIP_BIG_PIN")) == 0;
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
          le = launch_rank(L.d_cand.p, d_counts, P.cand_cap, reinterpret_cast<uint32_t*>(L.d_ctl.p + 64), L.d_sorted.p,
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
          static const bool env_nocompact = getenv("SASSY_HIP_COMPACT_CIGARS") && atoi(getenv("SASSY_HIP_COMPACT_CIGARS")) == 0;
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
  static const bool env_noadopt = getenv("SASSY_HIP_ADOPT") && atoi(getenv("SASSY_HIP_ADOPT")) == 0;
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
  g_marks.mark("copy out");
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
  g_marks.mark("sort");

  // ---- seams: reports that depend on how a plateau was entered left of their chunk ----
  bool any_cond = false;
  for (const Candidate& c : out.cands) any_cond |= (c.flags & kCandCond) != 0;
  if (fused && any_cond) {  // the chunk chain that resolves it exists only in the classic path
    if (getenv("SASSY_HIP_DEBUG_FUSED"))
      for (const Candidate& c : out.cands)
        if (c.flags & kCandCond)
          fprintf(stderr, "[sassy-hip] fused launch: conditional report at %llu cost %d (buffer: offset %llu, halo %llu, len %llu, text end %d)\n",
                  (unsigned long long)c.pos, c.cost, (unsigned long long)sh.global_offset, (unsigned long long)sh.halo_len,
                  (unsigned long long)sh.text_len, (int)sh.text_end);
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
      if (getenv("SASSY_HIP_DEBUG_FUSED")) fprintf(stderr, "[sassy-hip] fused launch: exit state undetermined (tail %u %u %u %u)\n", tail[0], tail[1], tail[2], tail[3]);
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
  g_marks.mark("seams");
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

static uint64_t required_halo_bytes(size_t pattern_len, size_t k) {
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
static int run_scan(sassy_SearcherType* S, const ShardView& sh, const PatternPlan& plan, uint32_t k,
                    bool all_minima, const uint8_t* pat, bool do_trace, uint64_t total_len, ScanOut& out) {
  static const int env_lanes = getenv("SASSY_HIP_LANES") ? atoi(getenv("SASSY_HIP_LANES")) : 1;
  static const uint64_t min_sub = getenv("SASSY_HIP_SUBSHARD_MIN") ? strtoull(getenv("SASSY_HIP_SUBSHARD_MIN"), nullptr, 10)
                                                                    : (128ull << 20);
  const uint64_t halo = required_halo_bytes(plan.m, k);
  const uint64_t own0 = sh.halo_len;
  const uint64_t owned_bytes = sh.text_len > own0 ? sh.text_len - own0 : 0;
  uint64_t nl = std::min<uint64_t>(std::min<int>(env_lanes, kMaxLanes), owned_bytes / std::max<uint64_t>(min_sub, 2 * halo + 64));
  if (nl < 2 || filter_piece_len(plan, k, S->prefilter) == 0)
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
    static const bool nowait = getenv("SASSY_HIP_PIPE_NOWAIT") != nullptr;
    job.wait_for = (j >= 1 && !nowait) ? S->lanes[j - 1].ev_filter_done : nullptr;
    job.signal_filter_done = j + 1 < n && !nowait;
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
static int run_scan(sassy_SearcherType* S, const ShardView& sh, const PatternPlan& plan, uint32_t k,
                    bool all_minima, const uint8_t* pat, bool do_trace, uint64_t total_len, ScanOut& out);
static int run_scan_ref_lanes(sassy_SearcherType* S, const uint8_t* d_text, uint64_t n, const PatternPlan& plan, uint32_t k,
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

// Several independent scans (different patterns over the same resident buffer) in flight, one per
// lane: while the GPU runs one pattern's kernels the host already queues the next one's and unpacks
// the previous one's results.  submit() blocks only when every lane is busy; results come back in
// submission order through the callback.
struct ScanQueue {
  struct Slot {
    PatternPlan plan;
    std::vector<uint8_t> pat;
    std::unique_ptr<ScanJob> job;
    uint64_t tag = 0;
    bool busy = false;
  };
  typedef std::function<int(uint64_t tag, ScanOut& so, const PatternPlan& plan, const uint8_t* pat)> Done;
  sassy_SearcherType* S;
  int n_lanes;
  Slot slots[kMaxLanes];
  int head = 0, tail = 0, in_flight = 0;  // ring over the lanes
  bool inputs_marked = false;
  Done done;

  ScanQueue(sassy_SearcherType* S_, Done d) : S(S_), done(std::move(d)) {
    static const int env = getenv("SASSY_HIP_QUEUE_LANES") ? atoi(getenv("SASSY_HIP_QUEUE_LANES")) : kMaxLanes;
    n_lanes = std::max(1, std::min(env, kMaxLanes));
  }
  int drain_one() {
    Slot& sl = slots[head];
    ScanOut so;
    int rc = sl.job->finish(so);
    sl.job.reset();
    sl.busy = false;
    head = (head + 1) % n_lanes;
    --in_flight;
    if (rc) return rc;
    return done(sl.tag, so, sl.plan, sl.pat.data());
  }
  int drain_all() {
    int first = 0;
    while (in_flight) {
      const int rc = drain_one();
      if (rc && !first) first = rc;
    }
    return first;
  }
  int submit(const PatternPlan& plan, const uint8_t* pat, const ShardView& sh, const TextTable& texts, uint32_t k,
             bool all_minima, bool do_trace, uint64_t total_len, uint64_t tag, unsigned long long* ext_bitmap = nullptr,
             uint32_t ext_q = 0, hipEvent_t ext_wait = nullptr, const ChunkDesc* ext_desc = nullptr,
             uint32_t ext_ndesc = 0) {
    if (in_flight == n_lanes)
      if (int rc = drain_one()) return rc;
    if (!inputs_marked) {
      // text uploads / the reverse kernel of this call were queued on the searcher's stream: every
      // other lane waits for them once
      HIP_TRY(hipEventRecord(S->ev_inputs, S->stream));
      for (int l = 1; l < n_lanes; ++l) HIP_TRY(hipStreamWaitEvent(S->lanes[l].stream, S->ev_inputs, 0));
      inputs_marked = true;
    }
    Slot& sl = slots[tail];
    sl.plan = plan;
    sl.pat.assign(pat, pat + plan.m);
    sl.tag = tag;
    sl.job.reset(new ScanJob(S, S->lanes[tail], sh, sl.plan, k, all_minima, sl.pat.data(), do_trace, total_len));
    sl.job->texts = texts;
    sl.job->texts.all_minima = all_minima ? 1u : 0u;
    sl.job->ext_bitmap = ext_bitmap;
    sl.job->ext_q = ext_q;
    sl.job->ext_wait = ext_wait;
    sl.job->ext_desc = ext_desc;
    sl.job->ext_ndesc = ext_ndesc;
    ScanJob& job = *sl.job;
    // the slot counts as in flight only once its kernels are queued: a job whose prepare() / enqueue()
    // failed must never reach finish() (it would read the lane's previous counts and re-run on
    // half-initialised parameters)
    int rc = job.prepare();
    if (rc == 0 && !job.empty) rc = job.enqueue(0);
    if (rc != 0) {
      (void)hipStreamSynchronize(S->lanes[tail].stream);  // whatever part of it was queued
      sl.job.reset();
      return rc;
    }
    sl.busy = true;
    tail = (tail + 1) % n_lanes;
    ++in_flight;
    return 0;
  }
  ~ScanQueue() {  // never leave work in flight behind an error return
    while (in_flight) {
      ScanOut so;
      (void)slots[head].job->finish(so);
      slots[head].job.reset();
      head = (head + 1) % n_lanes;
      --in_flight;
    }
  }
};

// Copies texts[i] (lens[i] bytes) to dst + start[i] and fills the gap up to the next text's start (or
// `total`) with `pad`; several threads when there is enough to copy (a 100 MB read set: 27 -> 3 ms).
static void layout_texts(uint8_t* dst, const uint8_t* const* texts, const size_t* lens, const uint64_t* start, size_t nt,
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
static int layout_and_upload(uint8_t* dst, uint8_t* d_dst, const uint8_t* const* texts, const size_t* lens,
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

// Host view of a multi-text buffer (see TextTable in common.h).  Null = the buffer is one text.
struct HostTexts {
  std::vector<uint64_t> start, len;
};
// [ts, te) of the text report c belongs to, in buffer coordinates
static inline void text_bounds(const HostTexts* ht, const Candidate& c, uint64_t total_len, uint64_t& ts, uint64_t& te,
                               uint64_t& text_idx) {
  if (ht) {
    text_idx = c.flags >> kCandTextShift;
    ts = ht->start[text_idx];
    te = ts + ht->len[text_idx];
  } else {
    text_idx = 0;
    ts = 0;
    te = total_len;
  }
}

// Append the matches of one scan to a result: the device already produced finished records and
// cigar text (trace_kernel.hip); only the pool offsets are rebased.  Returns the index of the
// first appended match.
static int append_matches(ScanOut& so, uint64_t total_len, const PatternPlan& plan, bool without_trace,
                          uint64_t pattern_idx, sassy_hip_Result* R, size_t& first, const HostTexts* ht = nullptr) {
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

// Searcher::search / search_all on one text (reference: src/search.rs:510-525, 685-700, 787-881).
// `text` is a host pointer unless TEXT_ON_DEVICE.
// End-position callback of search_with_fn (reference: src/search.rs:767-784, applied at :895-906).
struct EndFilter {
  sassy_hip_end_filter fn = nullptr;
  void* user = nullptr;
};

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
static int post_filter(sassy_SearcherType* S, ScanOut& so, const PatternPlan& plan, const uint8_t* pat, uint32_t k,
                       int strand, const uint8_t* h_text, const uint8_t* d_text, uint64_t tlen, bool with_trace,
                       const EndFilter& ef, const HostTexts* ht = nullptr) {
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

static int search_text(sassy_SearcherType* S, const uint8_t* pattern, size_t plen, const uint8_t* text,
                       size_t tlen, size_t k, uint32_t flags, uint64_t pattern_idx, bool fwd_strand,
                       bool rc_strand, sassy_hip_Result* R, const EndFilter& ef = EndFilter(),
                       bool already_uploaded = false) {
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
  static const int env_fuse = getenv("SASSY_HIP_RC_FUSED") ? atoi(getenv("SASSY_HIP_RC_FUSED")) : 1;
  // the reference's lane reports (run_scan_ref_lanes): single texts, no overhang; each strand lane by lane
  const uint32_t ref_lanes = (S->ref_lanes == 4 || S->ref_lanes == 8) && std::isnan(S->alpha) ? S->ref_lanes : 0u;
  // Shapes of the paired filter: it exists as the fused launch of ONE strand only, and two of them (the Rc strand's on the
  // reversed copy) beat the forward strand's streaming DP with the Rc marks in it (m = 23, k = 3: 1.4 against 1.8 ms).
  // SASSY_HIP_PAIR_RC=0: as before.
  static const int env_pair_rc = getenv("SASSY_HIP_PAIR_RC") ? atoi(getenv("SASSY_HIP_PAIR_RC")) : 1;
  uint32_t ps_ = 0, pq_ = 0;
  const bool pair_strands = env_pair_rc != 0 && pair_env() != 0 && prefilter_mode(S->prefilter) < 0 && S->fuse && !wo && k <= 0xFFFFu &&
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
  static const bool env_two = !(getenv("SASSY_HIP_STRANDS_IN_FLIGHT") && atoi(getenv("SASSY_HIP_STRANDS_IN_FLIGHT")) == 0);
  static const bool env_one_lane = !getenv("SASSY_HIP_LANES") || atoi(getenv("SASSY_HIP_LANES")) <= 1;
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

// The tail of the one-pass searches of many patterns (pattern-tiled scan, seeded search): d_tiled_list holds
// `count` records (pattern, end position, cost) -- EVERY end position with cost <= k of every pattern, in any
// order, `copies`: possibly several times.  Sort by (pattern, position), apply the report rule to each run
// (sort_kernels.hip), trace the reports with one wavefront each (the report's pattern comes with it), apply the
// searcher's report filters per pattern and append the records to R.
// tt / ht: the buffer holds several texts (device / host tables): a report learns its text from its position,
// reports inside a separator are moved to their text's end (search_all: dropped), the records carry text-relative
// coordinates and the text's index in the buffer.
// defer (search_many over a batch of texts, traced, no report filters, not search_all): the records stay on the
// device -- in the buffers of lane defer->lane -- for assemble_many, nothing is appended to R.
struct ManyDefer {
  int lane = 0;
  ManyPart part{nullptr, nullptr, 0};
  uint32_t str_stride = 0;
};
static int finish_pattern_list(sassy_SearcherType* s, const sassy_hip_Encoded* e, const PatternPlan& plan0,
                               const uint8_t* tptr, const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all,
                               bool wo, uint32_t count, bool copies, sassy_hip_Result* R,
                               const TextTable* tt = nullptr, const HostTexts* ht = nullptr, ManyDefer* defer = nullptr) {
  if (defer && (wo || all || !tt)) return fail(SASSY_HIP_EINVAL, "internal: deferred records need a traced, multi-text search");
  ScanLane& L = s->lanes[0];
  hipStream_t st = s->stream;
  const uint32_t m = (uint32_t)e->plen;
  uint32_t counts[2] = {count, 0};
  g_marks.mark("list: scan");
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
  g_marks.mark("list: sort+rule");
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
    const uint64_t wstride = (band + win + opsb + strb + 15) / 16 * 16;
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
      const int env_tt = getenv("SASSY_HIP_ENCODED_TRACE_THREADS") ? atoi(getenv("SASSY_HIP_ENCODED_TRACE_THREADS")) : -1;
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
    const bool env_nopin = getenv("SASSY_HIP_ENCODED_PIN") && atoi(getenv("SASSY_HIP_ENCODED_PIN")) == 0;  // (per call: tests flip it)
    const bool no_filters = std::isnan(s->max_n_frac) && !s->only_best;
    if (!env_nopin && no_filters && !tt && R->matches.empty() && R->pool.empty() && !R->pin.h && n_rep >= 1024 &&
        text_len < (1ull << 39) && m + k < 0xFFFFu && k < 0x7FFFu && strb % 16 == 0) {
      const size_t n = n_rep;
      ScanLane& LO = s->lanes[2];  // (its record / string buffers take the ordered result)
      if (int rc = LO.d_trace.reserve(n)) return rc;
      if (int rc = LO.d_str.reserve(n * strb)) return rc;
      if (int rc = L.d_sort.reserve(encoded_scratch_bytes(n_rep))) return rc;
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
        g_marks.mark("list: pinned rows");
        return 0;
      }
      (void)hipGetLastError();  // (no pinned block of that size: the host's way)
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
  g_marks.mark("list: trace+copy");
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
    g_marks.mark("list: adopt rows");
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
    if (int rc = list.reserve(std::max<size_t>((size_t)1 << 18, (size_t)got + 1024))) return rc;
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
      if (getenv("SASSY_HIP_DEBUG_ZONES")) fprintf(stderr, "[tiled] list of %u records (attempt %d, capacity %u): too many\n", got, attempt, P.cand_cap);
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
static int search_encoded_tiled(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr,
                                const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all, bool wo,
                                sassy_hip_Result* R, bool* done, const TextTable* tt = nullptr,
                                const HostTexts* ht = nullptr, ManyDefer* defer = nullptr, const TiledPerText* pt = nullptr) {
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
  static const bool dbg = getenv("SASSY_HIP_DEBUG_ZONES") != nullptr;
  if (dbg) fprintf(stderr, "[zones] runs %u ends %u hard %u\n", cnt[0], cnt[1], cnt[2]);
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
  if (dbg) fprintf(stderr, "[zones] %zu zones, %llu bytes\n", n_zones, (unsigned long long)Z);
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
  if (dbg) fprintf(stderr, "[zones] tiled scan ok=%d records %u\n", (int)zok, zc);
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
static void seed_layout(int profile, const uint8_t* const* patterns, size_t npat, uint32_t m, uint32_t k, uint32_t* p_end,
                        uint32_t* p_len) {
  const uint32_t pieces = k + 1, q = m / pieces, spare = m - q * pieces;
  for (uint32_t pc = 0; pc < pieces; ++pc) {
    const uint32_t len = q + (pc < spare ? 1u : 0u);
    p_end[pc] = pc * q + std::min(pc, spare) + len;
    p_len[pc] = std::min(len, kSeedMaxLen);
  }
  if (profile != PROFILE_IUPAC || getenv("SASSY_HIP_SEED_LAYOUT") != nullptr || npat == 0) return;
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
static int search_encoded_seeded(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr,
                                 const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all, bool wo,
                                 sassy_hip_Result* R, bool* done, const TextTable* tt = nullptr,
                                 const HostTexts* ht = nullptr, bool dirty_text = false, ManyDefer* defer = nullptr,
                                 const TiledPerText* ov = nullptr) {
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
    seed_layout(s->profile, rows.data(), npat, m, k, p_end, p_len);
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
  static const bool env_sub = !(getenv("SASSY_HIP_SEED_SUBTEST") && atoi(getenv("SASSY_HIP_SEED_SUBTEST")) == 0);
  static const bool env_narrow = !(getenv("SASSY_HIP_SEED_NARROW") && atoi(getenv("SASSY_HIP_SEED_NARROW")) == 0);
  static const bool env_pos64 = getenv("SASSY_HIP_SEED_POS64") && atoi(getenv("SASSY_HIP_SEED_POS64")) != 0;  // (tests)
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
  static const uint64_t env_waves = getenv("SASSY_HIP_SEED_WAVES") ? (uint64_t)atoll(getenv("SASSY_HIP_SEED_WAVES")) : 0ull;
  // (65 536 waves: 13 rounds of the chip's 5 120 resident waves -- the last round's ragged end is 4 % of config 4 with 16 384;
  // a smaller text: ten steps per wave, at least 4 096 waves -- a workgroup stages 16 KiB of seed bits before its first step:
  // 330 MB of reads in 65 536 waves of 2.5 steps were 3.9 ms for both strands, 2.3 in 16 384)
  const uint64_t steps_total = std::max<uint64_t>(1, (text_len + 2047) / 2048);
  const uint64_t waves = std::min<uint64_t>(env_waves ? env_waves : std::max<uint64_t>(4096, std::min<uint64_t>(65536, steps_total / 10)), steps_total);
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

static void reset_stats(sassy_SearcherType* S) { S->stats = sassy_hip_Stats{}; }
// The synchronous entry points use the searcher's lanes (streams, device buffers, pinned result areas) themselves:
// with a ticket open they would overwrite what its finish() is going to read.
static bool tickets_open(const sassy_SearcherType* s) {
  for (const sassy_hip_Ticket* t : s->lane_ticket)
    if (t) return true;
  return false;
}
// Runs the rest of the scope on the searcher's device and restores the thread's current device afterwards.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(sassy_SearcherType* s) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    if (s->device < 0) s->device = cur;  // first use binds the searcher
    s->bound = true;
    if (s->device != cur && hipSetDevice(s->device) == hipSuccess) prev = cur;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define SASSY_NO_TICKETS(s)                                                                                              \
  do {                                                                                                                   \
    if (tickets_open(s))                                                                                                 \
      return fail(SASSY_HIP_EINVAL, "searches are in flight on this searcher (sassy_hip_search_shard_begin): finish them first"); \
  } while (0)

}  // namespace sassy_hip

// ======================================================================== C-ABI
extern "C" {

const char* sassy_hip_last_error(void) { return g_err.c_str(); }
const char* sassy_hip_version(void) { return "sassy-hip 0.1 (gfx950)"; }

int sassy_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static bool parse_alphabet(const char* alphabet, Profile& pr) {
  std::string a(alphabet ? alphabet : "");
  for (char& c : a) c = (char)tolower((unsigned char)c);
  if (a == "dna") pr = PROFILE_DNA;
  else if (a == "iupac") pr = PROFILE_IUPAC;
  else if (a == "ascii") pr = PROFILE_ASCII;
  else return false;
  return true;
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
  return 0;
}

int sassy_hip_set_reference_lanes(sassy_SearcherType* s, int lanes) {
  if (!s || (lanes != 0 && lanes != 4 && lanes != 8)) return fail(SASSY_HIP_EINVAL, "reference lanes must be 0, 4 or 8");
  s->ref_lanes = (uint32_t)lanes;
  return 0;
}

int sassy_hip_set_geometry_tuner(sassy_SearcherType* s, int on) {
  if (!s) return fail(SASSY_HIP_EINVAL, "null searcher");
  s->tune = on != 0;
  return 0;
}

int sassy_hip_set_pipe_depth(sassy_SearcherType* s, int depth) {
  if (!s || depth < 1 || depth > kMaxLanes) return fail(SASSY_HIP_EINVAL, "pipe depth must be 1 .. 4");
  for (sassy_hip_Ticket* t : s->lane_ticket)
    if (t) return fail(SASSY_HIP_EINVAL, "searches are in flight");
  s->pipe_depth = depth;
  s->last_begun_lane = -1;
  return 0;
}

int sassy_hip_set_timing(sassy_SearcherType* s, int level) {
  if (!s || level < 0 || level > 2) return fail(SASSY_HIP_EINVAL, "timing level must be 0, 1 or 2");
  s->timing = level;
  return 0;
}

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
static bool acgt_only(const uint8_t* p, size_t n) {
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
static double seeded_hit_rate(size_t m, size_t k) {
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
static double seeded_estimate(size_t m, size_t k, size_t n_patterns, uint64_t text_len) {
  const double hits = seeded_hit_rate(m, k) * (double)text_len * (double)n_patterns;
  return 3e-4 + (double)text_len / 3e11 + hits * (m <= 32 ? 8e-12 : 16e-12);
}

static bool many_tiled_wanted(const sassy_SearcherType* s, const size_t* pattern_lens, size_t n_patterns, uint64_t total, size_t k) {
  if (n_patterns == 0 || pattern_lens[0] > 64 || 2 * k + 3 > 64 || n_patterns >= (1u << 24)) return false;
  for (size_t pi = 1; pi < n_patterns; ++pi)
    if (pattern_lens[pi] != pattern_lens[0]) return false;
  const int env_many = getenv("SASSY_HIP_MANY_TILED") ? atoi(getenv("SASSY_HIP_MANY_TILED")) : -1;  // (per call: tests flip it)
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

static int search_many_batched(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                               size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                               size_t k, uint32_t flags, sassy_hip_Result* R, bool& handled, bool tiled_only = false) {
  handled = false;
  static const bool off = getenv("SASSY_HIP_BATCH_TEXTS") && atoi(getenv("SASSY_HIP_BATCH_TEXTS")) == 0;
  if (off || n_texts < 2 || n_patterns == 0 || (flags & SASSY_HIP_TEXT_ON_DEVICE) || s->profile == PROFILE_ASCII) return 0;
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
      g_marks.start();
      if (int rc = s->reserve_stage(total + 64)) return rc;
      hbuf = s->h_stage;
      if (ht.start[0] > 0) memset(hbuf, 'X', ht.start[0]);
      if (int rc = s->d_text.reserve(total + 64)) return rc;
      if (int rc = s->d_tables.reserve(4 * nt)) return rc;
      if (int rc = layout_and_upload(hbuf, s->d_text.p, texts + t0, text_lens + t0, ht.start.data(), nt, total, (uint8_t)'X',
                                     s->stream)) return rc;
      g_marks.mark("batch layout + upload");
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
          const int env_seed = getenv("SASSY_HIP_MANY_SEEDED") ? atoi(getenv("SASSY_HIP_MANY_SEEDED")) : -1;
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
          const bool env_noasm = getenv("SASSY_HIP_MANY_ASSEMBLE") && atoi(getenv("SASSY_HIP_MANY_ASSEMBLE")) == 0;  // (per call: tests flip it)
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
            g_marks.mark("batch: strand rows");
          }
          if (tiled_done && on_device) {
            if (int rc = assemble_many(s, defer[0], defer[1], (uint32_t)nt, d_tab + nt, t0, R)) return rc;
            g_marks.mark("batch: assemble");
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

static int search_many_pertext(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                               size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                               size_t k, uint32_t flags, sassy_hip_Result* R, bool& handled) {
  handled = false;
  static const bool off = getenv("SASSY_HIP_BATCH_TEXTS") && atoi(getenv("SASSY_HIP_BATCH_TEXTS")) == 0;
  if (off || n_texts < 2 || n_patterns == 0 || (flags & SASSY_HIP_TEXT_ON_DEVICE)) return 0;
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
    const bool env_off = getenv("SASSY_HIP_OVERHANG_TILED") && atoi(getenv("SASSY_HIP_OVERHANG_TILED")) == 0;  // (per call: tests flip it)
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
    const bool env_off = getenv("SASSY_HIP_OVERHANG_SEEDED") && atoi(getenv("SASSY_HIP_OVERHANG_SEEDED")) == 0;  // (per call)
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
    g_marks.start();
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
      g_marks.mark("pertext layout");
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
      g_marks.mark("pertext upload");
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
        const bool env_noasm = getenv("SASSY_HIP_MANY_ASSEMBLE") && atoi(getenv("SASSY_HIP_MANY_ASSEMBLE")) == 0;
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
          g_marks.mark("pertext one pass");
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
      g_marks.mark("pertext jobs");
    }
    t0 = t1;
  }
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
  g_marks.mark("many: searches");
  std::stable_sort(R->matches.begin(), R->matches.end(), [](const sassy_hip_Match& a, const sassy_hip_Match& b) {
    if (a.pattern_idx != b.pattern_idx) return a.pattern_idx < b.pattern_idx;
    return a.text_idx < b.text_idx;
  });
  g_marks.mark("many: order");
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
    g_marks.mark("append_matches");
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
  Profile pr;
  if (rc && parse_alphabet(m->alphabet.c_str(), pr) && pr == PROFILE_ASCII)
    return fail(SASSY_HIP_EUNSUPPORTED, "reverse complement is not defined for the ascii alphabet");
  m->rc = rc != 0;
  return 0;
}
int sassy_hip_multi_set_replicated(sassy_hip_Multi* m, int on) {
  if (!m) return fail(SASSY_HIP_EINVAL, "null argument");
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
  for (sassy_hip_Multi::Part& p : m->parts)
    if (p.open_tickets) return fail(SASSY_HIP_EINVAL, "searches are in flight on this multi-searcher: finish them first");
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
long sassy_hip_seed_layout(const char* alphabet, const uint8_t* const* patterns, size_t n_patterns, size_t pattern_len, size_t k,
                           uint32_t* out_end, uint32_t* out_len) {
  if (!alphabet || !patterns || !out_end || !out_len || pattern_len == 0 || pattern_len > 64 || k > 7 || pattern_len / (k + 1) < 1)
    return -1;
  const std::string a(alphabet);
  const int profile = a == "dna" ? PROFILE_DNA : a == "iupac" ? PROFILE_IUPAC : a == "ascii" ? PROFILE_ASCII : -1;
  if (profile < 0) return -1;
  seed_layout(profile, patterns, n_patterns, (uint32_t)pattern_len, (uint32_t)k, out_end, out_len);
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
    // searches in flight, 0.585 ms per search as it is): every filter waiting for the END of the previous one
    // (SASSY_HIP_PIPE_CHAIN=1, kept as a switch: 0.64 -- the filters of two searches fill each other's ramp-up and
    // drain, a strict sequence leaves those bubbles), and the filter as two half launches with the next search
    // waiting for the event in between (0.65).  Holding back a search that is begun right behind another one
    // (device-side delay) changed nothing either: the slower first ~20 searches of a stream (0.67 ms) are the
    // device's clocks coming up, not the phase of the two searches -- 50 searches of any kind in front remove it.
    static const bool env_chain = getenv("SASSY_HIP_PIPE_CHAIN") && atoi(getenv("SASSY_HIP_PIPE_CHAIN")) != 0;
    const int prev = s->last_begun_lane;
    if (env_chain && prev >= 0 && prev != lane && s->lane_ticket[prev] && s->lane_ticket[prev]->job) {
      ScanJob* pj = static_cast<ScanJob*>(s->lane_ticket[prev]->job.get());
      if (pj->filtered && !pj->empty && !pj->ext_bitmap && !pj->ext_desc) job->wait_for = s->lanes[prev].ev_filter_done;
    }
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
static std::vector<int> drop_in_devices() {
  std::vector<int> devs;
  const char* e = getenv("SASSY_HIP_DEVICES");
  if (!e || !*e) return devs;
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
  static const std::vector<int> devs = drop_in_devices();
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
    const uint64_t multi_min = getenv("SASSY_HIP_MULTI_MIN_TEXT") ? strtoull(getenv("SASSY_HIP_MULTI_MIN_TEXT"), nullptr, 10)
                                                                   : (16ull << 20);
    const uint32_t mq = (uint32_t)std::min<size_t>(e->plen / (k + 1), 12);
    const bool multi = s->profile == PROFILE_DNA && std::isnan(s->alpha) && e->patterns.size() >= 8 && k + 1 <= 8 &&
                       mq >= 6 && text_len >= multi_min && (((uintptr_t)tptr) & 15) == 0;
    // Many patterns: the pattern-tiled scan does all of them in one pass, where one scan per pattern pays a kernel
    // chain each.  Measured (tools/bench_encoded.py, config 4's shape; profiles/r02_encoded_paths.txt): the tiled
    // kernel advances 3.8e10 (character x group of 64 patterns) per second; a chain costs ~60 us, or 40 us +
    // 7.8e-14 s per text byte behind the multi-pattern prefilter (long plain-ACGT texts), or 60 us + 5.5e-13 s
    // per byte with its own filter pass (texts with other letters).  SASSY_HIP_TILED=0 / 1 forces the choice.
    const int env_tiled = getenv("SASSY_HIP_TILED") ? atoi(getenv("SASSY_HIP_TILED")) : -1;  // (read per call: tests flip it)
    const double tiled_bias = getenv("SASSY_HIP_TILED_BUDGET") ? atof(getenv("SASSY_HIP_TILED_BUDGET")) : 1.0;
    const bool tiled_ok = s->profile != PROFILE_ASCII && std::isnan(s->alpha) && 2 * k + 3 <= 64 &&
                          e->patterns.size() < (1u << 24) && text_len < (1ull << 40);
    const uint64_t tiled_groups = (e->patterns.size() + 63) / 64;
    const double est_tiled = (double)text_len * (double)tiled_groups / 3.8e10 + 1e-4;
    const double est_chains = (double)e->patterns.size() *
        (multi ? 40e-6 + 7.8e-14 * (double)text_len
               : 60e-6 + (text_len >= multi_min ? 5.5e-13 * (double)text_len : 0.0));
    bool tiled = tiled_ok && e->patterns.size() >= 2 && est_tiled <= tiled_bias * est_chains;
    if (env_tiled >= 0) tiled = tiled_ok && env_tiled != 0;
    // Many patterns, long text, selective pieces: seed -> verify -> report (seed_kernels.hip) reads the text
    // once for all patterns.  Expected cost per (character, pattern): hit rate x window x ~24 operations, against
    // 17 for the pattern-tiled scan (SASSY_HIP_SEEDED=0 / 1 forces the choice).
    const int env_seeded = getenv("SASSY_HIP_SEEDED") ? atoi(getenv("SASSY_HIP_SEEDED")) : -1;
    bool seeded = false;
    // (an Iupac searcher's text with other letters: the seeded search plus the pattern-tiled scan around those letters,
    // search_encoded_seeded / seeded_dirty_zones; SASSY_HIP_SEEDED_DIRTY=0: not for such texts)
    static const bool env_dirty = !(getenv("SASSY_HIP_SEEDED_DIRTY") && atoi(getenv("SASSY_HIP_SEEDED_DIRTY")) == 0);
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
  if (!d_text || !pattern || stride == 0) return fail(SASSY_HIP_EINVAL, "bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  std::vector<uint64_t> pos;
  std::vector<uint8_t> val;
  uint64_t cnt = 0;
  for (uint64_t q = 0;; ++q) {
    const uint64_t p = q * stride + stride / 2;
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
