// host_internal.h -- what the host units of libsassy_hip.so share: the searcher, its lanes and buffers, one scan job,
// the scan queue, the result types, and the functions one unit calls in another.  Units:
//   scan_driver.hip   one pattern over one buffer: ScanJob (prepare / enqueue / finish), sub-shards, strands, search_text
//   many_patterns.hip search_encoded / search_many: pattern-tiled scan, seeded search, lists -> reports -> records
//   multi_device.hip  sassy_hip_multi_*: one text over several devices inside one process
//   c_abi.hip         the C-ABI of include/sassy.h + sassy_hip.h, the switch table, synthetic inputs
#pragma once
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
#include "switches.h"

namespace sassy_hip {
// (thread_local LaunchEvents g_launch_events: defined in scan_driver.hip, declared in common.h)


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
hipError_t launch_compact_chunks(const ChunkDesc* d_regions, const uint32_t* d_region_count, uint32_t n_regions, ChunkDesc* d_desc,
                                 uint32_t* d_desc_count, uint32_t desc_cap, uint32_t* d_fuse_word, hipStream_t stream);
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


extern thread_local std::string g_err;  // (defined in c_abi.hip: sassy_hip_last_error reads it)
inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
inline int hip_fail(hipError_t e, const char* what) {
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
extern PinPool g_pin_pool;  // (defined in scan_driver.hip)

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
  // The control block has a twin: a search clears the OTHER one behind its last kernel (64 bytes, while the host is busy
  // with this search's result), so the next search on this lane starts with its filter instead of a memset launch.
  DevBuf<uint8_t> d_ctl_twin;
  int ctl_cur = 0;                         // 0: d_ctl, 1: d_ctl_twin holds the running search's control block
  bool ctl_clean[2] = {false, false};      // the block's first 64 bytes are zero (cleared behind the previous search)
  hipEvent_t ev_done = nullptr;            // recorded behind a search's last kernel, in front of the twin's clear
  DevBuf<uint32_t> d_flags;     // dense results: "does any record need the host's attention" (report_flags_kernel)
  DevBuf<Candidate> d_cand, d_sorted;
  DevBuf<MatchOut> d_trace;
  DevBuf<ChunkDesc> d_desc, d_regions;   // (d_regions / d_region_count: the counting filter's own chunk list, one region per wave)
  DevBuf<uint32_t> d_region_count;
  DevBuf<uint32_t> d_carry;     // per-row carries of patterns too long for LDS (ScanParams::carry_global)
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
    HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    ready = true;
    return 0;
  }
  void destroy() {
    if (h_up) (void)hipHostFree(h_up);
    h_up = nullptr; h_up_cap = h_up_used = 0;
    d_state.release(); d_scratch.release(); d_str.release(); d_ctl.release(); d_ctl_twin.release(); d_sort.release(); d_scratch2.release(); d_flags.release();
    for (unsigned char*& b : h_bulk) { if (b) (void)hipHostFree(b); b = nullptr; }
    d_cand.release(); d_sorted.release(); d_trace.release(); d_desc.release(); d_regions.release(); d_region_count.release(); d_carry.release();
    d_pattern.release(); d_table.release(); d_rowoff.release(); d_ovtab.release(); d_stash.release();
    if (h_pin) g_pin_pool.give(PinBlock{h_pin, h_pin_dev, h_pin_cap, h_pin_device});
    h_pin = nullptr;
    for (hipEvent_t e : {ev_a, ev_b, ev_c, ev_f, ev_filter_done, ev_done})
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
  // every path-forcing / tuning switch (switches.h): defaults + the environment, read ONCE, here, when the searcher is
  // made; sassy_hip_set_option changes an entry afterwards.  apply_switches() copies the entries that have setters of
  // their own (sassy_hip_set_fused / _reference_lanes / _pipe_depth / _geometry_tuner / _timing) into their members.
  Switches sw = load_switches();
  void apply_switches() {
    fuse = sw.fused != 0;
    ref_lanes = (sw.ref_lanes == 4 || sw.ref_lanes == 8) ? (uint32_t)sw.ref_lanes : 0u;
    pipe_depth = (int)std::max<long>(1, std::min<long>(sw.pipe_depth, kMaxLanes));
    tune = sw.tune != 0;
    timing = (int)sw.timing;
  }
  sassy_SearcherType() { apply_switches(); }
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
  bool fuse = true;
  // sassy_hip_set_reference_lanes: 0 = the definition (one pass), 4 / 8 = the reference binary's lane reports
  // (anything but 4 or 8 in the environment is ignored, as the setter rejects it)
  uint32_t ref_lanes = 0;
  // searches in flight (sassy_hip_search_shard_begin / sassy_hip_search_finish): the ticket that owns each lane
  struct sassy_hip_Ticket* lane_ticket[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  int last_begun_lane = -1;
  int pipe_depth = 2;
  // reporting modes of the reference's Searcher (src/search.rs:442-475)
  float alpha = NAN;             // overhang cost per pattern character (NaN = no overhang), Iupac only
  long max_overhang = -1;        // with_max_overhang(): -1 = none
  bool only_best = false;        // only_best_match(): one match per strand, minimal cost, rightmost end
  float max_n_frac = NAN;        // with_max_n_frac(): NaN = off (the reference's None)
  GeoTuner tuner, tuner_scan;    // prefilter / streaming-DP geometry per resident text
  // the on-line geometry tuner is opt-in (sassy_hip_set_geometry_tuner, SASSY_HIP_TUNE=1): see stream_geometry
  bool tune = false;
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
  int timing = 1;
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

inline double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}


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

inline uint32_t warmup_blocks(uint32_t m, uint32_t k) { return (m + k + 1 + 63) / 64; }

// The three prefilter kernels (scan_kernel.hip): which one evaluates the pieces.
enum FilterKind : uint32_t {
  kFilterGeneric = 1,  // filter_kernel: slot masks in LDS, any profile, <= 255 piece rows
  kFilterPlanes = 2,   // filter_dna_kernel: Dna, <= 8 pieces
  kFilterTable = 3,    // filter_table_kernel: q-gram bit table, Dna / Iupac, 7 <= q <= 9
  kFilterCount = 4,    // filter_count_kernel: q-gram lemma (count the pattern's q-grams per window), Dna / Iupac
};

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
  uint8_t* ctl_base = nullptr;       // this search's control block (the lane's d_ctl or its twin)
  bool ctl_pre_cleared = false;      // ... whose first 64 bytes the previous search cleared
  bool wait_ev_done = false;         // the host waits for L.ev_done (the twin's clear follows it in the stream)
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
  bool rows_declined = false;                          // the few-chunks list kernel found more chunks than it takes: the lane-per-chunk kernel runs them
  bool count_direct = false;                           // counting filter: it files the chunk descriptors itself (no bitmap, no chunk-list launch)
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
    n_lanes = kMaxLanes;
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

// Host view of a multi-text buffer (see TextTable in common.h).  Null = the buffer is one text.
struct HostTexts {
  std::vector<uint64_t> start, len;
};
// [ts, te) of the text report c belongs to, in buffer coordinates
inline void text_bounds(const HostTexts* ht, const Candidate& c, uint64_t total_len, uint64_t& ts, uint64_t& te,
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

// Searcher::search / search_all on one text (reference: src/search.rs:510-525, 685-700, 787-881).
// `text` is a host pointer unless TEXT_ON_DEVICE.
// End-position callback of search_with_fn (reference: src/search.rs:767-784, applied at :895-906).
struct EndFilter {
  sassy_hip_end_filter fn = nullptr;
  void* user = nullptr;
};

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
inline void reset_stats(sassy_SearcherType* S) { S->stats = sassy_hip_Stats{}; }
// The synchronous entry points use the searcher's lanes (streams, device buffers, pinned result areas) themselves:
// with a ticket open they would overwrite what its finish() is going to read.
inline bool tickets_open(const sassy_SearcherType* s) {
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

// ---- functions one unit calls in another ----
struct TiledPerText;  // (many_patterns.hip)
// c_abi.hip
bool parse_alphabet(const char* alphabet, Profile& pr);
// many_patterns.hip
int search_many_batched(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                               size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                               size_t k, uint32_t flags, sassy_hip_Result* R, bool& handled, bool tiled_only = false);
double seeded_hit_rate(size_t m, size_t k);
double seeded_estimate(size_t m, size_t k, size_t n_patterns, uint64_t text_len);
int search_many_pertext(sassy_SearcherType* s, const uint8_t* const* patterns, const size_t* pattern_lens,
                               size_t n_patterns, const uint8_t* const* texts, const size_t* text_lens, size_t n_texts,
                               size_t k, uint32_t flags, sassy_hip_Result* R, bool& handled);
int search_encoded_tiled(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr,
                                const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all, bool wo,
                                sassy_hip_Result* R, bool* done, const TextTable* tt = nullptr,
                                const HostTexts* ht = nullptr, ManyDefer* defer = nullptr, const TiledPerText* pt = nullptr);
int search_encoded_seeded(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr,
                                 const uint8_t* h_text, uint64_t text_len, uint32_t k, bool all, bool wo,
                                 sassy_hip_Result* R, bool* done, const TextTable* tt = nullptr,
                                 const HostTexts* ht = nullptr, bool dirty_text = false, ManyDefer* defer = nullptr,
                                 const TiledPerText* ov = nullptr);
int search_encoded_overhang(sassy_SearcherType* s, const sassy_hip_Encoded* e, const uint8_t* tptr, const uint8_t* h_text,
                            uint64_t text_len, uint32_t k, bool all, bool wo, sassy_hip_Result* R, bool* done);
bool many_tiled_wanted(const sassy_SearcherType* s, const size_t* pattern_lens, size_t n_patterns, uint64_t total, size_t k);
bool acgt_only(const uint8_t* p, size_t n);
// scan_driver.hip
int stream_geometry(ScanParams& P, uint64_t owned, uint32_t extra_front, uint32_t* grid, int wpc = 16,
                    GeoTuner* tuner = nullptr, const void* tune_text = nullptr, uint64_t tune_len = 0, uint32_t tune_kind = 0);
int post_filter(sassy_SearcherType* S, ScanOut& so, const PatternPlan& plan, const uint8_t* pat, uint32_t k,
                       int strand, const uint8_t* h_text, const uint8_t* d_text, uint64_t tlen, bool with_trace,
                       const EndFilter& ef, const HostTexts* ht = nullptr);
int append_matches(ScanOut& so, uint64_t total_len, const PatternPlan& plan, bool without_trace,
                          uint64_t pattern_idx, sassy_hip_Result* R, size_t& first, const HostTexts* ht = nullptr);
int layout_and_upload(uint8_t* dst, uint8_t* d_dst, const uint8_t* const* texts, const size_t* lens,
                             const uint64_t* start, size_t nt, uint64_t total, uint8_t pad, hipStream_t stream);
void layout_texts(uint8_t* dst, const uint8_t* const* texts, const size_t* lens, const uint64_t* start, size_t nt,
                         uint64_t total, uint8_t pad);
int search_text(sassy_SearcherType* S, const uint8_t* pattern, size_t plen, const uint8_t* text,
                       size_t tlen, size_t k, uint32_t flags, uint64_t pattern_idx, bool fwd_strand,
                       bool rc_strand, sassy_hip_Result* R, const EndFilter& ef = EndFilter(),
                       bool already_uploaded = false);
int run_scan(sassy_SearcherType* S, const ShardView& sh, const PatternPlan& plan, uint32_t k,
                    bool all_minima, const uint8_t* pat, bool do_trace, uint64_t total_len, ScanOut& out);
int run_scan_ref_lanes(sassy_SearcherType* S, const uint8_t* d_text, uint64_t n, const PatternPlan& plan, uint32_t k,
                              bool all_minima, const uint8_t* pat, bool do_trace, uint32_t lanes, ScanOut& out);
uint64_t required_halo_bytes(size_t pattern_len, size_t k);


}  // namespace sassy_hip
