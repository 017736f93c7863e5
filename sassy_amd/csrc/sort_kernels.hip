// sort_kernels.hip -- reports into result order on the device when there are too many for the counting
// ranker (aux_kernels.hip: more than kRankLimit) -- dense results: plants every few KB, microsatellite
// patterns on repeat-rich text, millions of matches.  The host's std::sort of 7.4e5 (index, position) pairs
// took 63 ms of a 115 ms search; rocPRIM's radix sort (the library GEMM-of-sorting: plain use, as the task
// allows for library primitives) does the same in a fraction of a millisecond.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>  // (rocprim's texture iterator uses memset without including it)

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace sassy_hip {

namespace {
__global__ __launch_bounds__(256) void sort_keys_kernel(const Candidate* __restrict__ cand, uint32_t count,
                                                        unsigned long long* __restrict__ keys, int by_tag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // multi-text buffers never come here; end positions are unique within one search of one strand.
  // by_tag (the pattern-tiled search): the flags' upper 24 bits name the pattern -- (pattern, position) order; by_tag =
  // the bits of a position (40, or fewer when the caller knows the text's length: a radix pass per 8 key bits)
  if (i < count)
    keys[i] = by_tag ? ((unsigned long long)(cand[i].flags >> kCandTextShift) << by_tag) | cand[i].pos : cand[i].pos;
}

// The reference's report rule on a complete, (pattern, position)-sorted list of ALL end positions with cost <= k
// (src/search.rs:1310-1368): inside a run of consecutive positions of one pattern, the rightmost position of
// every plateau that was entered by a decrease and is left by an increase.  A run starts and ends next to a
// cost > k, so its first plateau counts as entered by a decrease and its last as left by an increase.
// The list may hold an entry several times (the seeded search sees a match through each of its intact pieces):
// the first copy stands for all.  all_minima: every distinct entry is a report.
//
// "entered by a decrease" needs the entry in front of the plateau.  A plateau can be as long as the list (poly-A,
// microsatellites, runs of N: 2^26 .. 2^28 entries), so no thread walks it: plateau_heads_kernel marks every entry
// that begins a plateau (not a copy of, and not the same-cost right neighbour of, the entry in front of it), an
// inclusive max-scan hands every entry the index of its plateau's first entry, and the rule looks at that one's
// predecessor.
__global__ __launch_bounds__(256) void plateau_heads_kernel(const Candidate* __restrict__ c, uint32_t count,
                                                            uint32_t* __restrict__ head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint32_t h = 0;
  if (i > 0) {
    const Candidate me = c[i], pv = c[i - 1];
    const bool same_plateau = (pv.flags >> kCandTextShift) == (me.flags >> kCandTextShift) &&
                              (pv.pos == me.pos || (pv.pos + 1 == me.pos && pv.cost == me.cost));
    h = same_plateau ? 0u : i;
  }
  head[i] = h;
}

__global__ __launch_bounds__(256) void flag_reports_kernel(const Candidate* __restrict__ c, uint32_t count,
                                                           unsigned char* __restrict__ keep, int all_minima,
                                                           const uint32_t* __restrict__ head) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Candidate me = c[i];
  const uint32_t tag = me.flags >> kCandTextShift;
  bool report = true;
  if (i > 0) {
    const Candidate pv = c[i - 1];
    if ((pv.flags >> kCandTextShift) == tag && pv.pos == me.pos) report = false;  // a copy
  }
  if (me.flags & kCandCont) report = all_minima != 0;  // its plateau goes on over positions the list leaves out
  if (report && !all_minima) {
    // the next distinct entry (copies are few: the seeded search lists an entry at most once per pattern piece, k + 1 <= 8)
    for (uint32_t j = i + 1; j < count; ++j) {
      const Candidate nx = c[j];
      if ((nx.flags >> kCandTextShift) != tag) break;
      if (nx.pos == me.pos) continue;
      if (nx.pos == me.pos + 1 && nx.cost <= me.cost) report = false;
      break;
    }
  }
  if (report && !all_minima) {  // the entry in front of the plateau: higher cost, or the run starts here
    const uint32_t h = head[i];
    if (h > 0) {
      const Candidate first = c[h], pv = c[h - 1];
      if ((pv.flags >> kCandTextShift) == tag && pv.pos + 1 == first.pos) report = pv.cost > me.cost;
    }
  }
  keep[i] = report ? 1 : 0;
}

// Does the sorted list of one strand's reports need the host's attention?  bit 1: a report is conditional (kCandCond),
// bit 2: a report is a copy of its left neighbour (fused filter: two lanes' windows share columns) or lies in front of
// min_pos (the previous shard's).  With neither (and no failed traceback, bit 0, from the traceback kernels) the rows
// are final as the traceback writes them: the result takes them where they lie.  The flags gather in a DEVICE word
// (an atomic on pinned host memory per wave cost this kernel 27 us for 13 000 reports); the host gets a copy.
__global__ __launch_bounds__(256) void report_flags_kernel(const Candidate* __restrict__ c, const uint32_t* __restrict__ count_ptr,
                                                           uint64_t min_pos, uint32_t* __restrict__ d_flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t count = *count_ptr;
  uint32_t hf = 0;
  if (i < count) {
    const Candidate me = c[i];
    if (me.flags & kCandCond) hf |= 2u;
    if ((me.flags & kCandDrop) || me.pos < min_pos || (i > 0 && c[i - 1].pos == me.pos)) hf |= 4u;
  }
  const unsigned long long b2 = __ballot((hf & 2u) != 0), b4 = __ballot((hf & 4u) != 0);
  if ((b2 | b4) != 0 && (threadIdx.x & 63u) == 0) atomicOr(d_flags, (b2 ? 2u : 0u) | (b4 ? 4u : 0u));
}

// The fused filter's windows may overlap: a report can be in the (sorted) list more than once, and windows that begin
// in the halo report end positions in front of min_pos.  keep[i] = first copy of an owned position.  A copy is
// unconditional when its window saw what settles the plateau state; then the report is certain whatever the other
// copies say: the first copy drops its kCandCond when any copy has none.
__global__ __launch_bounds__(256) void unique_flags_kernel(Candidate* __restrict__ c, uint32_t count, uint64_t min_pos,
                                                           unsigned char* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t pos = c[i].pos;
  const bool first = i == 0 || c[i - 1].pos != pos;
  if (first && (c[i].flags & kCandCond)) {
    for (uint32_t j = i + 1; j < count && c[j].pos == pos; ++j)
      if (!(c[j].flags & kCandCond)) { c[i].flags &= ~kCandCond; break; }
  }
  keep[i] = (first && pos >= min_pos) ? 1 : 0;
}

// Reports of many patterns over a multi-text buffer learn their text (the largest t with start[t] <= position).  A
// report that ends inside the separator behind text t stands for the end-of-text report of text t and is moved
// there; in search_all mode such positions do not exist in a single-text search and are dropped (as
// aux_kernels.hip: rank_scatter_kernel does for the scans of one pattern).
__global__ __launch_bounds__(256) void assign_texts_kernel(Candidate* __restrict__ rep, uint32_t count, const TextTable T,
                                                           uint32_t* __restrict__ report_text) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Candidate v = rep[i];
  uint32_t lo = 0, hi = T.n;  // invariant: start[lo] <= pos < start[hi]
  while (lo + 1 < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (T.start[mid] <= v.pos) lo = mid; else hi = mid;
  }
  const uint64_t te = T.start[lo] + T.len[lo];
  if (v.pos > te + T.ov_steps) {  // (overhang: the virtual columns behind a text are end positions of its own)
    if (T.all_minima) v.flags |= kCandDrop;
    else v.pos = te;
    rep[i] = v;
  }
  report_text[i] = lo;
}

// ---- search_many over a batch of texts, both strands: the records of the two strands' passes into result order ----
// (host.hip: assemble_many).  The result order is the one a stable sort by (pattern, text) of [forward records, Rc
// records] gives; the Rc pass saw the batch reversed: text r of its buffer is text n_texts - 1 - r, and its
// coordinates count from the text's end (src/search.rs:859-873).
// (flip: the Rc pass saw the batch reversed AS A WHOLE -- its text r is text n_texts - 1 - r; else every text was reversed
// in its own slot: the per-text layout of the overhang searches)
__global__ __launch_bounds__(256) void many_keys_kernel(const ManyPart a, const ManyPart b, uint32_t n_texts, int flip,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n + b.n) return;
  const bool rc = i >= a.n;
  const MatchOut* row = rc ? b.rows + (i - a.n) : a.rows + i;
  const unsigned long long text = (rc && flip) ? (unsigned long long)(n_texts - 1) - row->text_idx : row->text_idx;
  keys[i] = (row->pattern_idx << 33) | (text << 1) | (rc ? 1ull : 0ull);
  idx[i] = i;
}

// record j of the result = record idx[j] of the passes, with its final text index, strand and coordinates; its
// cigar string moves to slot j of the result's pool.  16 lanes per record: the 64 bytes of the row and the string
// slot travel as 16-byte pieces.
__global__ __launch_bounds__(256) void many_rows_kernel(const ManyPart a, const ManyPart b, uint32_t n_texts, int flip,
                                                        const uint64_t* __restrict__ text_len, uint64_t first_text,
                                                        const uint32_t* __restrict__ idx, uint32_t str_stride,
                                                        MatchOut* __restrict__ out_rows, char* __restrict__ out_strs,
                                                        uint32_t* __restrict__ flags) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = t >> 4, part = t & 15u;
  if (j >= a.n + b.n) return;
  const uint32_t i = idx[j];
  const bool rc = i >= a.n;
  const uint32_t li = rc ? i - a.n : i;
  const MatchOut* src = (rc ? b.rows : a.rows) + li;
  const char* sstr = (rc ? b.strs : a.strs) + (size_t)li * str_stride;
  for (uint32_t x = part; x < str_stride / 16; x += 16)
    reinterpret_cast<uint4*>(out_strs + (size_t)j * str_stride)[x] = reinterpret_cast<const uint4*>(sstr)[x];
  if (part == 0) {
    MatchOut r = *src;
    if (r.pad_[0] == kTraceFailed) atomicOr(flags, 1u);
    if (rc) {
      const uint64_t tx = flip ? (uint64_t)(n_texts - 1) - r.text_idx : r.text_idx;
      const uint64_t len = text_len[tx], rs = r.text_start, re = r.text_end;
      r.text_idx = tx;
      r.text_start = len - re;
      r.text_end = len - rs;
      r.strand = 1;
    }
    r.text_idx += first_text;
    r.cigar_off = j * str_stride;
    out_rows[j] = r;
  }
}
}  // namespace

// ---------------------------------------------------------------- ordered compaction of Candidate records by byte flags
// out[0 .. *out_count) = the records of in[0 .. count) whose keep byte is set, order kept.  (rocprim::select took 0.6 ms
// for 2.3 M records -- 4 G records/s -- where the sort in front of it takes 0.11 ms per pass: its one-kernel look-back
// chain runs at a fraction of the memory's rate for 16-byte records with byte flags.  Here: a count per tile of 2048
// records, an exclusive scan of the tile counts, a scatter -- three launches, 30 us for the same list.)
namespace {
constexpr uint32_t kCompactTile = 2048;  // records per workgroup: 256 threads x 8
__device__ __forceinline__ uint32_t tile_flags8(const unsigned char* __restrict__ keep, uint32_t count, uint32_t first) {
  // the 8 flags of this thread as bits (first is a multiple of 8; the flag array is padded to whole 256-byte lines)
  uint32_t bits = 0;
  if (first + 8 <= count) {
    const uint2 v = *reinterpret_cast<const uint2*>(keep + first);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bits |= (((v.x >> (8 * j)) & 0xFFu) ? 1u : 0u) << j;
      bits |= (((v.y >> (8 * j)) & 0xFFu) ? 1u : 0u) << (4 + j);
    }
  } else {
    for (uint32_t j = 0; j < 8 && first + j < count; ++j) bits |= (keep[first + j] ? 1u : 0u) << j;
  }
  return bits;
}
__global__ __launch_bounds__(256) void compact_count_kernel(const unsigned char* __restrict__ keep, uint32_t count,
                                                            uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t first = blockIdx.x * kCompactTile + threadIdx.x * 8;
  uint32_t n = first < count ? (uint32_t)__popc(tile_flags8(keep, count, first)) : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
  if ((threadIdx.x & 63u) == 0) wave_sum[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(const Candidate* __restrict__ in, const unsigned char* __restrict__ keep,
                                                              uint32_t count, const uint32_t* __restrict__ tile_first,
                                                              const uint32_t* __restrict__ tile_count, Candidate* __restrict__ out,
                                                              uint32_t* __restrict__ out_count) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t first = blockIdx.x * kCompactTile + threadIdx.x * 8;
  const uint32_t bits = first < count ? tile_flags8(keep, count, first) : 0u;
  const uint32_t mine = (uint32_t)__popc(bits);
  uint32_t incl = mine;  // inclusive prefix within the wave
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d);
    if (lane >= (uint32_t)d) incl += up;
  }
  if (lane == 63) wave_sum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t base = tile_first[blockIdx.x];
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += wave_sum[w];
  uint32_t at = base + incl - mine;
  for (uint32_t b = bits; b; b &= b - 1) out[at++] = in[first + (uint32_t)__builtin_ctz(b)];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *out_count = tile_first[blockIdx.x] + tile_count[blockIdx.x];
}
size_t compact_flagged_scratch(uint32_t count) {
  const size_t tiles = ((size_t)count + kCompactTile - 1) / kCompactTile;
  size_t temp = 0;
  (void)rocprim::exclusive_scan(nullptr, temp, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u, tiles,
                                rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  return 2 * ((tiles * 4 + 255) / 256 * 256) + temp + 256;
}
// (keep: padded to whole 256-byte lines by the callers, 8-byte aligned)
hipError_t compact_flagged(const Candidate* d_in, const unsigned char* d_keep, uint32_t count, Candidate* d_out, uint32_t* d_out_count,
                           void* d_scratch, size_t scratch_bytes, hipStream_t stream) {
  const size_t tiles = ((size_t)count + kCompactTile - 1) / kCompactTile;
  const size_t tb = (tiles * 4 + 255) / 256 * 256;
  if (scratch_bytes < 2 * tb) return hipErrorInvalidValue;
  uint32_t* tile_count = static_cast<uint32_t*>(d_scratch);
  uint32_t* tile_first = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(d_scratch) + tb);
  void* temp = static_cast<unsigned char*>(d_scratch) + 2 * tb;
  size_t temp_bytes = scratch_bytes - 2 * tb;
  hipLaunchKernelGGL(compact_count_kernel, dim3((uint32_t)tiles), dim3(256), 0, stream, d_keep, count, tile_count);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(temp, temp_bytes, tile_count, tile_first, 0u, tiles, rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(compact_scatter_kernel, dim3((uint32_t)tiles), dim3(256), 0, stream, d_in, d_keep, count, tile_first, tile_count,
                     d_out, d_out_count);
  return hipGetLastError();
}
}  // namespace

// Bytes of scratch launch_sort_candidates needs for `count` reports (keys in, keys out, rocPRIM's own).
size_t sort_scratch_bytes(uint32_t count) {
  size_t temp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp, static_cast<unsigned long long*>(nullptr),
                                  static_cast<unsigned long long*>(nullptr), static_cast<Candidate*>(nullptr),
                                  static_cast<Candidate*>(nullptr), (size_t)count, 0, 64, hipStream_t(nullptr));
  return 2 * ((size_t)count * 8 + 256) + temp + 256;
}

// sorted[0 .. count) = cand[0 .. count) by ascending end position (by_tag != 0: by (flags >> 8, end position), the position
// in the key's low by_tag bits: 1 stands for 40).
hipError_t launch_report_flags(const Candidate* d_list, uint32_t max_count, const uint32_t* d_count, uint64_t min_pos,
                               uint32_t* d_flags, hipStream_t stream) {
  if (max_count == 0) return hipSuccess;
  hipLaunchKernelGGL(report_flags_kernel, dim3((max_count + 255) / 256), dim3(256), 0, stream, d_list, d_count, min_pos, d_flags);
  return hipGetLastError();
}

// out[0 .. *d_out_count) = sorted[0 .. count) without copies and without the reports in front of min_pos (order kept).
size_t unique_scratch_bytes(uint32_t count) {
  size_t temp = 0;
  temp = compact_flagged_scratch(count);
  return ((size_t)count + 255) / 256 * 256 + temp + 256;
}
hipError_t launch_unique_reports(Candidate* d_sorted, uint32_t count, uint64_t min_pos, Candidate* d_out, uint32_t* d_out_count,
                                 void* d_scratch, size_t scratch_bytes, hipStream_t stream) {
  if (count == 0) return hipMemsetAsync(d_out_count, 0, 4, stream);
  const size_t flag_bytes = ((size_t)count + 255) / 256 * 256;
  if (scratch_bytes < flag_bytes) return hipErrorInvalidValue;
  unsigned char* keep = static_cast<unsigned char*>(d_scratch);
  hipLaunchKernelGGL(unique_flags_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_sorted, count, min_pos, keep);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  size_t temp_bytes = scratch_bytes - flag_bytes;
  return compact_flagged(d_sorted, keep, count, d_out, d_out_count, keep + flag_bytes, temp_bytes, stream);
}

// key_bits: the keys' significant bits (a radix pass per 8 bits: end positions of a 3 GB text need 32, not 64)
hipError_t launch_sort_candidates(const Candidate* d_cand, Candidate* d_sorted, uint32_t count, void* d_scratch,
                                  size_t scratch_bytes, hipStream_t stream, int by_tag, int key_bits) {
  if (count == 0) return hipSuccess;
  const size_t key_bytes = ((size_t)count * 8 + 255) / 256 * 256;
  if (scratch_bytes < 2 * key_bytes) return hipErrorInvalidValue;
  unsigned long long* keys_in = static_cast<unsigned long long*>(d_scratch);
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(d_scratch) + key_bytes);
  void* temp = static_cast<unsigned char*>(d_scratch) + 2 * key_bytes;
  size_t temp_bytes = scratch_bytes - 2 * key_bytes;
  hipLaunchKernelGGL(sort_keys_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_cand, count, keys_in, by_tag == 1 ? 40 : by_tag);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const unsigned end_bit = (key_bits > 0 && key_bits < 64) ? (unsigned)key_bits : 64u;
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, d_cand, d_sorted, (size_t)count, 0, end_bit, stream);
}

// Bytes of scratch launch_select_reports needs: keep flags, plateau heads (marks, scanned), rocPRIM's own for the
// larger of its two calls.
static size_t heads_bytes(uint32_t count) { return ((size_t)count * 4 + 255) / 256 * 256; }
size_t select_scratch_bytes(uint32_t count) {
  size_t temp = 0, temp2 = 0;
  temp = compact_flagged_scratch(count);
  (void)rocprim::inclusive_scan(nullptr, temp2, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                (size_t)count, rocprim::maximum<uint32_t>(), hipStream_t(nullptr));
  return ((size_t)count + 255) / 256 * 256 + 2 * heads_bytes(count) + std::max(temp, temp2) + 256;
}

// sel[0 .. *sel_count) = the reports among sorted[0 .. count) (flag_reports_kernel), order kept.
// all_minima: the distinct entries.
hipError_t launch_select_reports(const Candidate* d_sorted, uint32_t count, Candidate* d_sel, uint32_t* d_sel_count,
                                 void* d_scratch, size_t scratch_bytes, hipStream_t stream, int all_minima) {
  if (count == 0) return hipMemsetAsync(d_sel_count, 0, 4, stream);
  const size_t flag_bytes = ((size_t)count + 255) / 256 * 256;
  const size_t hb = heads_bytes(count);
  if (scratch_bytes < flag_bytes + 2 * hb) return hipErrorInvalidValue;
  unsigned char* keep = static_cast<unsigned char*>(d_scratch);
  uint32_t* marks = reinterpret_cast<uint32_t*>(keep + flag_bytes);
  uint32_t* heads = reinterpret_cast<uint32_t*>(keep + flag_bytes + hb);
  void* temp = keep + flag_bytes + 2 * hb;
  size_t temp_bytes = scratch_bytes - flag_bytes - 2 * hb;
  hipError_t e;
  if (!all_minima) {
    hipLaunchKernelGGL(plateau_heads_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_sorted, count, marks);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    e = rocprim::inclusive_scan(temp, temp_bytes, marks, heads, (size_t)count, rocprim::maximum<uint32_t>(), stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(flag_reports_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_sorted, count, keep, all_minima,
                     all_minima ? nullptr : heads);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return compact_flagged(d_sorted, keep, count, d_sel, d_sel_count, temp, temp_bytes, stream);
}

// out[0 .. *out_count) = the records of in[0 .. count) whose keep byte is set, order kept (scratch: select_scratch_bytes).
hipError_t launch_compact_candidates(const Candidate* d_in, uint32_t count, const unsigned char* d_keep, Candidate* d_out,
                                     uint32_t* d_out_count, void* d_scratch, size_t scratch_bytes, hipStream_t stream) {
  if (count == 0) return hipMemsetAsync(d_out_count, 0, 4, stream);
  return compact_flagged(d_in, d_keep, count, d_out, d_out_count, d_scratch, scratch_bytes, stream);
}

// keep[i] = record i ends in the INSIDE of its text: behind the first `edge` columns and not behind the text's end (the
// seeded search of an overhang batch: what overhang changes comes from tiled_pertext_kernel's edge segments)
__global__ __launch_bounds__(256) void keep_interior_kernel(const Candidate* __restrict__ rec, uint32_t count, const TextTable T,
                                                            uint32_t edge, unsigned char* __restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint64_t pos = rec[i].pos;
  uint32_t lo = 0, hi = T.n;  // invariant: start[lo] <= pos < start[hi]
  while (lo + 1 < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (T.start[mid] <= pos) lo = mid; else hi = mid;
  }
  const uint64_t rel = pos - T.start[lo];
  const uint64_t len = T.len[lo];
  // (a text of at most `edge` characters is the edge segments' as a whole)
  keep[i] = (len > edge && rel > edge && rel <= len) ? 1 : 0;
}
hipError_t launch_keep_interior(const Candidate* d_rec, uint32_t count, const TextTable& texts, uint32_t edge, unsigned char* d_keep,
                                hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(keep_interior_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_rec, count, texts, edge, d_keep);
  return hipGetLastError();
}

hipError_t launch_assign_texts(Candidate* d_rep, uint32_t count, const TextTable& texts, uint32_t* d_report_text,
                               hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(assign_texts_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_rep, count, texts, d_report_text);
  return hipGetLastError();
}


// ---- host.hip: finish_pattern_list, dense results of search_encoded ----
// The rows of a one-pass search of many patterns as the traceback left them (pattern_idx = the rc-expanded pattern, in
// (pattern, end position) order) -> the records of the result in the order sassy_hip_search_encoded documents and its host
// path sorts into: (pattern_idx mod P, text_start, text_end, cost, strand) -- the key the reference's own differential
// test sorts by (pattern_tiling/search.rs:748-757); the reference reports pattern_idx mod P with strand = Rc for the
// appended reverse complements (tqueries.rs:74-80, trace.rs:444-449).  Two stable radix sorts of (key, index) pairs --
// the minor key (length, cost, strand) first --, then one kernel writes every record and its cigar string to its place.
namespace {
__global__ __launch_bounds__(256) void encoded_minor_keys_kernel(const MatchOut* __restrict__ rows, uint32_t n, uint64_t n_original,
                                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const MatchOut r = rows[i];
  const uint32_t len = (uint32_t)(r.text_end - r.text_start) & 0xFFFFu, cost = (uint32_t)r.cost & 0x7FFFu;
  keys[i] = (len << 16) | (cost << 1) | (r.pattern_idx >= n_original ? 1u : 0u);
  idx[i] = i;
}
__global__ __launch_bounds__(256) void encoded_major_keys_kernel(const MatchOut* __restrict__ rows, uint32_t n, uint64_t n_original,
                                                                 const uint32_t* __restrict__ idx, unsigned long long* __restrict__ keys) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= n) return;
  const MatchOut* r = rows + idx[j];
  keys[j] = ((r->pattern_idx % n_original) << 39) | (r->text_start & ((1ull << 39) - 1ull));
}
// bytes of record j's cigar string in the result's pool: the text and its NUL
__global__ __launch_bounds__(256) void encoded_lens_kernel(const MatchOut* __restrict__ rows, uint32_t n, const uint32_t* __restrict__ idx,
                                                           uint32_t* __restrict__ lens) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j < n) lens[j] = rows[idx[j]].cigar_len + 1u;
}
// record j of the result = row idx[j]; its cigar string goes to offs[j] of a pool WITHOUT the slots' padding (a slot is
// 2 (m + k + 1) + 2 bytes rounded up, a cigar of a 23-mer a dozen: 1.1 GB of the 2.2 GB a guide set's 17 M matches took
// over the PCIe link were NULs).  8 lanes per record, 8 string bytes each; flags[1] = the pool's size.
__global__ __launch_bounds__(256) void encoded_rows_kernel(const MatchOut* __restrict__ rows, const char* __restrict__ strs, uint32_t n,
                                                           uint64_t n_original, const uint32_t* __restrict__ idx, uint32_t str_stride,
                                                           const uint32_t* __restrict__ offs, MatchOut* __restrict__ out_rows,
                                                           char* __restrict__ out_strs, uint32_t* __restrict__ flags) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = t >> 3, part = t & 7u;
  if (j >= n) return;
  const uint32_t i = idx[j];
  const uint32_t len = rows[i].cigar_len + 1u, off = offs[j];
  const char* sstr = strs + (size_t)i * str_stride;
  for (uint32_t x = part * 8u; x < len; x += 64u) {
    const uint32_t e = min(x + 8u, len);
    for (uint32_t y = x; y < e; ++y) out_strs[(size_t)off + y] = y + 1u < len ? sstr[y] : '\0';
  }
  if (part == 0) {
    MatchOut r = rows[i];
    if (r.pad_[0] == kTraceFailed) atomicOr(flags, 1u);
    const uint64_t p = r.pattern_idx;
    r.pattern_idx = p % n_original;
    r.strand = p >= n_original ? 1 : 0;
    r.cigar_off = off;
    out_rows[j] = r;
    if (j + 1 == n) flags[1] = off + len;
  }
}
}  // namespace
size_t encoded_scratch_bytes(uint32_t count) {
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), (size_t)count, 0, 32, hipStream_t(nullptr));
  (void)rocprim::radix_sort_pairs(nullptr, t2, static_cast<unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), (size_t)count, 0, 64, hipStream_t(nullptr));
  size_t t3 = 0;
  (void)rocprim::exclusive_scan(nullptr, t3, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u, (size_t)count,
                                rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  return 2 * (((size_t)count * 8 + 255) / 256 * 256) + 2 * (((size_t)count * 4 + 255) / 256 * 256) + std::max(std::max(t1, t2), t3) + 256;
}
// key_bits: significant bits of (pattern_idx mod P) << 39 | text_start.  d_flags: two words -- [0] a traceback failed,
// [1] bytes of the compacted cigar pool (<= count * str_stride, which d_strs must hold).
hipError_t launch_assemble_encoded(const MatchOut* d_rows_in, const char* d_strs_in, uint32_t count, uint64_t n_original, uint32_t str_stride,
                                   int key_bits, MatchOut* d_rows, char* d_strs, uint32_t* d_flags, void* d_scratch, size_t scratch_bytes,
                                   hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const size_t kb = ((size_t)count * 8 + 255) / 256 * 256, ib = ((size_t)count * 4 + 255) / 256 * 256;
  if (scratch_bytes < 2 * kb + 2 * ib) return hipErrorInvalidValue;
  unsigned char* base = static_cast<unsigned char*>(d_scratch);
  unsigned long long* k64_in = reinterpret_cast<unsigned long long*>(base);
  unsigned long long* k64_out = reinterpret_cast<unsigned long long*>(base + kb);
  uint32_t* k32_in = reinterpret_cast<uint32_t*>(base);        // (the minor keys use the same space, before the major ones)
  uint32_t* k32_out = reinterpret_cast<uint32_t*>(base + kb);
  uint32_t* idx_a = reinterpret_cast<uint32_t*>(base + 2 * kb);
  uint32_t* idx_b = reinterpret_cast<uint32_t*>(base + 2 * kb + ib);
  void* temp = base + 2 * kb + 2 * ib;
  size_t temp_bytes = scratch_bytes - (2 * kb + 2 * ib);
  const uint32_t grid = (count + 255) / 256;
  hipLaunchKernelGGL(encoded_minor_keys_kernel, dim3(grid), dim3(256), 0, stream, d_rows_in, count, n_original, k32_in, idx_a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::radix_sort_pairs(temp, temp_bytes, k32_in, k32_out, idx_a, idx_b, (size_t)count, 0, 32, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(encoded_major_keys_kernel, dim3(grid), dim3(256), 0, stream, d_rows_in, count, n_original, idx_b, k64_in);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::radix_sort_pairs(temp, temp_bytes, k64_in, k64_out, idx_b, idx_a, (size_t)count, 0, (unsigned)std::min(64, std::max(8, key_bits)), stream);
  if (e != hipSuccess) return e;
  // (the key areas are free again: string lengths and their offsets)
  uint32_t* lens = reinterpret_cast<uint32_t*>(base);
  uint32_t* offs = reinterpret_cast<uint32_t*>(base + kb);
  hipLaunchKernelGGL(encoded_lens_kernel, dim3(grid), dim3(256), 0, stream, d_rows_in, count, idx_a, lens);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(temp, temp_bytes, lens, offs, 0u, (size_t)count, rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(encoded_rows_kernel, dim3((uint32_t)(((uint64_t)count * 8 + 255) / 256)), dim3(256), 0, stream, d_rows_in, d_strs_in, count,
                     n_original, idx_a, str_stride, offs, d_rows, d_strs, d_flags);
  return hipGetLastError();
}

// ---- host.hip: ScanJob::finish, dense results of one pattern ----
// The cigar strings of rows[0 .. *count_ptr) out of their slots (str_stride bytes each: 2 (m + k + 1) + 2 rounded up, a
// dozen of them used) into one pool without the padding; the rows learn their new offsets; total[0] = the pool's bytes.
// A dense result's strings were 60 of the 108 MB that crossed the PCIe link for 743 000 matches.
namespace {
__global__ __launch_bounds__(256) void cigar_lens_kernel(const MatchOut* __restrict__ rows, uint32_t max_count,
                                                         const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ lens) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= max_count) return;
  lens[i] = i < *count_ptr ? rows[i].cigar_len + 1u : 0u;
}
__global__ __launch_bounds__(256) void cigar_compact_kernel(MatchOut* __restrict__ rows, const char* __restrict__ strs, uint32_t max_count,
                                                            const uint32_t* __restrict__ count_ptr, uint32_t str_stride,
                                                            const uint32_t* __restrict__ offs, char* __restrict__ out_strs,
                                                            uint32_t* __restrict__ total) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = t >> 2, part = t & 3u;
  const uint32_t n = *count_ptr < max_count ? *count_ptr : max_count;
  if (i >= n) return;
  const uint32_t len = rows[i].cigar_len + 1u, off = offs[i];
  const char* sstr = strs + (size_t)i * str_stride;
  for (uint32_t y = part; y < len; y += 4u) out_strs[(size_t)off + y] = y + 1u < len ? sstr[y] : '\0';
  if (part == 0) {
    // (the other lanes of the record read cigar_len only: the offset is the row's last word to change)
    rows[i].cigar_off = off;
    if (i + 1 == n) total[0] = off + len;
  }
}
}  // namespace
size_t compact_scratch_bytes(uint32_t max_count, uint32_t str_stride) {
  size_t t = 0;
  (void)rocprim::exclusive_scan(nullptr, t, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u, (size_t)max_count,
                                rocprim::plus<uint32_t>(), hipStream_t(nullptr));
  return 2 * (((size_t)max_count * 4 + 255) / 256 * 256) + ((size_t)max_count * str_stride + 255) / 256 * 256 + t + 256;
}
// *d_out_strs = where the compacted pool lies inside the scratch area
hipError_t launch_compact_cigars(MatchOut* d_rows, const char* d_strs, uint32_t max_count, const uint32_t* d_count, uint32_t str_stride,
                                 uint32_t* d_total, void* d_scratch, size_t scratch_bytes, const char** d_out_strs, hipStream_t stream) {
  if (max_count == 0) return hipSuccess;
  const size_t ib = ((size_t)max_count * 4 + 255) / 256 * 256, sb = ((size_t)max_count * str_stride + 255) / 256 * 256;
  if (scratch_bytes < 2 * ib + sb) return hipErrorInvalidValue;
  unsigned char* base = static_cast<unsigned char*>(d_scratch);
  uint32_t* lens = reinterpret_cast<uint32_t*>(base);
  uint32_t* offs = reinterpret_cast<uint32_t*>(base + ib);
  char* out = reinterpret_cast<char*>(base + 2 * ib);
  void* temp = base + 2 * ib + sb;
  size_t temp_bytes = scratch_bytes - (2 * ib + sb);
  hipLaunchKernelGGL(cigar_lens_kernel, dim3((max_count + 255) / 256), dim3(256), 0, stream, d_rows, max_count, d_count, lens);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::exclusive_scan(temp, temp_bytes, lens, offs, 0u, (size_t)max_count, rocprim::plus<uint32_t>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cigar_compact_kernel, dim3((uint32_t)(((uint64_t)max_count * 4 + 255) / 256)), dim3(256), 0, stream, d_rows, d_strs,
                     max_count, d_count, str_stride, offs, out, d_total);
  *d_out_strs = out;
  return hipGetLastError();
}

// ---- host.hip: assemble_many ----
size_t many_scratch_bytes(uint32_t count) {
  size_t temp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp, static_cast<unsigned long long*>(nullptr),
                                  static_cast<unsigned long long*>(nullptr), static_cast<uint32_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), (size_t)count, 0, 64, hipStream_t(nullptr));
  return 2 * (((size_t)count * 8 + 255) / 256 * 256) + 2 * (((size_t)count * 4 + 255) / 256 * 256) + temp + 256;
}
hipError_t launch_assemble_many(const ManyPart& a, const ManyPart& b, uint32_t n_texts, const uint64_t* d_text_len,
                                uint64_t first_text, uint32_t str_stride, MatchOut* d_rows, char* d_strs, uint32_t* d_flags,
                                void* d_scratch, size_t scratch_bytes, hipStream_t stream, int flip) {
  const uint32_t count = a.n + b.n;
  if (count == 0) return hipSuccess;
  const size_t kb = ((size_t)count * 8 + 255) / 256 * 256, ib = ((size_t)count * 4 + 255) / 256 * 256;
  if (scratch_bytes < 2 * kb + 2 * ib) return hipErrorInvalidValue;
  unsigned char* base = static_cast<unsigned char*>(d_scratch);
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(base);
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(base + kb);
  uint32_t* idx_in = reinterpret_cast<uint32_t*>(base + 2 * kb);
  uint32_t* idx_out = reinterpret_cast<uint32_t*>(base + 2 * kb + ib);
  void* temp = base + 2 * kb + 2 * ib;
  size_t temp_bytes = scratch_bytes - (2 * kb + 2 * ib);
  hipLaunchKernelGGL(many_keys_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, a, b, n_texts, flip, keys_in, idx_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (size_t)count, 0, 58, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(many_rows_kernel, dim3((uint32_t)(((uint64_t)count * 16 + 255) / 256)), dim3(256), 0, stream, a, b, n_texts, flip,
                     d_text_len, first_text, idx_out, str_stride, d_rows, d_strs, d_flags);
  return hipGetLastError();
}

}  // namespace sassy_hip
