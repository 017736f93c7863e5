// common.h -- structs shared by the host side and the HIP kernels of libsassy_hip.so.
#pragma once
#include <atomic>
#include <cstdint>

namespace sassy_hip {

enum Profile : uint32_t { PROFILE_ASCII = 0, PROFILE_DNA = 1, PROFILE_IUPAC = 2 };
// Kernel-side only (ScanParams::profile, template argument): Ascii patterns with more than kMaxSlots distinct bytes.
// No mask per distinct byte: the block's 8 bit planes are the "slots", a row's Eq word is computed from them and the
// row's pattern byte (the row table then holds the bytes themselves): 16 VALU more per row, any pattern.
constexpr uint32_t PROFILE_ASCII_BYTES = 3;

constexpr int kWave = 64;              // gfx950 wavefront
constexpr int kWavesPerGroup = 4;      // 256-thread workgroups, every wave works alone
constexpr uint32_t kCarryListGroups = 1024;  // list_kernel<.., GC>: workgroups that share a long pattern's chunk list
constexpr int kTileBytes = 64 * 128;   // text tile of one wave: 64 lane chunks x 2 blocks x 64 B,
                                       // 16-byte slots XOR-swizzled by (owner>>1)&7
constexpr uint32_t kRegionSlots = 16;  // chunk descriptors a wave of the counting filter can file (count_direct)
constexpr int kMaxSlots = 64;          // profile slots (distinct pattern letters) per search: Dna 4, Iupac <= 16,
                                       // Ascii <= 64 distinct pattern bytes

// Control block (64 bytes, device): u32 [0] reports, [1] chunk descriptors | bytes 16..47: u64
// counters | words 12..15 (tail): the chunk that ends the buffer -- own_lo, exit state, descriptor
// flags, found -- written by the scan / list kernel for the shard seam protocol.
constexpr int kCtlTailWord = 12;
// u32 [2] of the control block: the fused filter (filter_dna_kernel<.., FUSED>) could not finish the search on
// its own -- bit 0: a wave found more chunks than its LDS queue holds.  The host then runs the classic chain
// (bitmap -> chunk list -> list kernel) for this search.
constexpr int kCtlFuseWord = 2;
constexpr uint32_t kFuseOverflow = 1u;

// candidate flags
constexpr uint32_t kCandCond = 1u;     // report depends on the plateau-entry direction left of the chunk
constexpr uint32_t kCandDrop = 2u;     // multi-text buffers: the report lies in a separator, not in a text
constexpr uint32_t kCandCont = 4u;     // lists of all end positions <= k: the cost stays the same over a stretch of
                                       // positions left out behind this one (the inside of a long run of N): not a plateau's end
constexpr int kCandTextShift = 8;      // multi-text buffers: bits 8..31 = index of the text (< 2^24)

// Several texts in one device buffer (search_texts / search_many on many short texts): text t
// occupies [start[t], start[t] + len[t]); the bytes between two texts are >= m+k+1 copies of a
// character that matches nothing ('X' for Iupac).  The scan runs over the whole buffer as if it were
// one text; the tables let the later stages cut the results back into per-text results.
struct TextTable {
  const uint64_t* start;  // device, n entries, ascending
  const uint64_t* len;    // device, n entries
  uint32_t n;             // 0 = the buffer is a single text
  uint32_t all_minima;    // search_all: reports inside a separator are dropped (else: moved to the text end)
  uint32_t per_text;      // 1: block-aligned texts scanned one lane per text -- the reports already carry
                          // their text index and lie where they belong (nothing to look up or move)
  uint32_t ov_steps;      // overhang: end positions up to len + ov_steps behind a text's start are the text's own (virtual
                          // 'N' columns), not positions inside a separator
};
// chunk exit states
constexpr uint8_t kStateDecFalse = 0, kStateDecTrue = 1, kStatePass = 2;

// chunk descriptor flags (list mode)
constexpr uint32_t kDescClearBefore = 1u;  // the block left of own_lo holds no cell <= k
constexpr uint32_t kDescWholeText = 2u;    // per-text mode: the chunk is one whole text (index in pad_), which
                                           // starts at block own_lo: exact start, own end-of-text rule

constexpr uint32_t kDescWindow = 4u;       // fused filter: the chunk's blocks start at any byte -- block b holds the text
                                           // bytes [64 b + shift, ..), shift = pad_ & 63 --, all of them are owned, and
                                           // the DP warms up inside them: columns up to (start + m + k) report nothing

// A chunk of consecutive text blocks handed to one lane of the list-mode DP kernel.  16 bytes.
struct ChunkDesc {
  uint32_t own_lo;   // first owned block (buffer-relative; texts up to 2^32 blocks = 256 GiB)
  uint32_t own_hi;   // one past the last owned block
  uint32_t flags;    // kDesc*
  uint32_t pad_;
};

// scan flags
constexpr uint32_t kScanAllMinima = 1u;  // report every end position with cost <= k
constexpr uint32_t kScanTextStart = 2u;  // buffer byte 0 is the true start of the text (column 0)
constexpr uint32_t kScanTextEnd = 4u;    // buffer end is the true end of the text
constexpr uint32_t kScanPerText = 16u;   // list mode over whole texts: one lane per text of a block-aligned
                                         // multi-text buffer (ScanParams::texts), no prefilter
constexpr uint32_t kScanStash = 64u;     // fused filter: a block that reports keeps its 64 text bytes for the traceback (TextStash)
constexpr uint32_t kScanNoRowCut = 32u;  // DP kernels: compute every pattern row of every block (SASSY_HIP_ROW_CUT=0)
constexpr uint32_t kScanOverhang = 8u;   // overhang (alpha): special left edge at the text start, virtual
                                         // 'N' columns and an extra cost past the text end

// The text under a report, kept by the lane that made it (fused filter: the block is in its registers, and the
// traceback would otherwise fetch the window from a random place of a multi-GB text -- a page walk per report,
// 20 of the traceback kernel's 40 us).  Slot s belongs to the reports whose flags carry (s + 1) << kCandTextShift
// (single-text buffers only: the field is the text index otherwise); the control block's word 3 counts the slots.
struct TextStash {
  uint64_t base;      // buffer offset of text[0]
  uint64_t pad_;
  uint32_t text[16];  // 64 bytes
};
constexpr int kCtlStashWord = 3;

// One (end position, cost) report of the scan kernel.  16 bytes.
struct Candidate {
  uint64_t pos;   // global end position (exclusive end of the match in the text)
  int32_t cost;
  uint32_t flags;
};

struct ScanParams {
  const uint8_t* text;        // device: first byte of the buffer (halo first, 16-byte aligned)
  uint64_t text_len;          // bytes in the buffer
  uint64_t n_blocks;          // ceil(text_len / 64)
  uint64_t first_owned_block; // halo_len / 64
  uint64_t global_offset;     // global position of text[0]
  uint64_t n_chunks;          // lane chunks covering [first_owned_block, n_blocks)
  uint32_t bpl;               // owned blocks per lane chunk
  uint32_t wb;                // warm-up blocks in front of every chunk: 64*wb >= m+k+1
  uint32_t m, k;
  uint32_t nwords;            // ceil(m / 32): 32 pattern rows per carry word
  uint32_t nslots;
  uint32_t flags;
  uint32_t profile;           // Profile enum
  uint32_t lds_per_wave;      // bytes
  uint32_t waves_per_group;   // scan_kernel / list_kernel: wavefronts per workgroup (4; fewer when a long pattern's
                              // per-row carries -- 64 bytes per 32 rows and lane -- would not fit four waves' LDS)
  uint32_t* carry_global;     // != nullptr: the per-row carries of every wave live here (128 * nwords words per wave), not in
                              // LDS -- patterns beyond ~9 800 rows (scan_kernel<.., GC> / list_kernel<.., GC>)
  uint32_t cand_cap;
  uint32_t n_iter;            // iterations of the block loop
  uint32_t stage_blocks;      // 1 or 2: text blocks per lane chunk fetched per staging step
  const uint32_t* row_tab;    // device, 8*nwords words: one byte per pattern row = 2 * its profile
                              // slot (row r of word w: byte r&3 of row_tab[8w + (r>>2)])
  Candidate* cand;            // device, cand_cap entries
  uint32_t* cand_count;       // device counter (keeps counting past cand_cap)
  uint8_t* chunk_state;       // device, n_chunks entries
  unsigned long long* counters; // optional device counters [0]=word rows, [1]=blocks; may be null
  // ---- prefilter (K0) / list mode (K1-list) ----
  uint32_t n_pieces;          // k+1 pattern pieces of piece_len rows each (rows 0 .. n_pieces*piece_len)
  uint32_t piece_len;
  // fast path (k+1 <= 8): slot*2 of row j of pieces 4g..4g+3, one byte per piece, for the shifted
  // rows j < piece_len-1 (piece_tab) and for the last, unshifted row (piece_last); missing pieces
  // repeat piece 0
  uint32_t piece_groups;      // 0: generic path (table in LDS), 1 or 2: fast path
  uint32_t piece_tab[2][12];
  uint32_t piece_last[2];
  // Dna bit-plane filter: per piece, bit j = code bit 0 / code bit 1 of piece row j
  uint32_t piece_planes;      // 1: use filter_dna_kernel (Dna, <= 8 pieces)
  uint32_t lin_steps;         // != 0: filter_dna_linear_kernel, 128-block steps per wave range
  uint32_t group_offset;      // filter_dna_kernel: first workgroup of this launch (the grid may be split in two launches)
  uint32_t piece_bits[8][2];
  // fused mode of filter_dna_kernel (one launch: filter, then the chunk DP of what the wave itself found):
  // fused = 1; dp_first_owned = first block whose end positions this launch reports (first_owned_block is the
  // filter's, which also looks at the last halo blocks); fuse_queue_cap = chunks a wave's LDS queue holds
  uint32_t fused;
  uint32_t fuse_queue_cap;
  uint32_t fuse_press;        // a wave runs a chunk-DP pass between two block pairs once this many chunks are queued
                              // (<= fuse_queue_cap - 128: between two looks every lane queues at most two)
  uint64_t dp_first_owned;
  TextStash* stash;           // fused: the text under the reports (stash_cap slots), or null
  uint32_t stash_cap;
  uint32_t piece_rem[8];      // pattern rows behind piece p: a match that contains the piece exactly, ending
                              // at text position e, ends in [e + rem - k, e + rem + k]  (read as int32: the paired
                              // filter's A-type sub-pieces count from their detection column, piece_len + 2 later)
  uint32_t pair_y[4];         // paired filter: the sibling sub-piece of piece p in the reading order its test uses (rows
                              // forwards for the B behind an A, backwards for the A in front of a B), code bit 0 / code
                              // bit 1 in byte p & 3 of pair_y[2 (p >> 2)] / pair_y[2 (p >> 2) + 1]
  uint32_t pair;              // != 0: the paired filter (filter_dna_kernel<.., PAIR>) with this many super-pieces of
                              // 2 * piece_len rows; pieces 2t / 2t+1 = the halves A / B of super-piece t
  // multi-pattern bit-plane filter (filter_dna_multi_kernel): multi_n patterns of equal length and
  // piece geometry; pattern p's piece_bits at multi_bits[16 p + 2 piece + plane], its hit bitmap at
  // hit_bitmap + p * multi_stride (64-bit words)
  const uint32_t* multi_bits;
  uint32_t multi_n;
  uint32_t multi_long;        // bit p: piece p has piece_len + 1 rows (the pattern's m mod (k+1) spare rows
                              // lengthen the first pieces: fewer chance hits at no loss of exactness)
  uint64_t multi_stride;
  // q-gram table filter: 4^piece_len bits, byte = code & (2^(2q-3)-1), bit = code >> (2q-3)
  const uint8_t* qgram_table; // device; null unless the table / count filter is used
  // q-gram counting filter (filter_count_kernel): piece_len = Q; one byte per (Q+R-1)-gram = how many
  // of its R q-grams occur in the pattern; a match can only end in a block whose last
  // count_window blocks hold >= count_thresh q-gram hits
  uint32_t count_r;           // positions per table lookup (1, 2 or 4)
  uint32_t count_window;      // W = ceil((m + k - Q) / 64) + 1 blocks
  uint32_t count_thresh;      // t = m + 1 - (k+1) Q
  uint32_t count_direct;      // 1: the filter files chunk descriptors itself -- wave w of the launch into desc[w * kRegionSlots ..],
                              // its count into region_count[w]; chunks begin at dp_first_owned or later, long runs leave in
                              // pieces of count_maxlen blocks: no bitmap, no chunk builder (compact_chunks_kernel packs the
                              // regions) -- one strand only (count_rc == 0), whole lines per staging step
  uint32_t count_maxlen;
  const uint32_t* region_count;
  unsigned long long* hit_bitmap;  // one bit per text block: an exact piece occurrence ends in it
  // ---- both strands from one pass over the forward text (reference: the Rc strand is complement(pattern)
  // against the REVERSED text, src/search.rs:813-878) ----
  // Filters: pieces flagged in piece_mirror / the counting filter with count_rc are evaluated on the
  // forward text for the Rc strand's pattern (their strings reversed) and mark, in hit_bitmap_rc, the
  // blocks OF THE REVERSED TEXT (block = reversed column / 64) in which such a match can end.
  unsigned long long* hit_bitmap_rc;
  uint32_t piece_mirror;      // bit p: piece p belongs to the Rc strand
  uint32_t count_rc;          // counting filter: the table holds both strands' q-grams, mark both bitmaps
  // DP kernels (list mode): rev_n != 0 = the text this launch scans is the reverse of the rev_n bytes at
  // `text` (logical byte i = text[rev_n - 1 - i]); never materialised
  uint64_t rev_n;
  unsigned long long* hit_count;   // device counter of hit blocks
  const ChunkDesc* desc;      // list mode: chunk descriptors
  const uint32_t* desc_count; // list mode: number of descriptors (device)
  uint32_t desc_cap;
  // list_words_kernel (a group of 2^list_group_log lanes per chunk, one pattern word each) takes the
  // launch when there are at most list_words_max chunks (0: never), list_kernel otherwise
  uint32_t list_words_max;
  uint32_t list_group_log;
  uint32_t list_rows;         // 1: list_rows_kernel (a lane per BLOCK of a chunk, groups of list_group lanes) takes those
                              // launches instead of list_words_kernel
  uint32_t list_group;        // 4 .. 64 lanes per chunk (list_rows_kernel)
  uint8_t slot_val[kMaxSlots]; // per slot: Dna 2-bit code, Iupac base-set nibble, Ascii byte
  // ---- overhang (kScanOverhang; reference: src/search.rs:347-356, 1274-1282, 1695-1748) ----
  const uint32_t* ov_tab;     // device, nwords words: left-edge vertical deltas at the text start
                              // (row r of word w at bit 31-r, like the carries)
  float alpha;                // cost per overhanging pattern character
  uint32_t ov_steps;          // virtual 'N' columns behind the text end (0 unless kScanTextEnd)
  // ---- per-text mode (kScanPerText): where the texts lie in the buffer ----
  const uint64_t* texts_start;  // device: first byte of text t (a multiple of 64)
  const uint64_t* texts_len;    // device: its length
};

// The pattern-tiled scan (tiled_kernel.hip; reference v2, src/pattern_tiling/search.rs:326-425).
struct TiledParams {
  const uint8_t* text_aligned;    // text - skew: 64-byte aligned (a block load never leaves the text's pages)
  uint32_t skew;                  // (address of the text) mod 64
  uint64_t text_len;
  const unsigned long long* peq;  // device: [classes][npat_padded], bit j of peq[c][p] = row j of pattern p matches class c
  uint32_t npat, npat_padded;     // patterns; padded to a multiple of 64
  uint32_t n_groups;              // npat_padded / 64
  uint32_t m, k;
  uint32_t classes;               // 4: Dna codes (c >> 1) & 3; 16: Iupac base-set nibbles
  uint32_t chunk;                 // bytes of the aligned array a wave owns (a multiple of 64)
  uint32_t warm_blocks;           // 64-byte blocks in front of a chunk: 64 * warm_blocks >= m + k
  uint64_t n_chunks;              // ceil((skew + text_len) / chunk)
  Candidate* cand;                // {pos, cost, flags = pattern << kCandTextShift}
  uint32_t* cand_count;
  uint32_t cand_cap;
  uint32_t cand_chunk;            // != 0: a wave takes its list slots this many at a time (holes: records with position 0;
                                  // tiled_kernel.hip: EmitCursor) -- the caller's list reader skips them
  uint32_t cand_stop;             // lanes stop counting once the counter is beyond this (> cand_cap and > the largest list
                                  // the host would retry with): a 32-bit counter that kept counting could wrap on a
                                  // dense shape and pass for a complete list
  const uint32_t* keep_bits;      // optional: bit p = end position p is wanted (others are computed, not listed)
  // ---- overhang in one pass (tiled_pertext_kernel; reference: the v2 scan takes overhang in its tiled loop,
  // src/pattern_tiling/search.rs:222-323, alpha_pattern :462-472) ----
  // n_texts != 0: text_aligned is a buffer of n_texts texts, text t in whole 64-byte blocks from texts_start[t] on,
  // texts_len[t] characters followed by >= ov_steps + 2 'N' (the virtual columns behind the text are real there, and the
  // end positions of two texts never touch).  Every text starts from the overhang column (ov_vp = the vertical deltas
  // floor((j+1) alpha) - floor(j alpha), ov_cost0 their sum); an end position i > len costs floor(alpha (i - len)) more;
  // positions texts_start[t] + 0 .. len + ov_steps are listed.
  const uint64_t* texts_start;
  const uint64_t* texts_len;
  uint32_t n_texts;
  uint32_t texts_per_wave;
  uint32_t ov_steps;
  float alpha;
  unsigned long long ov_vp;
  int32_t ov_cost0;
  uint32_t edge_cols;             // != 0: only the end positions overhang changes (see tiled_pertext_kernel)
};

// The seeded search of many patterns over a long text (seed_kernels.hip).
constexpr int kSeedPosShift = 27;        // candidate = (seed end position << 27) | (pattern << 3) | piece
constexpr int kSeedWindowDwords = 16;    // verification window: m + 3k + 1 <= 64 characters
constexpr uint32_t kSeedMaxLen = 10;     // longest seed (table of 4^10 entries); longer pieces use their last 10 rows
struct SeedParams {
  const uint8_t* text;            // 16-byte aligned
  uint64_t text_len;
  uint32_t len[2];                // seed lengths of the two tables (0: unused)
  const uint32_t* start[2];       // 4^len + 1 offsets into entries
  const uint32_t* entries[2];     // (pattern << 3) | piece, grouped by seed code
  const void* peq;                // per pattern the match masks of the four Dna codes: 4 x u32 (m <= 32) or 4 x u64
  uint32_t m, k;
  uint64_t rem_packed;            // byte p = pattern rows behind piece p (per-lane lookups: no arrays in the arguments)
  Candidate* out;                 // every (pattern, end position, cost <= k) in a hit's range
  uint32_t* out_count;
  uint32_t out_cap;
  uint32_t out_stop;              // lanes stop counting beyond this (see TiledParams::cand_stop)
  unsigned long long* hit_count;  // optional: [0] table hits, [1] hits that passed the sub-piece test
  // ---- the sub-piece test in front of the verification (m <= 32; sub == nullptr: off) ----
  // A hit says piece p is intact at i.  The other rows of the pattern hold at most k edits, so of any k+1 disjoint
  // sub-pieces of them one is intact too, at most k characters off the seed's diagonal.  The test reads one window
  // of the 2-bit text, win_dwords dwords from the dword that holds character i - win_left.  Sub-piece u of piece p =
  // rows [a, a + len), off = the characters from the window's start to where it lies at its leftmost shift (k left
  // of the seed's diagonal): sub[8 p + u] = 2a | (32 - 2 len) << 8 | 2 (off & 15) << 16 | (off >> 4) << 24 (the shift
  // amounts the test uses; (off & 15) + 2k + len <= 32, off < 16 (win_dwords - 2)); sub[8 p] = 0xFF in its low byte:
  // no test for piece p.
  const uint32_t* sub;
  const uint32_t* packed_text;    // 2-bit Dna codes of the text, 16 characters per dword
  // both tables' entries, table 1 behind table 0: (pattern << 3 | piece, the pattern's packed rows -- row r at bits
  // 2r --, 0); pat_care: 32 bytes per entry, the second half = 11 at the rows the test may compare (concrete
  // bases), 00 elsewhere
  const uint4* entries16;
  uint32_t entries16_off1;        // index of table 1's first entry
  uint32_t win_left, win_dwords;  // (4: the narrow layout, 5: the wide one -- seed_kernels.hip: test_issue)
  uint32_t pat_care;
  uint32_t pos64;                 // positions beyond 32 bits (entries with care words)
  const uint32_t* seed_bits;      // bit c of table t's part (offset bits_off[t] words): some seed of table t ends with the
                                  // min(len, 8) characters c -- staged in LDS, tested before the tables are read
  uint32_t bits_off[2];
  uint32_t separators;            // 1: a multi-text buffer -- text bytes with bit 3 set ('X', the separator) match no row
};

// A finished match record as the trace kernel writes it; same layout as sassy_hip_Match
// (include/sassy_hip.h).  64 bytes.
struct MatchOut {
  uint64_t pattern_idx, text_idx;
  uint64_t text_start, text_end, pattern_start, pattern_end;
  int32_t cost;
  uint8_t strand;
  uint8_t pad_[3];
  uint32_t cigar_off;  // = record index * str_stride: the string slots double as the cigar pool
  uint32_t cigar_len;
};

// Set by the host right in front of a fused filter launch it wants timed; the launcher hands the events to the
// dispatch (hipExtLaunchKernelGGL) and clears them.
struct LaunchEvents {
  hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchEvents g_launch_events;

// One strand's pass of search_many over a batch of texts, records still on the device (sort_kernels.hip: assemble).
struct ManyPart {
  const MatchOut* rows;
  const char* strs;   // record i's cigar string at i * str_stride
  uint32_t n;
};

// Reports are ranked (sorted by end position) on the device up to this many; beyond it the host sorts.
constexpr uint32_t kRankLimit = 32768;
// The trace kernel keeps its 64 per-thread slices (+ the pattern) in LDS up to this many bytes.
constexpr uint32_t kTraceLdsLimit = 128 * 1024;

struct TraceParams {
  const uint8_t* text;      // device buffer the candidates refer to
  uint64_t rev_n;           // != 0: the candidates refer to the reverse of text[0 .. rev_n)
  uint64_t global_offset;   // global position of text[0]
  uint64_t total_len;       // length of the whole text (window end is clipped to it)
  const Candidate* cand;    // reports in output order (ranked by rank_kernel)
  const uint32_t* cand_count;
  uint32_t cand_cap;
  uint32_t m, k;
  uint32_t profile;
  const uint8_t* pattern;   // device copy of the (strand-specific) pattern
  uint32_t pattern_stride;  // != 0: many patterns of m bytes each, pattern_stride bytes apart; a report's pattern is
                            // flags >> kCandTextShift (single-text buffers only) and goes into its row's pattern_idx
  uint8_t* scratch;         // nthreads * scratch_stride bytes (used when the slices do not fit LDS)
  uint32_t scratch_stride;  // bytes per thread: band | window | ops | cigar text
  uint32_t band_bytes;      // (m+1) * (2k+3) * sizeof(cell), rounded up to 4
  uint32_t win_bytes;       // m+k rounded up to 4
  MatchOut* out;            // cand_cap records
  uint8_t* out_str;         // cand_cap * str_stride bytes: NUL-terminated cigar text per record
  uint32_t str_stride;      // >= 2*(m+k+1) + 2
  uint32_t ops_bytes;       // m+k+1 rounded up to 4; the slice ends with str_stride bytes of cigar text
  uint32_t wave_mode;       // 1: trace_wave_kernel (one wavefront per report; slices are per wave)
  uint32_t count_min, count_max;  // the kernel runs only when count_min < number of reports <= count_max
  TextTable texts;          // n = 0 unless the buffer holds several texts
  // overhang (reference: src/trace.rs:36-47, 273-335): use_alpha = 1 switches the window's left
  // column to floor(min(j, max_overhang) * alpha) + max(0, j - max_overhang), pads the window with
  // 'N' behind the text and lets the walk stop at column 0
  uint32_t use_alpha;
  float alpha;
  uint32_t max_overhang;    // 0xFFFFFFFF = none
  // the first host_cap records also go straight to device-mapped pinned host memory
  MatchOut* host_out;
  uint8_t* host_str;
  uint32_t host_cap;
  // trace_wave_kernel ranking its own reports (single text, at most kTraceWaveMax reports): `unsorted` is
  // the append-order list, `cand` the sorted one it fills; host_cand / host_ctl: the host copies the
  // rank kernels would have written (sorted head of the list, 64-byte control block)
  const Candidate* unsorted;
  uint32_t dedup;           // self-ranking: the list may hold a report twice (fused filter: two lanes' chunks share a
                            // block); the later copy becomes a kCandDrop record in the slot behind its twin
  const TextStash* stash;   // single text: reports whose flags name a slot (< stash_cap) find their window there
  uint32_t stash_cap;
  unsigned long long* probe;  // debugging (SASSY_HIP_TRACE_PROBE): 8 counters, 100 MHz ticks per phase summed over the waves
  uint32_t rank_lds;        // self-ranking: up to this many reports the workgroups rank from an LDS copy of the end
                            // positions (8 bytes each behind the four slices; 0: from the list in L2)
  Candidate* host_cand;
  uint4* host_ctl;
  // host_flags (pinned, zeroed by the host before the launch): bit 0 a traceback failed, bit 1 a report is
  // conditional (kCandCond), bit 2 a record was dropped (duplicate / in front of min_pos) -- with none of them set the
  // host takes the records as they lie in the pinned block
  uint32_t* host_flags;
  uint64_t min_pos;         // self-ranking with dedup: reports in front of this end position are not this launch's
  // many patterns over a multi-text buffer (pattern_stride != 0 and texts.n != 0): the flags name the pattern, the
  // text of report c is report_text[c]
  const uint32_t* report_text;
};
constexpr uint32_t kTraceWaveDummy = 128;  // trace_wave_kernel: bytes at the end of a wave's slice that nothing reads
// MatchOut::pad_[0] of a record whose traceback found no ancestor / exceeded the scanned cost
constexpr uint8_t kTraceFailed = 1;

#if defined(__HIPCC__)
// "Has this been done on the calling thread's current device?"  hipFuncSetAttribute acts on the function of ONE device:
// a flag per process left every device but the first at the 64 KiB default (the in-process multi-device searcher runs
// one worker thread per device through the same launchers).  Racing threads at worst set the attribute twice.
struct DeviceOnce {
  std::atomic<uint64_t> mask{0};
  static uint64_t bit() {
    int d = 0;
    (void)hipGetDevice(&d);
    return 1ull << (d & 63);
  }
  bool need() const { return (mask.load(std::memory_order_acquire) & bit()) == 0; }
  void done() { mask.fetch_or(bit(), std::memory_order_release); }
};

// 16 text bytes of a streaming kernel.  NT: as a non-temporal load (global_load_dwordx4 ... nt) -- the text is read
// once, and lines that do not linger in L2 / the Infinity Cache leave the HBM stream 10 % faster (bit-plane filter,
// 3 GB: 0.565 -> 0.506 ms).  Only where a lane takes whole 128-byte lines per step: a kernel that comes back for the
// other half of a line one step later must find it in L2.
template <bool NT>
__device__ __forceinline__ uint4 stream_load16(const uint8_t* p) {
  if constexpr (NT) {
    typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
    const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}
#endif
// which streaming kernels read the text with non-temporal loads (measured per kernel)
#ifndef SASSY_NT_DNA
#define SASSY_NT_DNA 1
#endif
#ifndef SASSY_NT_SCAN
#define SASSY_NT_SCAN 0    // (the streaming DP is VALU-bound and stages half lines: 1.07 -> 1.16 ms with it)
#endif
#ifndef SASSY_NT_COUNT
#define SASSY_NT_COUNT 1   // (with two blocks per staging step, the default since: 0.64 -> 0.57-0.61 ms; with one 0.87)
#endif
#ifndef SASSY_NT_TABLE
#define SASSY_NT_TABLE 0
#endif
#ifndef SASSY_NT_GENERIC
#define SASSY_NT_GENERIC 1
#endif

}  // namespace sassy_hip
