/*
 * sassy_hip.h -- additive C-ABI of libsassy_hip.so (everything the reference exposes only
 * through its Rust API, plus the device-resident entry points the benchmark and the multi-GPU
 * driver need).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions: functions returning int return 0 on success and a negative code on error;
 * sassy_hip_last_error() then holds a message (thread-local).  Nothing here falls back to a CPU
 * implementation: without a usable HIP device the search calls fail with SASSY_HIP_ENODEVICE.
 */
#ifndef SASSY_HIP_H
#define SASSY_HIP_H

#include "sassy.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SASSY_HIP_EINVAL (-1)    /* bad argument (null pointer, invalid pattern, ...) */
#define SASSY_HIP_ENODEVICE (-2) /* no usable HIP device / HIP runtime error */
#define SASSY_HIP_EUNSUPPORTED (-3)
#define SASSY_HIP_ENOMEM (-4)

/* search flags */
#define SASSY_HIP_ALL_MINIMA 1u     /* Searcher::search_all (src/search.rs:685-700) */
#define SASSY_HIP_WITHOUT_TRACE 2u  /* Searcher::without_trace (src/search.rs:448-451,1464-1475) */
#define SASSY_HIP_TEXT_ON_DEVICE 4u /* `text` is a device pointer (e.g. a torch tensor's data_ptr), 16-byte aligned; the
                                       kernels read whole 64-byte blocks: the allocation must be readable up to the
                                       next multiple of 64 bytes behind the text (any hipMalloc'ed buffer is -- a
                                       sub-range that ends exactly at the end of an allocation of a non-multiple size
                                       is not) */
#define SASSY_HIP_TEXT_UNCHANGED 8u /* with TEXT_ON_DEVICE: the bytes at `text` are the same as in this searcher's
                                       previous call with this pointer and length (many patterns, one resident
                                       text): the reversed copy the Rc strand scans is reused, not rebuilt */

/* Full match record = reference Match (src/search.rs:35-62).  cigar is the SAM text the
 * reference's Cigar::to_string gives ("3=1X"), stored in the result's string pool.
 * without_trace: text_start = pattern_start = UINT64_MAX and an empty cigar, as the reference. */
typedef struct sassy_hip_Match {
  uint64_t pattern_idx;
  uint64_t text_idx;
  uint64_t text_start;
  uint64_t text_end;
  uint64_t pattern_start;
  uint64_t pattern_end;
  int32_t cost;
  uint8_t strand; /* 0 = Fwd, 1 = Rc */
  uint8_t pad_[3];
  uint32_t cigar_off; /* offset of the NUL-terminated cigar string in the pool (the pool's bytes between the strings
                         are unspecified) */
  uint32_t cigar_len;
} sassy_hip_Match;

typedef struct sassy_hip_Result sassy_hip_Result;       /* opaque, owns matches + cigar pool */
typedef struct sassy_hip_Encoded sassy_hip_Encoded;     /* opaque EncodedPatterns */

/* Per-call statistics of the last search on a searcher (for bench.py's roofline object). */
typedef struct sassy_hip_Stats {
  double scan_ms;        /* HIP-event time of the scan path (filter + chunk list + DP, or the streaming
                            scan), all strands; 0 unless timed (sassy_hip_set_timing) */
  double trace_ms;       /* HIP-event time of report ranking + traceback kernels (timing level 2) */
  double total_ms;       /* host wall time of the whole call */
  uint64_t text_bytes;   /* algorithmic bytes scanned (text_len per strand) */
  uint64_t scan_launches;
  uint64_t candidates;   /* (end_pos, cost) records produced by the scan */
  uint64_t cond_resolved;/* candidates whose plateau-entry direction needed the chunk-state chain */
  uint64_t chunks;       /* lane chunks the text was cut into */
  uint64_t blocks;       /* 64-byte text blocks visited incl. warm-up */
  uint64_t word_rows;    /* DP word-rows computed (0 unless the kernel was built with counters) */
  uint32_t blocks_per_chunk;
  uint32_t warmup_blocks;
  uint32_t grid;
  uint32_t filtered;     /* 0: DP over every block (scan_kernel); 1: prefilter (filter_kernel) -> chunk list ->
                            DP on the listed chunks; 2: the same with the Dna bit-plane prefilter (filter_dna_kernel);
                            3: q-gram piece table (filter_table_kernel); 4: q-gram counting (filter_count_kernel);
                            5: the pattern-tiled scan of search_encoded (tiled_kernel: all patterns in one pass);
                            6: the seeded search of search_encoded (seed_kernels: seed table lookups, one lane per hit) */
  double filter_ms;      /* HIP-event time of the prefilter kernel (part of scan_ms) */
  uint64_t hit_blocks;   /* text blocks in which an exact pattern piece ends */
  uint32_t piece_len;    /* rows per pattern piece (k+1 pieces) / q-gram length Q (filtered = 4), 0 when unfiltered */
  uint32_t fused;        /* 1: the bit-plane prefilter ran the chunk DP of what it found itself (one launch for
                            filter + chunk list + DP; sassy_hip_set_fused) */
  double host_enqueue_ms; /* host wall time spent queueing work on the stream */
  double host_wait_ms;    /* host wall time blocked in the stream synchronisation */
  double host_post_ms;    /* host wall time after it: sort, seams, cigar strings, result records */
  uint64_t live_blocks;   /* blocks whose last row passed the cheap "may hold a cell <= k" test and were
                             walked column by column (0 unless counters are enabled) */
  uint32_t pair;          /* != 0: the fused launch ran the PAIRED filter with this many super-pieces of 2 * piece_len
                             rows (one half exact, the other half with <= 1 edit next to it; SASSY_HIP_PAIR=0: never) */
  uint32_t reserved_;
} sassy_hip_Stats;

const char *sassy_hip_last_error(void);
const char *sassy_hip_version(void);
int sassy_hip_device_count(void); /* number of visible HIP devices, 0 if none / no runtime */

/* Mirrors Searcher::new(rc, alpha) (src/search.rs:486-503) without aborting: NULL on error.
 * alpha = NAN: no overhang.  Otherwise (Iupac only, 0 <= alpha <= 1; src/search.rs:373-400) the
 * pattern may hang over either end of the text at alpha per overhanging character: matches then
 * carry pattern_start > 0 / pattern_end < pattern_len.  Overhang searches stream the full DP (the
 * pigeonhole prefilter does not cover partial patterns) and report nothing for an empty text. */
sassy_SearcherType *sassy_hip_searcher_new(const char *alphabet, bool rc, float alpha);
/* The HIP device a searcher works on.  A searcher binds itself to the calling thread's current device at its first
 * search (HIP's current device is per host thread); sassy_hip_set_device chooses one before that.  From then on
 * every entry point runs on that device, whatever thread calls it, and restores the thread's current device
 * before it returns.  sassy_hip_get_device: -1 while unbound. */
int sassy_hip_set_device(sassy_SearcherType *s, int device);
int sassy_hip_get_device(const sassy_SearcherType *s);
/* Use an existing HIP stream (hipStream_t) for all work of this searcher; NULL = own stream. */
int sassy_hip_set_stream(sassy_SearcherType *s, void *hip_stream);
int sassy_hip_get_stats(const sassy_SearcherType *s, sassy_hip_Stats *out);
/* HIP-event timing behind the stats: 0 = none, 1 = the dominant kernel only (prefilter, or the
 * streaming scan when unfiltered; default), 2 = every phase (scan_ms, filter_ms, trace_ms).  Each
 * recorded event costs a few microseconds of stream idle time. */
int sassy_hip_set_timing(sassy_SearcherType *s, int level);
/* The switch table (sassy_amd/csrc/switches.h): every switch that forces a kernel path or sets a tuning
 * value.  A searcher fills its copy ONCE, when it is made: the defaults, then the environment variables
 * SASSY_HIP_<NAME> that are set -- no search entry point reads the environment.  sassy_hip_set_option changes one entry
 * of an existing searcher (name: lower case, without the prefix -- "fused", "filter_kind", "pair", ...; refused while
 * searches are in flight), sassy_hip_get_option reads one, sassy_hip_option_table returns "name<TAB>default<TAB>what it
 * does" lines for all of them.  All settings give the same matches: the switches exist so that every path can be
 * checked against the oracle (tests/test_gpu_parity.py) and timed apart (tools/). */
int sassy_hip_set_option(sassy_SearcherType *s, const char *name, long value);
int sassy_hip_get_option(const sassy_SearcherType *s, const char *name, long *value);
const char *sassy_hip_option_table(void);
/* Which scan path a searcher takes: -1 = the library's choice (exact prefilter where the pattern's pieces are
 * selective, the streaming DP otherwise; process-wide override: SASSY_HIP_PREFILTER), 0 = always the streaming
 * DP over every block, 1 = prefilter also with short pieces.  All give the same matches; the setting exists so
 * that the paths can be checked against each other (tests) and timed apart. */
int sassy_hip_set_prefilter(sassy_SearcherType *s, int mode);
/* The bit-plane prefilter (Dna, <= 8 pieces, one strand, one text) can finish the scan in its own launch: every
 * wavefront runs the chunk DP over the match-end blocks it found itself when it has streamed its text range
 * (no hit bitmap, no chunk-list kernel, no list kernel).  on = 1 (default; process-wide: SASSY_HIP_FUSED=0 turns it
 * off), 0 = always the classic chain.  Same matches either way; stats.fused tells which one ran.
 * An Iupac searcher takes the same launch when its pattern holds plain A C G T only (<= 4 pieces), on ANY text: the lane
 * that owns a block checks it for other letters and queues the columns a match touching them can end in; the chunk DP of
 * such a launch builds the Iupac profile's masks; the inside of a run of N is walked over, not searched (cost 0
 * everywhere: nothing to report under the report rule) -- SASSY_HIP_IUPAC_PLANES=0 turns the launch off.
 * What the fused launch cannot finish alone (a report on a long FLAT plateau of cost > 0 whose beginning no window
 * sees, a shard's exit state on such a plateau) sends that one search to the classic chain. */
int sassy_hip_set_fused(sassy_SearcherType *s, int on);
/* Which reports a search of ONE text returns on low-complexity text (sassy_hip_search, the drop-in search):
 * 0 (default) = the definition -- one left-to-right pass over the text, independent of any chunking;
 * 4 / 8 = what the reference binary built for AVX2 / AVX-512 returns: it cuts the text into 4 / 8 lanes that each
 * start with decreasing = true (src/search.rs:1016-1056, 1202-1240), which adds reports for <=k plateaus that
 * were entered by an increase left of a lane's start (never on random text).  Process default:
 * SASSY_HIP_REF_LANES.  Not applied with overhang, nor to shards.  search_many / search_encoded need no such
 * mode: with several patterns or texts the reference gives every lane a whole text (chunk_offset_blocks = 0,
 * src/search.rs:1034-1048; the v2 scan keeps a pattern per lane), so its reports ARE the definition's there. */
int sassy_hip_set_reference_lanes(sassy_SearcherType *s, int lanes);
/* On-line tuner of the streaming kernels' lane-chunk length for a resident text (off by default, or
 * SASSY_HIP_TUNE=1): the first ~36 searches of a (text, filter kind) try neighbouring geometries -- each a
 * complete, exact search -- and the rest use the fastest.  Worth it for the latency of lone searches on one
 * text at sizes where the default geometry is unlucky (up to 10 %); with searches in flight it moves the time
 * per search by 0-2 %. */
int sassy_hip_set_geometry_tuner(sassy_SearcherType *s, int on);
/* Count DP word-rows / blocks in the scan kernel (stats.word_rows, stats.blocks); off by default. */
int sassy_hip_enable_counters(sassy_SearcherType *s, int on);

/* Reporting modes of the reference's Searcher builder, applied per strand by sassy_hip_search,
 * sassy_hip_search_with_fn and sassy_hip_search_encoded (not by search_shard):
 *   only_best_match()  (src/search.rs:442-446, 1392-1412): one match per strand -- minimal cost,
 *                      rightmost end position;
 *   with_max_n_frac(f) (src/search.rs:454-475, src/n_filter.rs): drop matches whose text span holds
 *                      more than the fraction f of 'N'/'n'; f = 1.0 or NAN switches it off. */
int sassy_hip_set_only_best_match(sassy_SearcherType *s, int on);
/* Searcher::with_max_overhang (src/search.rs:436-440): at most this many overhanging characters are
 * priced with alpha (the rest cost 1 each); negative = no limit. */
int sassy_hip_set_max_overhang(sassy_SearcherType *s, long max_overhang);
int sassy_hip_set_max_n_frac(sassy_SearcherType *s, float max_n_frac);

/* Searcher::search / search_all with full Match records (src/search.rs:510-525, 685-700). */
int sassy_hip_search(sassy_SearcherType *s, const uint8_t *pattern, size_t pattern_len,
                     const uint8_t *text, size_t text_len, size_t k, uint32_t flags,
                     sassy_hip_Result **out);

/* Searcher::search_with_fn (src/search.rs:767-784): keep only the end positions for which the
 * callback returns non-zero.  It sees what the reference's closure sees: the pattern, the text up
 * to the end position (text_till_end[0 .. end_pos)) and the strand (0 Fwd, 1 Rc); for the Rc strand
 * both are the complemented pattern and the REVERSED text the scan ran on.  Host text only. */
typedef int (*sassy_hip_end_filter)(const uint8_t *pattern, size_t pattern_len,
                                    const uint8_t *text_till_end, size_t end_pos, int strand,
                                    void *user);
int sassy_hip_search_with_fn(sassy_SearcherType *s, const uint8_t *pattern, size_t pattern_len,
                             const uint8_t *text, size_t text_len, size_t k, uint32_t flags,
                             sassy_hip_end_filter fn, void *user, sassy_hip_Result **out);

/* Searcher::search_many / search_patterns / search_texts (src/search.rs:531-678): every pattern in
 * every text; matches carry pattern_idx and text_idx and come pattern-major (pattern 0 in text 0,
 * pattern 0 in text 1, ...), each pair in `search` order (Fwd by end position, then Rc) -- the order
 * of the reference's SearchMode::Single.  With SASSY_HIP_TEXT_ON_DEVICE the text pointers are
 * device pointers (16-byte aligned). */
int sassy_hip_search_many(sassy_SearcherType *s, const uint8_t *const *patterns,
                          const size_t *pattern_lens, size_t n_patterns, const uint8_t *const *texts,
                          const size_t *text_lens, size_t n_texts, size_t k, uint32_t flags,
                          sassy_hip_Result **out);

/* One row of the reference CLI's match table (bin/grep.rs:465-470 header, :710-757 rows):
 *   pat_id  text_id  cost  strand  start  end  match_region  cigar
 * match_region = text[start..end), reverse-complemented for Rc matches unless `sam`; the cigar is
 * reversed for Rc matches if `sam`.  `m` is a match record, `cigar` its NUL-terminated cigar text
 * (sassy_hip_result_cigars(r) + m->cigar_off), `text` the host copy of the text it refers to.
 * Writes at most cap bytes (NUL-terminated, '\n'-terminated row) and returns the length the full
 * row needs (excluding the NUL), or a negative error code.  sassy_hip_tsv_header() is the header line. */
const char *sassy_hip_tsv_header(void);
long sassy_hip_format_tsv(const sassy_SearcherType *s, const sassy_hip_Match *m, const char *cigar,
                          const char *pat_id, const char *text_id, const uint8_t *text,
                          size_t text_len, int sam, char *buf, size_t cap);

/* One shard of a larger text that lives on this device (multi-GPU).
 * d_text points at the first byte of the halo; the shard owns global end positions whose
 * 64-byte block lies in [global_offset, global_offset + shard_len); halo_len bytes precede it
 * (halo_len = 0 for the first shard, otherwise >= sassy_hip_required_halo(m, k); halo_len,
 * global_offset and -- except for the last shard -- shard_len are multiples of 64).
 * total_len is the length of the whole text (the end-of-text rule is applied by the shard that
 * contains it).  Matches carry global coordinates.  Forward strand only. */
int sassy_hip_search_shard(sassy_SearcherType *s, const uint8_t *pattern, size_t pattern_len,
                           const uint8_t *d_text, uint64_t halo_len, uint64_t shard_len,
                           uint64_t global_offset, uint64_t total_len, size_t k, uint32_t flags,
                           sassy_hip_Result **out);
uint64_t sassy_hip_required_halo(size_t pattern_len, size_t k);

/* Searches in flight.  A stream of searches over a resident text (many patterns against one genome; the
 * reference's model is one Searcher per thread fed from a queue, bin/grep.rs:476-503) is pipelined on the
 * device: sassy_hip_search_shard_begin queues the whole kernel chain of one search (same arguments and rules
 * as sassy_hip_search_shard) and returns a ticket at once; sassy_hip_search_finish waits for that search and
 * hands out its result (out = NULL: wait and discard).  Up to 2 searches (sassy_hip_set_pipe_depth, at most 4) may
 * be in flight per searcher -- begin fails with SASSY_HIP_EINVAL beyond that --; they finish in any order the
 * caller likes, each result is exactly what sassy_hip_search_shard returns.  The short, latency-bound tail of
 * search i (chunk list, chunk DP, traceback) then runs underneath the bandwidth-bound prefilter of search
 * i+1.  The pattern is copied; the text must stay valid and unchanged until the ticket is finished.  Every
 * ticket must be finished before the searcher is freed (tickets still open then are dropped).  The calls of
 * one searcher must still come from one thread at a time.  sassy_hip_get_stats describes the search finished last.
 * While a ticket is open the searcher's synchronous entry points (sassy_hip_search, _search_shard, _search_many,
 * _search_encoded, _search_with_fn, the drop-in search) and sassy_hip_set_stream fail with SASSY_HIP_EINVAL: they
 * use the same streams and result buffers. */
/* The results of consecutive shards of one text (results[0] the leftmost), merged into one: matches in text order
 * with their cigars; a shard's conditional report (sassy_hip_result_conditional_index) is kept iff the plateau
 * state arriving from the shards on its left is TRUE (exit states, PASS = "ask further left").  incoming_state is the
 * state in front of results[0]: 1 (TRUE) when results[0] begins the text, 2 (PASS) when that is unknown -- then a
 * conditional report of the first shards stays conditional in the merged result, whose exit state and conditional
 * index make it a shard result again (merges nest).  This is what sassy_amd/multigpu.py does with gathered rows; a C
 * or Rust host that searched its shards with sassy_hip_search_shard on several devices calls it directly. */
int sassy_hip_merge_shards(const sassy_hip_Result *const *results, size_t n, int incoming_state, sassy_hip_Result **out);

/* One text over several devices inside one process (the reference's thread fan-out, bin/grep.rs:476-503, with a
 * GPU per thread): the text is cut into as many shards of whole 64-byte blocks as there are entries in `devices`
 * (NULL / 0: every visible device; a device may be named more than once -- several shards on one GPU), every shard
 * resident on its device with sassy_hip_required_halo(max_pattern_len, max_k) bytes of the text in front of it.  Each
 * device has a host thread of its own: sassy_hip_multi_set_text uploads all shards at once (one PCIe link per
 * device), sassy_hip_multi_search runs sassy_hip_search_shard on all devices at once and merges
 * (sassy_hip_merge_shards).  Forward strand, like the shard calls; flags: SASSY_HIP_ALL_MINIMA,
 * SASSY_HIP_WITHOUT_TRACE.  alpha = NAN (overhang needs the whole text in one buffer).  One call at a time per
 * multi-searcher. */
typedef struct sassy_hip_Multi sassy_hip_Multi;
sassy_hip_Multi *sassy_hip_multi_new(const char *alphabet, float alpha, const int *devices, size_t n_devices);
size_t sassy_hip_multi_shards(const sassy_hip_Multi *m);
int sassy_hip_multi_device(const sassy_hip_Multi *m, size_t shard);
sassy_SearcherType *sassy_hip_multi_searcher(sassy_hip_Multi *m, size_t shard); /* (for the setters; owned by m) */
int sassy_hip_multi_set_text(sassy_hip_Multi *m, const uint8_t *text, size_t len, size_t max_pattern_len, size_t max_k);
/* the synthetic text of sassy_hip_generate_dna / sassy_hip_plant, every device generating its own shard in place */
int sassy_hip_multi_generate_dna(sassy_hip_Multi *m, uint64_t len, uint64_t seed, size_t max_pattern_len, size_t max_k);
int sassy_hip_multi_plant(sassy_hip_Multi *m, uint64_t seed, const uint8_t *pattern, size_t pattern_len, size_t k,
                          uint64_t stride, uint64_t *planted);
int sassy_hip_multi_search(sassy_hip_Multi *m, const uint8_t *pattern, size_t pattern_len, size_t k, uint32_t flags,
                           sassy_hip_Result **out);
/* Both strands (reference: Searcher::new_rc; the CLI's default, bin/grep.rs:476-503 fans such searches out over its
 * threads): sassy_hip_multi_search then appends the Rc strand's matches to the forward ones, as sassy_hip_search does.
 * The Rc strand is complement(pattern) against the REVERSED text (src/search.rs:813-878): every device keeps a few
 * bytes of text on BOTH sides of its shard and searches its share of the reversed text as a shard of its own
 * (reversed with the reverse kernel, never uploaded twice); the shard results are chained in reversed order. */
int sassy_hip_multi_set_rc(sassy_hip_Multi *m, int rc);
/* search_encoded_patterns over several devices shards the PATTERNS (bin/crispr.rs:188-196): every device
 * scans the whole text for its share -- no halo, no seam, the same gather.  That needs the whole text on every device:
 * call sassy_hip_multi_set_replicated(m, 1) BEFORE the text is set / generated (0 = text shards again).  The result's
 * pattern_idx refers to the caller's list; order: device by device (the reference's order is an implementation
 * artefact too -- compare sorted).  Both strands with sassy_hip_multi_set_rc. */
int sassy_hip_multi_set_replicated(sassy_hip_Multi *m, int on);
int sassy_hip_multi_search_encoded(sassy_hip_Multi *m, const uint8_t *patterns, size_t n_patterns, size_t pattern_len,
                                   size_t k, uint32_t flags, sassy_hip_Result **out);
/* search_many over several devices shards the TEXTS (host pointers; whole texts, contiguous runs of about equal total
 * length per device; src/search.rs:531-603 does the same over threads): every device searches all patterns in its
 * texts, text_idx refers to the caller's list.  Needs no resident text. */
int sassy_hip_multi_search_many(sassy_hip_Multi *m, const uint8_t *const *patterns, const size_t *pattern_lens,
                                size_t n_patterns, const uint8_t *const *texts, const size_t *text_lens, size_t n_texts,
                                size_t k, uint32_t flags, sassy_hip_Result **out);
/* Searches in flight over several devices (the reference's worker threads never idle between two tasks,
 * bin/grep.rs:516-537, src/search.rs:531-603): begin() queues one shard search per device and strand
 * (sassy_hip_search_shard_begin on every device's host thread) and returns; finish() waits for that search on every
 * device and merges the shard results -- the tail of search i (chunk DP, tracebacks, the host's merge) runs under the
 * text stream of search i + 1 on every device.  Up to `depth` searches per multi-searcher (1 .. 4, default 3); tickets
 * may be finished in any order.  While a ticket is open every entry point that rewrites, re-lays-out or reallocates the
 * resident shards or uses the devices' searchers is refused with SASSY_HIP_EINVAL: sassy_hip_multi_set_text,
 * _generate_dna, _plant, _set_rc, _set_replicated, _search, _search_encoded and _search_many (the searches in flight
 * read those buffers).  Same result as sassy_hip_multi_search, both strands included (sassy_hip_multi_set_rc). */
typedef struct sassy_hip_MultiTicket sassy_hip_MultiTicket;
int sassy_hip_multi_set_pipe_depth(sassy_hip_Multi *m, int depth);
int sassy_hip_multi_search_begin(sassy_hip_Multi *m, const uint8_t *pattern, size_t pattern_len, size_t k,
                                 uint32_t flags, sassy_hip_MultiTicket **out);
int sassy_hip_multi_search_finish(sassy_hip_Multi *m, sassy_hip_MultiTicket *t, sassy_hip_Result **out);
/* The layout arithmetic of a multi-searcher without any device: part i's {offset, len, halo in front, bytes kept
 * behind, first / one-past-last forward byte of its share of the REVERSED text, that share's halo} in
 * out[7 i .. 7 i + 6] (out may be NULL).  Returns the number of parts that hold a share of the text (a text of less
 * than 64 n (n + 2) bytes is one device's), or -1 if a part's resident bytes would not cover its share of the
 * reversed text (never, by construction: what the tests pin). */
long sassy_hip_multi_layout(uint64_t len, size_t n_parts, size_t max_pattern_len, size_t max_k, uint64_t *out);
/* Where the seeded search of sassy_hip_search_encoded (many patterns, a long text) puts its seeds: k + 1 disjoint
 * pieces of the pattern's rows -- out_end[i] = one past the last row, out_len[i] = rows (<= 10) -- of at most two
 * lengths; with ambiguity letters in the patterns (alphabet "iupac") placed where the expected number of table
 * hits is smallest.  Host arithmetic, no device (tests).  Returns k + 1, or -1 for arguments out of range
 * (pattern_len 1 .. 64, k <= 7, k + 1 <= pattern_len). */
long sassy_hip_seed_layout(const char *alphabet, const uint8_t *const *patterns, size_t n_patterns, size_t pattern_len,
                           size_t k, uint32_t *out_end, uint32_t *out_len);
/* The 64 table rows of the sub-piece test that runs in front of the seeded search's verification, for seeds
 * (seed_end[i], seed_len[i]), i <= k, of a pattern of pattern_len <= 32 rows: out_rows[8 p + u] = 2a | (32 - 2 len) << 8 |
 * 2 (off & 15) << 16 | (off >> 4) << 24 for sub-piece u (rows [a, a + len)) of piece p, off = its leftmost shift in
 * characters from the start of the one text window the test reads (*out_win_left characters in front of the seed's
 * end); low byte 0xFF in out_rows[8 p]: no test for piece p.  Returns the largest off (<= 47), -1 for arguments out
 * of range.  Host arithmetic, no device (tests). */
long sassy_hip_seed_test_rows(size_t pattern_len, size_t k, const uint32_t *seed_end, const uint32_t *seed_len,
                              uint32_t *out_rows, uint32_t *out_win_left);
void sassy_hip_multi_free(sassy_hip_Multi *m);

typedef struct sassy_hip_Ticket sassy_hip_Ticket;
int sassy_hip_search_shard_begin(sassy_SearcherType *s, const uint8_t *pattern, size_t pattern_len,
                                 const uint8_t *d_text, uint64_t halo_len, uint64_t shard_len,
                                 uint64_t global_offset, uint64_t total_len, size_t k, uint32_t flags,
                                 sassy_hip_Ticket **out);
int sassy_hip_search_finish(sassy_SearcherType *s, sassy_hip_Ticket *ticket, sassy_hip_Result **out);
/* How many searches sassy_hip_search_shard_begin keeps in flight (1 .. 4, default 2 or SASSY_HIP_PIPE_DEPTH);
 * only while none is in flight. */
int sassy_hip_set_pipe_depth(sassy_SearcherType *s, int depth);

size_t sassy_hip_result_len(const sassy_hip_Result *r);
const sassy_hip_Match *sassy_hip_result_matches(const sassy_hip_Result *r);
const char *sassy_hip_result_cigars(const sassy_hip_Result *r); /* string pool */
size_t sassy_hip_result_cigars_len(const sassy_hip_Result *r);   /* bytes in the pool */
/* Wire format of the multi-GPU match gather (sassy_amd/multigpu.py): one row of 7 + cigar_bytes/8
 * int64 per match -- pattern_idx, text_start, text_end, pattern_start, pattern_end, cost, strand, then
 * cigar_bytes bytes of NUL-padded cigar text.  Pure host helper (no device needed); `cigar_bytes` is a
 * multiple of 8.  SASSY_HIP_EINVAL if a cigar string does not fit. */
int sassy_hip_pack_rows(const sassy_hip_Match *matches, size_t n, const char *cigars, size_t cigars_len,
                        int64_t *rows, size_t cigar_bytes);
/* Shard bookkeeping for the cross-shard plateau rule (see DESIGN.md "seams"):
 * entry_state: 0 = the shard's first report did not depend on the previous shard,
 *              1 = it did (the record with SASSY flag is still in the result, marked below);
 * exit_state:  0 = decreasing FALSE, 1 = decreasing TRUE, 2 = PASS (undetermined, inherit). */
int sassy_hip_result_exit_state(const sassy_hip_Result *r);
int64_t sassy_hip_result_conditional_index(const sassy_hip_Result *r); /* -1 if none */
void sassy_hip_result_free(sassy_hip_Result *r);

/* Searcher::encode_patterns / search_encoded_patterns (src/search.rs:404-423):
 * npat patterns of equal length plen (<= 64) stored back to back.
 * The many-pattern calls (this one and sassy_hip_search_many): the reports are the definition's (one left-to-right
 * pass per pattern and text) -- as in the reference, whose lanes hold whole texts / patterns there (no lane seams:
 * sassy_hip_set_reference_lanes has nothing to reproduce).  Limit: with an overhang
 * searcher (alpha) the one-pass kernels (seeded search, pattern-tiled scan) are not used: the patterns then run
 * one kernel chain each, correct but at the speed of single searches (the reference's v2 scans overhang in its
 * tiled loop, src/pattern_tiling/search.rs:222-323). */
sassy_hip_Encoded *sassy_hip_encode_patterns(sassy_SearcherType *s, const uint8_t *patterns,
                                             size_t npat, size_t plen);
void sassy_hip_encoded_free(sassy_hip_Encoded *e);
int sassy_hip_search_encoded(sassy_SearcherType *s, const sassy_hip_Encoded *e,
                             const uint8_t *text, size_t text_len, size_t k, uint32_t flags,
                             sassy_hip_Result **out);

/* Synthetic inputs generated in place on the device; same function as
 * oracle/sassy_oracle.c:orc_generate_dna / orc_plant_window, checked byte for byte in tests. */
int sassy_hip_generate_dna(uint8_t *d_text, uint64_t n, uint64_t seed, uint64_t first,
                           void *hip_stream);
/* A repeat-rich synthetic text (measurement and test input, like sassy_hip_generate_dna): 4 KiB regions of
 * i.i.d. ACGT (the majority), microsatellites (6 %), copies of 4 interspersed-repeat families with 8 %
 * divergence (10 %), soft-masked stretches (1 %) and, with with_n != 0, runs of 'N' (2 %); every byte is a pure
 * function of (seed, first + index).  See sassy_amd/csrc/aux_kernels.hip: genome_like_byte. */
int sassy_hip_generate_genome_like(uint8_t *d_text, uint64_t n, uint64_t seed, uint64_t first, int with_n,
                                   void *hip_stream);
int sassy_hip_plant(uint8_t *d_text, uint64_t n, uint64_t first, uint64_t total_n, uint64_t seed,
                    const uint8_t *pattern, size_t pattern_len, size_t k, uint64_t stride,
                    void *hip_stream, uint64_t *planted);
/* ... with the plants `phase` bytes further on (q * stride + stride / 2 + phase): several patterns in one text. */
int sassy_hip_plant_phase(uint8_t *d_text, uint64_t n, uint64_t first, uint64_t total_n, uint64_t seed,
                          const uint8_t *pattern, size_t pattern_len, size_t k, uint64_t stride, uint64_t phase,
                          void *hip_stream, uint64_t *planted);

/* Plain device memory helpers so that non-torch callers (C, tests) can use the device paths. */
void *sassy_hip_malloc(size_t bytes);
void sassy_hip_free(void *d_ptr);
int sassy_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int sassy_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SASSY_HIP_H */
