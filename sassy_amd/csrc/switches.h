// switches.h -- every path-forcing / tuning switch of libsassy_hip.so, in ONE table.
//
// A searcher reads the table once, when it is created (sassy_hip_searcher_new, the workers of sassy_hip_multi_new): every
// entry takes its default, then the value of the environment variable SASSY_HIP_<NAME> if that is set; afterwards
// sassy_hip_set_option(searcher, "<name>", value) changes one entry of that searcher.  No search entry point reads the
// environment.  The switches exist for the parity tests (tests/test_gpu_parity.py: every entry is forced by a test named
// in DESIGN.md 5.7) and for the timing tools under tools/; a production caller needs none of them.
#pragma once
#include <cstdint>
#include <string>

namespace sassy_hip {

// X(field, default, "what it does")
#define SASSY_HIP_SWITCHES(X)                                                                                               \
  X(prefilter, -1, "-1: a prefilter where its pieces are selective; 0: never (streaming DP over every block); 1: also with 2-row pieces") \
  X(fused, 1, "the bit-plane filter runs the chunk DP of what it finds in the same launch (0: bitmap -> chunk list -> list kernel)") \
  X(ref_lanes, 0, "4 / 8: reproduce the reference binary's 4- / 8-lane reports on a single text (0: the one-pass definition)") \
  X(pipe_depth, 2, "searches in flight per searcher (1 .. 4) for search_shard_begin / _finish")                          \
  X(tune, 0, "1: the on-line geometry tuner tries neighbouring lane-chunk lengths during a resident text's first searches") \
  X(timing, 1, "HIP-event timing: 0 none, 1 the dominant kernel, 2 every phase")                                            \
  X(ctl_twin, 1, "a search clears the lane's other control block behind its last kernel, the next search starts without a memset launch (0: a memset in front of every search)") \
  X(row_cut, 1, "0: the DP kernels compute every pattern row of every block (no wave-voted stop)")                         \
  X(stage_blocks, 0, "streaming DP: text blocks per lane and staging step (1 / 2; 0 = default 1)")                         \
  X(filter_kind, 0, "force a prefilter kernel where it applies: 1 slot masks, 2 bit planes, 3 q-gram table, 4 q-gram counting") \
  X(self_rank, 1, "0: separate rank kernels instead of the traceback waves ranking their reports themselves")               \
  X(filter_linear, 0, "> 0: filter_dna_linear_kernel with the text cut into this many wave ranges")                         \
  X(trace_wave, 1, "0: thread-per-report traceback only")                                                                   \
  X(iupac_planes, 1, "0: Iupac searchers with plain patterns take the Iupac chain, never the bit-plane launch with the text check") \
  X(short_pieces, 1, "0: no fused launch for shapes whose pigeonhole pieces are 6 rows")                                    \
  X(pair, 1, "0: no paired filter; 2: the counting filter keeps the shapes it is selective for")                            \
  X(pair_rc, 1, "0: both strands of a paired-filter shape through the forward strand's streaming DP with Rc marks")         \
  X(count_stage_blocks, 0, "counting filter: 1 = half lines per staging step (default: whole lines)")                       \
  X(count_wpg, 0, "counting filter: waves per workgroup, 4 or 16 (0: by LDS fit)")                                          \
  X(count_fused, 1, "the counting filter hands its candidate blocks to the chunk DP without bitmap and chunk-list launch (0: classic chain)") \
  X(fused_probe, 0, "1: the fused launch reports where its waves spend their time; 2: no chunk DP at all (timing only, no reports)") \
  X(fused_press, 0, "> 0: a wave of the fused launch runs a chunk-DP pass once this many windows are queued")              \
  X(ext_events, 1, "0: timing events around the fused launch instead of carried by the dispatch")                           \
  X(list_words, 1, "multi-word chunk DP of few chunks: 1 a lane per block (list_rows_kernel), 2 a lane per pattern word, 0 the lane-per-chunk kernel only")                                              \
  X(trace_threads, 0, ">= 64: threads of the thread-per-report traceback launch")                                           \
  X(trace_probe, 0, "1: the traceback waves report microseconds per phase to stderr")                                       \
  X(big_pin, 1, "0: dense results through the host's vectors instead of one pinned block")                                  \
  X(compact_cigars, 1, "0: dense results keep their cigar slots' padding")                                                  \
  X(adopt, 1, "0: results are copied out of the pinned block the kernels wrote them into")                                  \
  X(lanes, 1, "2 .. 4: one search cut into sub-shards on that many streams")                                                \
  X(subshard_min, 128 << 20, "smallest sub-shard in bytes for lanes > 1")                                                   \
  X(rc_fused, 1, "0: the Rc strand from a reversed copy instead of Rc marks made by the forward pass")                      \
  X(strands_in_flight, 1, "0: two strands that are two searches run one after the other")                                   \
  X(encoded_trace_threads, -1, "search_encoded: threads of the dense traceback launch (-1: by result size)")                \
  X(encoded_pin, 1, "0: search_encoded's dense results through the host's vectors")                                         \
  X(seed_layout, 1, "0: the seeded search cuts its seeds evenly instead of by expected table hits")                         \
  X(seed_subtest, 1, "0: seeded search without the sub-piece test")                                                         \
  X(seed_narrow, 1, "0: never the narrow (4-dword) sub-piece test layout")                                                  \
  X(seed_pos64, 0, "1: force the layout for positions beyond 32 bits")                                                      \
  X(many_tiled, -1, "search_many: 1 force / 0 forbid the one-pass paths (-1: by size)")                                     \
  X(many_seeded, -1, "search_many: 1 force / 0 forbid the seeded search inside the one-pass path")                          \
  X(many_assemble, 1, "0: search_many's records ordered by the host")                                                       \
  X(overhang_tiled, 1, "0: overhang with many patterns as one launch per pattern and strand")                               \
  X(overhang_seeded, 1, "0: overhang with many patterns through the per-text tiled scan over everything")                   \
  X(multi_min_text, 16 << 20, "search_encoded: smallest text (bytes) for the multi-pattern prefilter")                       \
  X(tiled, -1, "search_encoded: 1 force / 0 forbid the pattern-tiled scan")                                                 \
  X(seeded, -1, "search_encoded: 1 force / 0 forbid the seeded search")

struct Switches {
#define SASSY_HIP_SWITCH_FIELD(name, dflt, doc) long name = (dflt);
  SASSY_HIP_SWITCHES(SASSY_HIP_SWITCH_FIELD)
#undef SASSY_HIP_SWITCH_FIELD
  // SASSY_HIP_DEVICES: the devices the drop-in search() of include/sassy.h fans a host text over ("all", or a list)
  std::string devices;
};

// defaults, then the environment (SASSY_HIP_<NAME in capitals>); the library's only reader of the environment
Switches load_switches();
// one entry by name (lower case, without the prefix); false: no such switch
bool set_switch(Switches& sw, const char* name, long value);
bool get_switch(const Switches& sw, const char* name, long* value);
// "name<TAB>default<TAB>what it does\n" per switch
const char* switch_table();

}  // namespace sassy_hip
