"""sassy_amd -- MI355X-native drop-in for sassy's bit-parallel search path.

The product is ``sassy_amd/lib/libsassy_hip.so`` (hand-written HIP for gfx950 behind the C-ABI of
``include/sassy.h`` / ``include/sassy_hip.h``).  This package is the thin Python host mirror of
the reference's Python / Rust searcher interface for that path:

    reference (src/python.rs:26-220)            here
    sassy.Searcher(alphabet, rc, alpha)     ->  sassy_amd.Searcher(alphabet, rc, alpha)
    .search(pattern, text, k)               ->  .search(pattern, text, k)
    .search_all(pattern, text, k)           ->  .search_all(pattern, text, k)
    Searcher::encode_patterns / search_encoded_patterns (src/search.rs:404-423)
                                            ->  .encode_patterns(...) / .search_encoded_patterns(...)
    Match getters pattern_idx, text_start, text_end, pattern_start, pattern_end, cost,
    strand ('+'/'-'), cigar                 ->  the same attribute names

There is no CPU implementation in here: without the compiled library or without a HIP device
every search raises ``SassyHipError``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from collections.abc import Sequence as _SequenceABC
from typing import List, Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libsassy_hip.so")
if os.environ.get("SASSY_HIP_LIBRARY"):  # A/B timing of two builds on one box (tools only)
    _SO = os.environ["SASSY_HIP_LIBRARY"]

ALL_MINIMA = 1
WITHOUT_TRACE = 2
TEXT_ON_DEVICE = 4
TEXT_UNCHANGED = 8
UINT64_MAX = (1 << 64) - 1


class SassyHipError(RuntimeError):
    pass


class CMatch(C.Structure):
    """include/sassy.h: sassy_Match (40 bytes, align 8)."""
    _fields_ = [
        ("text_start", C.c_size_t),
        ("text_end", C.c_size_t),
        ("pattern_start", C.c_size_t),
        ("pattern_end", C.c_size_t),
        ("cost", C.c_int32),
        ("strand", C.c_uint8),
    ]


class _HipMatch(C.Structure):
    _fields_ = [
        ("pattern_idx", C.c_uint64),
        ("text_idx", C.c_uint64),
        ("text_start", C.c_uint64),
        ("text_end", C.c_uint64),
        ("pattern_start", C.c_uint64),
        ("pattern_end", C.c_uint64),
        ("cost", C.c_int32),
        ("strand", C.c_uint8),
        ("pad_", C.c_uint8 * 3),
        ("cigar_off", C.c_uint32),
        ("cigar_len", C.c_uint32),
    ]


# int fn(pattern, pattern_len, text_till_end, end_pos, strand, user)  (include/sassy_hip.h)
END_FILTER = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, C.c_int,
                         C.c_void_p)


class Stats(C.Structure):
    _fields_ = [
        ("scan_ms", C.c_double),
        ("trace_ms", C.c_double),
        ("total_ms", C.c_double),
        ("text_bytes", C.c_uint64),
        ("scan_launches", C.c_uint64),
        ("candidates", C.c_uint64),
        ("cond_resolved", C.c_uint64),
        ("chunks", C.c_uint64),
        ("blocks", C.c_uint64),
        ("word_rows", C.c_uint64),
        ("blocks_per_chunk", C.c_uint32),
        ("warmup_blocks", C.c_uint32),
        ("grid", C.c_uint32),
        ("filtered", C.c_uint32),
        ("filter_ms", C.c_double),
        ("hit_blocks", C.c_uint64),
        ("piece_len", C.c_uint32),
        ("fused", C.c_uint32),
        ("host_enqueue_ms", C.c_double),
        ("host_wait_ms", C.c_double),
        ("host_post_ms", C.c_double),
        ("live_blocks", C.c_uint64),
        ("pair", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_ if f != "pad_"}


def option_table():
    """[(name, default, what it does)] of every switch of the library (csrc/switches.h)."""
    rows = []
    for line in lib().sassy_hip_option_table().decode().splitlines():
        name, dflt, doc = line.split("\t", 2)
        rows.append((name, dflt, doc))
    return rows


# every symbol include/sassy.h and include/sassy_hip.h declare
EXPORTED_SYMBOLS = [
    "sassy_searcher", "sassy_searcher_free", "search", "sassy_matches_free",
    "sassy_hip_last_error", "sassy_hip_version", "sassy_hip_device_count",
    "sassy_hip_searcher_new", "sassy_hip_set_stream", "sassy_hip_get_stats",
    "sassy_hip_search", "sassy_hip_search_shard", "sassy_hip_required_halo",
    "sassy_hip_search_shard_begin", "sassy_hip_search_finish", "sassy_hip_set_pipe_depth", "sassy_hip_set_geometry_tuner", "sassy_hip_set_reference_lanes",
    "sassy_hip_result_len", "sassy_hip_result_matches", "sassy_hip_result_cigars",
    "sassy_hip_result_cigars_len", "sassy_hip_pack_rows", "sassy_hip_enable_counters", "sassy_hip_set_timing",
    "sassy_hip_set_only_best_match", "sassy_hip_set_max_n_frac", "sassy_hip_search_with_fn",
    "sassy_hip_set_max_overhang", "sassy_hip_set_prefilter", "sassy_hip_set_fused",
    "sassy_hip_set_option", "sassy_hip_get_option", "sassy_hip_option_table", "sassy_hip_plant_phase",
    "sassy_hip_set_device", "sassy_hip_get_device", "sassy_hip_merge_shards",
    "sassy_hip_multi_new", "sassy_hip_multi_shards", "sassy_hip_multi_device", "sassy_hip_multi_searcher",
    "sassy_hip_multi_set_text", "sassy_hip_multi_generate_dna", "sassy_hip_multi_plant", "sassy_hip_multi_search",
    "sassy_hip_multi_free",
    "sassy_hip_search_many", "sassy_hip_tsv_header", "sassy_hip_format_tsv",
    "sassy_hip_result_exit_state", "sassy_hip_result_conditional_index", "sassy_hip_result_free",
    "sassy_hip_encode_patterns", "sassy_hip_encoded_free", "sassy_hip_search_encoded",
    "sassy_hip_multi_set_rc", "sassy_hip_multi_set_replicated", "sassy_hip_multi_search_encoded", "sassy_hip_multi_search_many",
    "sassy_hip_multi_set_pipe_depth", "sassy_hip_multi_search_begin", "sassy_hip_multi_search_finish", "sassy_hip_multi_layout", "sassy_hip_seed_layout", "sassy_hip_seed_test_rows",
    "sassy_hip_generate_dna", "sassy_hip_generate_genome_like", "sassy_hip_plant",
    "sassy_hip_malloc", "sassy_hip_free", "sassy_hip_memcpy_h2d", "sassy_hip_memcpy_d2h",
]

_lib = None


def library_path() -> str:
    return _SO


def lib():
    """The loaded C-ABI library.  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise SassyHipError(
            f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  There is no pure-Python / CPU fallback for the search path.")
    L = C.CDLL(_SO)
    vp, u8p, sz = C.c_void_p, C.c_char_p, C.c_size_t
    L.sassy_hip_last_error.restype = C.c_char_p
    L.sassy_hip_version.restype = C.c_char_p
    L.sassy_hip_device_count.restype = C.c_int
    L.sassy_hip_searcher_new.restype = vp
    L.sassy_hip_searcher_new.argtypes = [C.c_char_p, C.c_bool, C.c_float]
    L.sassy_searcher.restype = vp
    L.sassy_searcher.argtypes = [C.c_char_p, C.c_bool, C.c_float]
    L.sassy_searcher_free.restype = None
    L.sassy_searcher_free.argtypes = [vp]
    L.search.restype = sz
    L.search.argtypes = [vp, u8p, sz, u8p, sz, sz, C.POINTER(C.POINTER(CMatch))]
    L.sassy_matches_free.restype = None
    L.sassy_matches_free.argtypes = [C.POINTER(CMatch), sz]
    L.sassy_hip_set_stream.restype = C.c_int
    L.sassy_hip_set_stream.argtypes = [vp, vp]
    L.sassy_hip_get_stats.restype = C.c_int
    L.sassy_hip_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.sassy_hip_enable_counters.restype = C.c_int
    L.sassy_hip_enable_counters.argtypes = [vp, C.c_int]
    L.sassy_hip_set_timing.restype = C.c_int
    L.sassy_hip_set_timing.argtypes = [vp, C.c_int]
    L.sassy_hip_set_prefilter.restype = C.c_int
    L.sassy_hip_set_prefilter.argtypes = [vp, C.c_int]
    L.sassy_hip_set_fused.restype = C.c_int
    L.sassy_hip_set_fused.argtypes = [vp, C.c_int]
    L.sassy_hip_set_option.restype = C.c_int
    L.sassy_hip_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.sassy_hip_get_option.restype = C.c_int
    L.sassy_hip_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long)]
    L.sassy_hip_option_table.restype = C.c_char_p
    L.sassy_hip_option_table.argtypes = []
    L.sassy_hip_set_device.restype = C.c_int
    L.sassy_hip_set_device.argtypes = [vp, C.c_int]
    L.sassy_hip_get_device.restype = C.c_int
    L.sassy_hip_get_device.argtypes = [vp]
    L.sassy_hip_merge_shards.restype = C.c_int
    L.sassy_hip_merge_shards.argtypes = [C.POINTER(vp), C.c_size_t, C.c_int, C.POINTER(vp)]
    L.sassy_hip_multi_new.restype = vp
    L.sassy_hip_multi_new.argtypes = [C.c_char_p, C.c_float, C.POINTER(C.c_int), C.c_size_t]
    L.sassy_hip_multi_shards.restype = C.c_size_t
    L.sassy_hip_multi_shards.argtypes = [vp]
    L.sassy_hip_multi_device.restype = C.c_int
    L.sassy_hip_multi_device.argtypes = [vp, C.c_size_t]
    L.sassy_hip_multi_searcher.restype = vp
    L.sassy_hip_multi_searcher.argtypes = [vp, C.c_size_t]
    L.sassy_hip_multi_set_text.restype = C.c_int
    L.sassy_hip_multi_set_text.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]
    L.sassy_hip_multi_generate_dna.restype = C.c_int
    L.sassy_hip_multi_generate_dna.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t]
    L.sassy_hip_multi_plant.restype = C.c_int
    L.sassy_hip_multi_plant.argtypes = [vp, C.c_uint64, C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sassy_hip_multi_search.restype = C.c_int
    L.sassy_hip_multi_search.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(vp)]
    if hasattr(L, "sassy_hip_multi_set_rc"):  # (an older build loaded for an A/B timing lacks the newer entry points)
        L.sassy_hip_multi_set_rc.restype = C.c_int
        L.sassy_hip_multi_set_rc.argtypes = [vp, C.c_int]
        L.sassy_hip_multi_set_replicated.restype = C.c_int
        L.sassy_hip_multi_set_replicated.argtypes = [vp, C.c_int]
        L.sassy_hip_multi_search_encoded.restype = C.c_int
        L.sassy_hip_multi_search_encoded.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(vp)]
        L.sassy_hip_multi_search_many.restype = C.c_int
        L.sassy_hip_multi_search_many.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t,
                                                  C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_uint32,
                                                  C.POINTER(vp)]
    if hasattr(L, "sassy_hip_multi_search_begin"):
        L.sassy_hip_multi_set_pipe_depth.restype = C.c_int
        L.sassy_hip_multi_set_pipe_depth.argtypes = [vp, C.c_int]
        L.sassy_hip_multi_search_begin.restype = C.c_int
        L.sassy_hip_multi_search_begin.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(vp)]
        L.sassy_hip_multi_search_finish.restype = C.c_int
        L.sassy_hip_multi_search_finish.argtypes = [vp, vp, C.POINTER(vp)]
        L.sassy_hip_multi_layout.restype = C.c_long
        L.sassy_hip_multi_layout.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64)]
    if hasattr(L, "sassy_hip_seed_layout"):
        L.sassy_hip_seed_layout.restype = C.c_long
        L.sassy_hip_seed_layout.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if hasattr(L, "sassy_hip_seed_test_rows"):
        L.sassy_hip_seed_test_rows.restype = C.c_long
        L.sassy_hip_seed_test_rows.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                               C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.sassy_hip_multi_free.restype = None
    L.sassy_hip_multi_free.argtypes = [vp]
    L.sassy_hip_set_only_best_match.restype = C.c_int
    L.sassy_hip_set_only_best_match.argtypes = [vp, C.c_int]
    L.sassy_hip_set_max_overhang.restype = C.c_int
    L.sassy_hip_set_max_overhang.argtypes = [vp, C.c_long]
    L.sassy_hip_set_max_n_frac.restype = C.c_int
    L.sassy_hip_set_max_n_frac.argtypes = [vp, C.c_float]
    L.sassy_hip_search_many.restype = C.c_int
    L.sassy_hip_search_many.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t,
                                        C.c_uint32, C.POINTER(vp)]
    L.sassy_hip_tsv_header.restype = C.c_char_p
    L.sassy_hip_tsv_header.argtypes = []
    L.sassy_hip_format_tsv.restype = C.c_long
    L.sassy_hip_format_tsv.argtypes = [vp, C.POINTER(_HipMatch), C.c_char_p, C.c_char_p, C.c_char_p, vp, C.c_size_t,
                                       C.c_int, C.c_char_p, C.c_size_t]
    L.sassy_hip_search_with_fn.restype = C.c_int
    L.sassy_hip_search_with_fn.argtypes = [vp, C.c_char_p, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_uint32,
                                           END_FILTER, vp, C.POINTER(vp)]
    L.sassy_hip_search.restype = C.c_int
    L.sassy_hip_search.argtypes = [vp, u8p, sz, vp, sz, sz, C.c_uint32, C.POINTER(vp)]
    L.sassy_hip_search_shard.restype = C.c_int
    L.sassy_hip_search_shard.argtypes = [vp, u8p, sz, vp, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_uint64, sz, C.c_uint32, C.POINTER(vp)]
    L.sassy_hip_search_shard_begin.restype = C.c_int
    L.sassy_hip_search_shard_begin.argtypes = [vp, u8p, sz, vp, C.c_uint64, C.c_uint64, C.c_uint64,
                                               C.c_uint64, sz, C.c_uint32, C.POINTER(vp)]
    L.sassy_hip_set_reference_lanes.restype = C.c_int
    L.sassy_hip_set_reference_lanes.argtypes = [vp, C.c_int]
    L.sassy_hip_set_geometry_tuner.restype = C.c_int
    L.sassy_hip_set_geometry_tuner.argtypes = [vp, C.c_int]
    L.sassy_hip_set_pipe_depth.restype = C.c_int
    L.sassy_hip_set_pipe_depth.argtypes = [vp, C.c_int]
    L.sassy_hip_search_finish.restype = C.c_int
    L.sassy_hip_search_finish.argtypes = [vp, vp, C.POINTER(vp)]
    L.sassy_hip_required_halo.restype = C.c_uint64
    L.sassy_hip_required_halo.argtypes = [sz, sz]
    L.sassy_hip_result_len.restype = sz
    L.sassy_hip_result_len.argtypes = [vp]
    L.sassy_hip_result_matches.restype = C.POINTER(_HipMatch)
    L.sassy_hip_result_matches.argtypes = [vp]
    L.sassy_hip_result_cigars.restype = vp
    L.sassy_hip_result_cigars.argtypes = [vp]
    L.sassy_hip_result_cigars_len.restype = sz
    L.sassy_hip_result_cigars_len.argtypes = [vp]
    L.sassy_hip_pack_rows.restype = C.c_int
    L.sassy_hip_pack_rows.argtypes = [vp, sz, C.c_char_p, sz, vp, sz]
    L.sassy_hip_result_exit_state.restype = C.c_int
    L.sassy_hip_result_exit_state.argtypes = [vp]
    L.sassy_hip_result_conditional_index.restype = C.c_int64
    L.sassy_hip_result_conditional_index.argtypes = [vp]
    L.sassy_hip_result_free.restype = None
    L.sassy_hip_result_free.argtypes = [vp]
    L.sassy_hip_encode_patterns.restype = vp
    L.sassy_hip_encode_patterns.argtypes = [vp, u8p, sz, sz]
    L.sassy_hip_encoded_free.restype = None
    L.sassy_hip_encoded_free.argtypes = [vp]
    L.sassy_hip_search_encoded.restype = C.c_int
    L.sassy_hip_search_encoded.argtypes = [vp, vp, vp, sz, sz, C.c_uint32, C.POINTER(vp)]
    L.sassy_hip_generate_dna.restype = C.c_int
    L.sassy_hip_generate_dna.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    L.sassy_hip_generate_genome_like.restype = C.c_int
    L.sassy_hip_generate_genome_like.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, vp]
    L.sassy_hip_plant.restype = C.c_int
    L.sassy_hip_plant.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u8p, sz, sz,
                                  C.c_uint64, vp, C.POINTER(C.c_uint64)]
    L.sassy_hip_plant_phase.restype = C.c_int
    L.sassy_hip_plant_phase.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u8p, sz, sz,
                                        C.c_uint64, C.c_uint64, vp, C.POINTER(C.c_uint64)]
    L.sassy_hip_malloc.restype = vp
    L.sassy_hip_malloc.argtypes = [sz]
    L.sassy_hip_free.restype = None
    L.sassy_hip_free.argtypes = [vp]
    L.sassy_hip_memcpy_h2d.restype = C.c_int
    L.sassy_hip_memcpy_h2d.argtypes = [vp, vp, sz]
    L.sassy_hip_memcpy_d2h.restype = C.c_int
    L.sassy_hip_memcpy_d2h.argtypes = [vp, vp, sz]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise SassyHipError(f"libsassy_hip error {rc}: {lib().sassy_hip_last_error().decode()}")


def device_count() -> int:
    return lib().sassy_hip_device_count()


@dataclass(frozen=True)
class Match:
    """Reference Match (src/search.rs:35-62) with the Python getters' conventions
    (src/python.rs:157-203): strand is '+' or '-', cigar is the SAM string."""
    pattern_idx: int
    text_start: int
    text_end: int
    pattern_start: int
    pattern_end: int
    cost: int
    strand: str
    cigar: str
    text_idx: int = 0

    def sort_key(self):
        return (self.pattern_idx, self.text_start, self.text_end, self.cost, self.strand, self.cigar)


def match_dtype():
    """numpy view of include/sassy_hip.h: sassy_hip_Match (64 bytes)."""
    import numpy as np
    return np.dtype([("pattern_idx", "<u8"), ("text_idx", "<u8"), ("text_start", "<u8"),
                     ("text_end", "<u8"), ("pattern_start", "<u8"), ("pattern_end", "<u8"),
                     ("cost", "<i4"), ("strand", "u1"), ("pad_", "u1", (3,)),
                     ("cigar_off", "<u4"), ("cigar_len", "<u4")])


def _bytes_at(addr, size: int) -> bytes:
    """`size` bytes at a C address (ctypes.string_at takes an int-sized length: results above 2 GiB need the array form)."""
    if not size:
        return b""
    if size < (1 << 31):
        return C.string_at(addr, size)
    return bytes((C.c_char * size).from_address(addr if isinstance(addr, int) else C.cast(addr, C.c_void_p).value))


class Result:
    """Matches of one call plus the shard bookkeeping (see include/sassy_hip.h).

    ``array`` is the C match array copied once into a numpy structured array (match_dtype) and
    ``pool`` the cigar string pool; ``matches`` materialises reference-style Match objects on
    first use (a Python object per match is far slower than the search itself)."""

    def __init__(self, handle):
        # The C call has left the finished records on the host (sassy_hip_Result); they are copied into
        # Python objects only when somebody looks at them (`array`, `pool`, `matches`).
        L = lib()
        self._h = handle
        self._n = L.sassy_hip_result_len(handle)
        self.exit_state = L.sassy_hip_result_exit_state(handle)
        self.conditional_index = L.sassy_hip_result_conditional_index(handle)
        self._array = None
        self._pool = None
        self._matches = None

    def _materialise(self):
        if self._array is not None:
            return
        import numpy as np
        L = lib()
        h, self._h = self._h, None
        try:
            n = self._n
            ptr = L.sassy_hip_result_matches(h)
            raw = _bytes_at(ptr, n * 64)
            self._array = np.frombuffer(raw, dtype=match_dtype())
            plen = L.sassy_hip_result_cigars_len(h)
            pool = L.sassy_hip_result_cigars(h)
            self._pool = _bytes_at(pool, plen)
        finally:
            L.sassy_hip_result_free(h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().sassy_hip_result_free(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    @property
    def array(self):
        """The C match array copied once into a numpy structured array (match_dtype)."""
        self._materialise()
        return self._array

    @property
    def pool(self) -> bytes:
        """The cigar string pool."""
        self._materialise()
        return self._pool

    def __len__(self):
        return self._n

    @property
    def matches(self) -> List["Match"]:
        """The records as reference-style Match objects -- always a real list, whatever the size of the result (a Python
        object per match costs 1.4 us: for results of 10^5 records and more use `.lazy_matches` or the numpy `.array`)."""
        if self._matches is None:
            self._matches = matches_from_array(self.array, self.pool)
        return self._matches

    @property
    def lazy_matches(self) -> "MatchList":
        """A read-only sequence over the numpy array and the cigar pool: Match objects are made when they are looked at."""
        return MatchList(self.array, self.pool)


def _bytes_payload_offset():
    """Where the payload of a bytes object sits behind id(obj) -- CPython's object layout, checked once at import against a
    known string; None (search_many then takes every text through _ptr_len) on any other interpreter or layout, or where
    a pointer does not fit the uint64 / size_t arrays the fast path builds."""
    import sys
    if sys.implementation.name != "cpython" or C.sizeof(C.c_void_p) != 8 or C.sizeof(C.c_size_t) != 8:
        return None
    off = bytes.__basicsize__ - 1
    probe = b"sassy-layout-probe"
    try:
        return off if C.string_at(id(probe) + off, len(probe)) == probe else None
    except Exception:
        return None


_BYTES_PAYLOAD_OFFSET = _bytes_payload_offset()


class MatchList(_SequenceABC):
    """A read-only sequence of Match objects over a Result's numpy array and cigar pool: len(), indexing, slicing and
    iteration behave like the list they replace, the objects are made when they are looked at.  `.array` / `.pool`:
    the columns themselves (match_dtype)."""

    __slots__ = ("array", "pool")

    def __init__(self, array, pool: bytes):
        self.array = array
        self.pool = pool

    def __len__(self):
        return len(self.array)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return matches_from_array(self.array[i], self.pool)
        return matches_from_array(self.array[i:i + 1] if i >= 0 else self.array[len(self.array) + i:len(self.array) + i + 1],
                                  self.pool)[0]

    def __iter__(self):
        for a in range(0, len(self.array), 4096):
            yield from matches_from_array(self.array[a:a + 4096], self.pool)

    def __eq__(self, other):
        if isinstance(other, (list, tuple, MatchList)):
            return len(other) == len(self) and all(x == y for x, y in zip(self, other))
        return NotImplemented

    def __repr__(self):
        return f"MatchList({len(self)} matches)"


def matches_from_array(array, pool: bytes) -> List["Match"]:
    out = []
    for r in array.tolist():
        (pidx, tidx, ts, te, ps, pe, cost, strand, _pad, coff, clen) = r
        out.append(Match(pidx, ts, te, ps, pe, cost, "-" if strand else "+",
                         pool[coff:coff + clen].decode() if clen else "", tidx))
    return out


def _ptr_len(text):
    """(address, length, keepalive, on_device) for bytes / numpy uint8 / torch uint8 tensors."""
    if isinstance(text, (bytes, bytearray)):
        b = bytes(text)
        return C.cast(C.c_char_p(b), C.c_void_p).value or 0, len(b), b, False
    if hasattr(text, "data_ptr"):  # torch tensor
        if text.dtype.itemsize != 1 or not text.is_contiguous():
            raise SassyHipError("text tensor must be contiguous uint8")
        return text.data_ptr(), text.numel(), text, text.is_cuda
    if hasattr(text, "ctypes"):  # numpy
        import numpy as np
        a = np.ascontiguousarray(text, dtype=np.uint8)
        return a.ctypes.data, a.size, a, False
    raise TypeError("text must be bytes, a numpy uint8 array or a torch uint8 tensor")


class TextBatch:
    """Many host texts as ONE buffer and two arrays: text i = buffer[starts[i] : starts[i] + lens[i]].  What a FASTA /
    FASTQ reader holds anyway; `search_many` takes it without touching a Python object per text (a list of 330 000
    `bytes` costs 25 ms to marshal -- more than the search).  `TextBatch.from_list(texts)` packs a list once, for
    several calls."""

    def __init__(self, buffer, starts, lens):
        import numpy as np
        self.buffer = np.ascontiguousarray(np.frombuffer(buffer, dtype=np.uint8) if isinstance(buffer, (bytes, bytearray, memoryview))
                                           else buffer, dtype=np.uint8)
        self.starts = np.ascontiguousarray(starts, dtype=np.uint64)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint64)
        if self.starts.shape != self.lens.shape or self.starts.ndim != 1:
            raise SassyHipError("TextBatch: starts and lens must be one-dimensional and of one length")
        if len(self.starts) and int((self.starts + self.lens).max()) > self.buffer.size:
            raise SassyHipError("TextBatch: a text ends behind the buffer")
        self._addr = self.starts + np.uint64(self.buffer.ctypes.data)  # (the buffer is kept alive by self)

    @classmethod
    def from_list(cls, texts: Sequence[bytes]) -> "TextBatch":
        import numpy as np
        lens = np.fromiter(map(len, texts), dtype=np.uint64, count=len(texts))
        starts = np.zeros(len(texts), dtype=np.uint64)
        if len(texts) > 1:
            np.cumsum(lens[:-1], out=starts[1:])
        return cls(b"".join(texts), starts, lens)

    def __len__(self):
        return len(self.starts)


class EncodedPatterns:
    """Reference EncodedPatterns (src/pattern_tiling/general.rs:132-150), opaque."""

    def __init__(self, handle, n, plen):
        self._h, self.n_patterns, self.pattern_len = handle, n, plen

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sassy_hip_encoded_free(self._h)
            self._h = None


class Searcher:
    """Mirror of sassy.Searcher (src/python.rs:26-64) / Searcher::<P>::new (src/search.rs:486-503).

    Note the reference's Python default rc=True; ``new_fwd``-style searchers pass rc=False."""

    def __init__(self, alphabet: str, rc: bool = True, alpha: Optional[float] = None):
        L = lib()
        a = float("nan") if alpha is None else float(alpha)
        self._h = L.sassy_hip_searcher_new(alphabet.encode(), bool(rc), a)
        if not self._h:
            raise SassyHipError(L.sassy_hip_last_error().decode())
        self.alphabet, self.rc = alphabet.lower(), bool(rc)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().sassy_searcher_free(self._h)
            except Exception:  # interpreter shutdown: the module globals may be gone already
                pass
            self._h = None

    # --- reference API ---
    def search(self, pattern: bytes, text, k: int) -> List[Match]:
        return self._search(pattern, text, k, 0).matches

    def search_all(self, pattern: bytes, text, k: int) -> List[Match]:
        return self._search(pattern, text, k, ALL_MINIMA).matches

    def search_without_trace(self, pattern: bytes, text, k: int) -> List[Match]:
        return self._search(pattern, text, k, WITHOUT_TRACE).matches

    def search_with_fn(self, pattern: bytes, text: bytes, k: int, all_minima: bool, filter_fn) -> List[Match]:
        """Searcher::search_with_fn (src/search.rs:767-784): keep the end positions for which
        filter_fn(pattern_of_strand: bytes, text_till_end: bytes, strand: '+'|'-') is true."""
        pattern, text = bytes(pattern), bytes(text)

        def tramp(p, plen, t, end, strand, _user):
            return 1 if filter_fn(C.string_at(p, plen), C.string_at(t, end), "-" if strand else "+") else 0

        cb = END_FILTER(tramp)
        out = C.c_void_p()
        addr = C.cast(C.c_char_p(text), C.c_void_p).value or 0
        _check(lib().sassy_hip_search_with_fn(self._h, pattern, len(pattern), addr, len(text), k,
                                              ALL_MINIMA if all_minima else 0, cb, None, C.byref(out)))
        return Result(out).matches

    def search_many(self, patterns: Sequence[bytes], texts: Sequence, k: int, all_minima: bool = False, as_result: bool = False):
        """Searcher::search_many in SearchMode::Single order (src/search.rs:531-560): every pattern in
        every text, pattern-major; matches carry pattern_idx and text_idx.  as_result: the Result itself (numpy `.array`,
        `.lazy_matches`) instead of a list of Match objects."""
        patterns = [bytes(p) for p in patterns]
        pp = (C.c_char_p * len(patterns))(*patterns)
        pl = (C.c_size_t * len(patterns))(*[len(p) for p in patterns])
        if isinstance(texts, TextBatch):
            n_texts, on_device = len(texts), False
            tp = texts._addr.ctypes.data_as(C.POINTER(C.c_void_p))
            tl = texts.lens.ctypes.data_as(C.POINTER(C.c_size_t))
        elif _BYTES_PAYLOAD_OFFSET is not None and len(texts) and set(map(type, texts)) == {bytes}:
            # a read set: ctypes fills the pointer array from the list itself (a few hundred thousand _ptr_len calls
            # cost more than the search)
            import numpy as np
            n_texts, on_device = len(texts), False
            # (CPython: the bytes of a bytes object sit bytes.__basicsize__ - 1 behind its address, and an object
            # array IS the array of these addresses; `held` keeps the objects alive for the call)
            held = np.empty(n_texts, dtype=object)
            held[:] = texts
            addr = np.frombuffer((C.c_uint64 * n_texts).from_address(held.ctypes.data), dtype=np.uint64) + np.uint64(_BYTES_PAYLOAD_OFFSET)
            lens = np.fromiter(map(len, texts), dtype=np.uint64, count=n_texts)
            tp = addr.ctypes.data_as(C.POINTER(C.c_void_p))
            tl = lens.ctypes.data_as(C.POINTER(C.c_size_t))
        else:
            infos = [_ptr_len(t) for t in texts]
            on_dev = [i[3] for i in infos]
            if any(on_dev) and not all(on_dev):
                raise SassyHipError("texts must be all on the host or all on the device")
            n_texts, on_device = len(infos), bool(infos and on_dev[0])
            tp = (C.c_void_p * n_texts)(*[i[0] for i in infos])
            tl = (C.c_size_t * n_texts)(*[i[1] for i in infos])
        flags = (ALL_MINIMA if all_minima else 0) | (TEXT_ON_DEVICE if on_device else 0)
        out = C.c_void_p()
        _check(lib().sassy_hip_search_many(self._h, pp, pl, len(patterns), tp, tl, n_texts, k, flags, C.byref(out)))
        r = Result(out)
        return r if as_result else r.matches

    def search_patterns(self, patterns: Sequence[bytes], text, k: int) -> List[Match]:
        """Searcher::search_patterns (src/search.rs:648-678): equal-length patterns in one text."""
        patterns = [bytes(p) for p in patterns]
        if patterns and any(len(p) != len(patterns[0]) for p in patterns):
            raise SassyHipError("All patterns passed to search_patterns must have the same length")
        return self.search_many(patterns, [text], k)

    def search_texts(self, pattern: bytes, texts: Sequence, k: int) -> List[Match]:
        """Searcher::search_texts (src/search.rs:615-637): one pattern in many texts."""
        return self.search_many([pattern], texts, k)

    def format_tsv(self, m: Match, pat_id: str, text_id: str, text: bytes, sam: bool = False) -> str:
        """One row of the reference CLI's match table (bin/grep.rs:710-757)."""
        cm = _HipMatch()
        cm.pattern_idx, cm.text_idx = m.pattern_idx, m.text_idx
        cm.text_start, cm.text_end = m.text_start, m.text_end
        cm.pattern_start, cm.pattern_end = m.pattern_start, m.pattern_end
        cm.cost, cm.strand = m.cost, 1 if m.strand == "-" else 0
        if isinstance(text, tuple):  # (address, length) of a text that lives in somebody's buffer (fastx.RecordBatch): no copy
            addr, tlen = int(text[0]), int(text[1])
        else:
            text = bytes(text)
            addr, tlen = C.cast(C.c_char_p(text), C.c_void_p).value or 0, len(text)
        args = (self._h, C.byref(cm), m.cigar.encode(), pat_id.encode(), text_id.encode(), addr, tlen, int(sam))
        need = lib().sassy_hip_format_tsv(*args, None, 0)
        if need < 0:
            raise SassyHipError(lib().sassy_hip_last_error().decode())
        buf = C.create_string_buffer(need + 1)
        lib().sassy_hip_format_tsv(*args, buf, need + 1)
        return buf.value.decode()

    def only_best_match(self, on: bool = True) -> "Searcher":
        """Searcher::only_best_match (src/search.rs:442-446)."""
        _check(lib().sassy_hip_set_only_best_match(self._h, int(on)))
        return self

    def with_max_overhang(self, max_overhang: Optional[int]) -> "Searcher":
        """Searcher::with_max_overhang (src/search.rs:436-440)."""
        _check(lib().sassy_hip_set_max_overhang(self._h, -1 if max_overhang is None else int(max_overhang)))
        return self

    def with_max_n_frac(self, max_n_frac: Optional[float]) -> "Searcher":
        """Searcher::with_max_n_frac / without_max_n_frac (src/search.rs:454-475)."""
        _check(lib().sassy_hip_set_max_n_frac(self._h, float("nan") if max_n_frac is None else float(max_n_frac)))
        return self

    def encode_patterns(self, patterns: Sequence[bytes]) -> EncodedPatterns:
        patterns = [bytes(p) for p in patterns]
        if not patterns:
            raise SassyHipError("No queries provided")
        plen = len(patterns[0])
        if any(len(p) != plen for p in patterns):
            raise SassyHipError("All pattern must have the same length")
        h = lib().sassy_hip_encode_patterns(self._h, b"".join(patterns), len(patterns), plen)
        if not h:
            raise SassyHipError(lib().sassy_hip_last_error().decode())
        return EncodedPatterns(h, len(patterns), plen)

    def search_encoded_patterns(self, encoded: EncodedPatterns, text, k: int,
                                all_minima: bool = False, as_result: bool = False, without_trace: bool = False):
        """Searcher::search_encoded_patterns (src/search.rs:415-423).  as_result: hand back the Result
        (numpy record array + cigar pool) instead of a list of Match objects -- for result sets with
        10^5 and more matches, where a Python object per match costs more than the search.
        without_trace: end positions and costs only (text_start = pattern_start = usize::MAX, empty cigar)."""
        addr, n, keep, on_dev = _ptr_len(text)
        out = C.c_void_p()
        flags = (ALL_MINIMA if all_minima else 0) | (TEXT_ON_DEVICE if on_dev else 0) | \
            (WITHOUT_TRACE if without_trace else 0)
        _check(lib().sassy_hip_search_encoded(self._h, encoded._h, addr, n, k, flags, C.byref(out)))
        r = Result(out)
        return r if as_result else r.matches

    # --- device-resident / multi-GPU entry points ---
    def search_shard(self, pattern: bytes, d_text_ptr: int, halo_len: int, shard_len: int,
                     global_offset: int, total_len: int, k: int, flags: int = 0) -> Result:
        out = C.c_void_p()
        pattern = bytes(pattern)
        _check(lib().sassy_hip_search_shard(self._h, pattern, len(pattern), d_text_ptr, halo_len,
                                            shard_len, global_offset, total_len, k, flags,
                                            C.byref(out)))
        return Result(out)

    def search_shard_begin(self, pattern: bytes, d_text_ptr: int, halo_len: int, shard_len: int,
                           global_offset: int, total_len: int, k: int, flags: int = 0) -> int:
        """Queue one search of a resident shard and return a ticket at once (sassy_hip_search_shard_begin);
        up to 2 may be in flight.  search_finish(ticket) waits for it and returns its Result."""
        out = C.c_void_p()
        pattern = bytes(pattern)
        _check(lib().sassy_hip_search_shard_begin(self._h, pattern, len(pattern), d_text_ptr, halo_len, shard_len,
                                                  global_offset, total_len, k, flags, C.byref(out)))
        return out.value

    def set_reference_lanes(self, lanes: int):
        """0 = the definition (default); 4 / 8 = the reports of the reference binary built for AVX2 / AVX-512."""
        _check(lib().sassy_hip_set_reference_lanes(self._h, int(lanes)))
        return self

    def set_geometry_tuner(self, on: bool = True):
        _check(lib().sassy_hip_set_geometry_tuner(self._h, int(on)))
        return self

    def set_pipe_depth(self, depth: int):
        _check(lib().sassy_hip_set_pipe_depth(self._h, int(depth)))
        return self

    def search_finish(self, ticket: int) -> Result:
        out = C.c_void_p()
        _check(lib().sassy_hip_search_finish(self._h, ticket, C.byref(out)))
        return Result(out)

    def set_stream(self, hip_stream_handle: int):
        _check(lib().sassy_hip_set_stream(self._h, hip_stream_handle or None))

    def set_timing(self, level: int):
        """0 = no HIP events, 1 = dominant kernel only (default), 2 = every phase."""
        _check(lib().sassy_hip_set_timing(self._h, int(level)))

    def set_prefilter(self, mode: int):
        """-1 = the library's choice, 0 = streaming DP over every block, 1 = prefilter also with short pieces."""
        _check(lib().sassy_hip_set_prefilter(self._h, int(mode)))
        return self

    def set_device(self, device: int):
        """Bind the searcher to a HIP device before its first search (default: the calling thread's current device)."""
        _check(lib().sassy_hip_set_device(self._h, int(device)))
        return self

    @property
    def device(self) -> int:
        return lib().sassy_hip_get_device(self._h)

    def set_fused(self, on: bool = True):
        """The bit-plane prefilter finishes the scan in its own launch (default) / always the classic kernel chain."""
        _check(lib().sassy_hip_set_fused(self._h, int(bool(on))))
        return self

    def set_option(self, name: str, value: int):
        """One entry of the searcher's switch table (csrc/switches.h): name in lower case without the SASSY_HIP_ prefix."""
        _check(lib().sassy_hip_set_option(self._h, name.encode(), int(value)))
        return self

    def get_option(self, name: str) -> int:
        v = C.c_long(0)
        _check(lib().sassy_hip_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def enable_counters(self, on: bool = True):
        _check(lib().sassy_hip_enable_counters(self._h, int(on)))

    def stats(self) -> dict:
        st = Stats()
        _check(lib().sassy_hip_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def text_unchanged(self, on: bool = True) -> "Searcher":
        """Promise that a device-resident text passed to the following searches holds the same bytes as
        in this searcher's previous call with that tensor (SASSY_HIP_TEXT_UNCHANGED): the reversed copy
        the Rc strand scans is then reused instead of rebuilt for every pattern."""
        self._text_unchanged = bool(on)
        return self

    def _search(self, pattern: bytes, text, k: int, flags: int) -> Result:
        pattern = bytes(pattern)
        addr, n, keep, on_dev = _ptr_len(text)
        if on_dev:
            flags |= TEXT_ON_DEVICE
            if getattr(self, "_text_unchanged", False):
                flags |= TEXT_UNCHANGED
        out = C.c_void_p()
        _check(lib().sassy_hip_search(self._h, pattern, len(pattern), addr, n, k, flags, C.byref(out)))
        return Result(out)


def required_halo(pattern_len: int, k: int) -> int:
    return lib().sassy_hip_required_halo(pattern_len, k)


def merge_shards(results: Sequence["Result"], incoming_state: int = 1) -> "Result":
    """sassy_hip_merge_shards: the Results of consecutive shards (leftmost first) as one Result."""
    L = lib()
    hs = (C.c_void_p * len(results))(*[r._h for r in results])
    if any(h is None for h in hs):
        raise SassyHipError("merge_shards needs results that have not been materialised yet (their C records)")
    out = C.c_void_p()
    _check(L.sassy_hip_merge_shards(hs, len(results), int(incoming_state), C.byref(out)))
    return Result(out)


class MultiSearcher:
    """One text over several devices inside one process (include/sassy_hip.h: sassy_hip_multi_*): a shard and a host
    thread per entry of `devices` (None: every visible device; a device may be named more than once)."""

    def __init__(self, alphabet: str, devices: Optional[Sequence[int]] = None, alpha: float = float("nan")):
        L = lib()
        arr = (C.c_int * len(devices))(*devices) if devices else None
        self._h = L.sassy_hip_multi_new(alphabet.encode(), alpha, arr, len(devices) if devices else 0)
        if not self._h:
            raise SassyHipError(L.sassy_hip_last_error().decode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().sassy_hip_multi_free(h)
            except Exception:
                pass

    @property
    def shards(self) -> int:
        return lib().sassy_hip_multi_shards(self._h)

    def devices(self) -> List[int]:
        return [lib().sassy_hip_multi_device(self._h, i) for i in range(self.shards)]

    def set_text(self, text: bytes, max_pattern_len: int, max_k: int):
        text = bytes(text)
        _check(lib().sassy_hip_multi_set_text(self._h, text, len(text), max_pattern_len, max_k))
        return self

    def generate_dna(self, n: int, seed: int, max_pattern_len: int, max_k: int):
        _check(lib().sassy_hip_multi_generate_dna(self._h, n, seed, max_pattern_len, max_k))
        return self

    def plant(self, seed: int, pattern: bytes, k: int, stride: int = 1 << 20) -> int:
        cnt = C.c_uint64()
        pattern = bytes(pattern)
        _check(lib().sassy_hip_multi_plant(self._h, seed, pattern, len(pattern), k, stride, C.byref(cnt)))
        return cnt.value

    def search(self, pattern: bytes, k: int, flags: int = 0) -> "Result":
        out = C.c_void_p()
        pattern = bytes(pattern)
        _check(lib().sassy_hip_multi_search(self._h, pattern, len(pattern), k, flags, C.byref(out)))
        return Result(out)

    def shard_stats(self, shard: int) -> dict:
        """Stats of the last search on one shard's searcher (sassy_hip_multi_searcher)."""
        st = Stats()
        h = lib().sassy_hip_multi_searcher(self._h, shard)
        if not h:
            raise SassyHipError("no such shard")
        _check(lib().sassy_hip_get_stats(h, C.byref(st)))
        return st.as_dict()

    def set_rc(self, rc: bool = True):
        """Both strands: searches append the Rc strand's matches (Searcher::new_rc)."""
        _check(lib().sassy_hip_multi_set_rc(self._h, int(bool(rc))))
        return self

    def set_replicated(self, on: bool = True):
        """Every device holds the whole text (before set_text / generate_dna): search_encoded shards the patterns."""
        _check(lib().sassy_hip_multi_set_replicated(self._h, int(bool(on))))
        return self

    def set_pipe_depth(self, depth: int) -> "MultiSearcher":
        _check(lib().sassy_hip_multi_set_pipe_depth(self._h, int(depth)))
        return self

    def search_begin(self, pattern: bytes, k: int, flags: int = 0) -> int:
        """Queues one search on every device and returns a ticket (sassy_hip_multi_search_begin)."""
        pattern = bytes(pattern)
        t = C.c_void_p()
        _check(lib().sassy_hip_multi_search_begin(self._h, pattern, len(pattern), k, flags, C.byref(t)))
        return t.value

    def search_finish(self, ticket: int) -> "Result":
        out = C.c_void_p()
        _check(lib().sassy_hip_multi_search_finish(self._h, ticket, C.byref(out)))
        return Result(out)

    def search_encoded(self, patterns: Sequence[bytes], k: int, flags: int = 0) -> "Result":
        patterns = [bytes(p) for p in patterns]
        if not patterns:
            raise SassyHipError("No queries provided")
        plen = len(patterns[0])
        if any(len(p) != plen for p in patterns):
            raise SassyHipError("All pattern must have the same length")
        out = C.c_void_p()
        _check(lib().sassy_hip_multi_search_encoded(self._h, b"".join(patterns), len(patterns), plen, k, flags, C.byref(out)))
        return Result(out)

    def search_many(self, patterns: Sequence[bytes], texts: Sequence[bytes], k: int, flags: int = 0) -> "Result":
        patterns = [bytes(p) for p in patterns]
        texts = [bytes(t) for t in texts]
        pp = (C.c_char_p * len(patterns))(*patterns)
        pl = (C.c_size_t * len(patterns))(*[len(p) for p in patterns])
        tp = (C.c_char_p * len(texts))(*texts)
        tl = (C.c_size_t * len(texts))(*[len(t) for t in texts])
        out = C.c_void_p()
        _check(lib().sassy_hip_multi_search_many(self._h, pp, pl, len(patterns), tp, tl, len(texts), k, flags, C.byref(out)))
        return Result(out)


def multi_layout(length: int, n_parts: int, max_pattern_len: int, max_k: int):
    """(parts that hold a share, [(offset, len, halo, halo_behind, rev_first, rev_end, rev_halo)] per part) -- the layout
    arithmetic of a MultiSearcher, no device needed (sassy_hip_multi_layout); -1 parts: a share is not covered."""
    out = (C.c_uint64 * (7 * n_parts))()
    e = lib().sassy_hip_multi_layout(length, n_parts, max_pattern_len, max_k, out)
    return e, [tuple(out[7 * i:7 * i + 7]) for i in range(n_parts)]


def seed_layout(alphabet: str, patterns: Sequence[bytes], k: int):
    """[(first row, rows)] of the k + 1 seeds the seeded search of search_encoded_patterns would use for these patterns
    (sassy_hip_seed_layout: host arithmetic, no device needed)."""
    patterns = [bytes(p) for p in patterns]
    if not patterns or any(len(p) != len(patterns[0]) for p in patterns):
        raise SassyHipError("seed_layout: patterns of one length, at least one")
    pp = (C.c_char_p * len(patterns))(*patterns)
    ends, lens = (C.c_uint32 * 8)(), (C.c_uint32 * 8)()
    n = lib().sassy_hip_seed_layout(alphabet.encode(), pp, len(patterns), len(patterns[0]), k, ends, lens)
    if n < 0:
        raise SassyHipError("seed_layout: arguments out of range")
    return [(ends[i] - lens[i], lens[i]) for i in range(n)]


def seed_test_rows(pattern_len: int, k: int, seeds):
    """(win_left, largest offset, {piece: [(first row, rows, offset)]}) -- the sub-piece test's layout for the seeds
    [(first row, rows)] (sassy_hip_seed_test_rows: host arithmetic, no device needed); a piece without a test is left out."""
    ends = (C.c_uint32 * 8)(*[a + ln for a, ln in seeds])
    lens = (C.c_uint32 * 8)(*[ln for _, ln in seeds])
    rows, win = (C.c_uint32 * 64)(), C.c_uint32()
    mx = lib().sassy_hip_seed_test_rows(pattern_len, k, ends, lens, rows, C.byref(win))
    if mx < 0:
        raise SassyHipError("seed_test_rows: arguments out of range")
    out = {}
    for p in range(k + 1):
        if (rows[8 * p] & 0xFF) == 0xFF:
            continue
        out[p] = [((r & 0xFF) // 2, (32 - ((r >> 8) & 0xFF)) // 2, ((r >> 16) & 0xFF) // 2 + 16 * (r >> 24))
                  for r in (rows[8 * p + u] for u in range(k + 1))]
    return win.value, mx, out


def generate_dna(d_ptr: int, n: int, seed: int, first: int = 0, stream: int = 0):
    """Fill device memory [d_ptr, d_ptr+n) with the synthetic ACGT text."""
    _check(lib().sassy_hip_generate_dna(d_ptr, n, seed, first, stream or None))


def generate_genome_like(d_ptr: int, n: int, seed: int, first: int = 0, with_n: bool = False, stream: int = 0):
    """Fill device memory with the repeat-rich synthetic text (microsatellites, repeat families, optional N runs)."""
    _check(lib().sassy_hip_generate_genome_like(d_ptr, n, seed, first, int(with_n), stream or None))


def plant(d_ptr: int, n: int, first: int, total_n: int, seed: int, pattern: bytes, k: int,
          stride: int = 1 << 20, stream: int = 0, phase: int = 0) -> int:
    cnt = C.c_uint64()
    pattern = bytes(pattern)
    _check(lib().sassy_hip_plant_phase(d_ptr, n, first, total_n, seed, pattern, len(pattern), k, stride, phase,
                                       stream or None, C.byref(cnt)))
    return cnt.value


class DeviceBuffer:
    """hipMalloc'ed bytes for callers without torch (tests, C-style use)."""

    def __init__(self, nbytes: int):
        self.nbytes = nbytes
        self.ptr = lib().sassy_hip_malloc(nbytes)
        if not self.ptr:
            raise SassyHipError(lib().sassy_hip_last_error().decode())

    def upload(self, data: bytes, offset: int = 0):
        _check(lib().sassy_hip_memcpy_h2d(self.ptr + offset, data, len(data)))

    def download_into(self, array, offset: int = 0):
        """Device bytes [offset, offset + array.nbytes) into a writable numpy array (no intermediate copy)."""
        _check(lib().sassy_hip_memcpy_d2h(array.ctypes.data, self.ptr + offset, array.nbytes))
        return array

    def download(self, nbytes: Optional[int] = None, offset: int = 0) -> bytes:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        buf = C.create_string_buffer(nbytes)
        _check(lib().sassy_hip_memcpy_d2h(buf, self.ptr + offset, nbytes))
        return buf.raw

    def free(self):
        if self.ptr:
            lib().sassy_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()
