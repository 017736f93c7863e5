"""CPU oracle for the sassy search path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  The product (``sassy_amd`` / ``libsassy_hip.so``) never does.

Two C files are compiled into ``oracle/_build/liboracle.so``:

* ``sassy_oracle.c``  -- the definition: naive O(n*m) DP, the report rule, the traceback
  (reference: src/search.rs:1286-1369, src/trace.rs:57-104,273-406).
* ``sassy_refstyle.c`` -- the reference-shaped bit-parallel scan (4 SIMD lanes = 4 text chunks,
  bounded rows, lane pruning; reference: src/search.rs:1008-1240, src/bitpacking.rs:63-85).

Parity pinning: ``tests/golden/kats.json`` holds the reference's own known-answer tests
(SURVEY App. B); ``tests/test_oracle_kats.py`` checks this oracle against every one of them.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "liboracle.so")
_SRCS = [os.path.join(_HERE, "sassy_oracle.c"), os.path.join(_HERE, "sassy_refstyle.c")]

PROFILES = {"ascii": 0, "dna": 1, "iupac": 2}


def build(force: bool = False) -> str:
    """Compile the oracle (gcc).  Returns the path of the shared object."""
    os.makedirs(_BUILD, exist_ok=True)
    srcs_present = all(os.path.exists(s) for s in _SRCS)
    stale = (not os.path.exists(_SO)) or (
        srcs_present and any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRCS)
    )
    if force or stale:
        cmd = ["gcc", "-O3", "-mavx2", "-mbmi2", "-shared", "-fPIC", "-pthread", "-o", _SO] + _SRCS + ["-lm"]
        subprocess.check_call(cmd)
    if srcs_present:
        _build_lanes8(force)
    return _SO


_SO8 = os.path.join(_BUILD, "liboracle_lanes8.so")


def _build_lanes8(force: bool = False) -> str:
    """The same sources with RS_LANES=8 (the reference's AVX-512 lane count) for the reference-shaped port."""
    os.makedirs(_BUILD, exist_ok=True)
    stale = (not os.path.exists(_SO8)) or any(os.path.exists(x) and os.path.getmtime(x) > os.path.getmtime(_SO8) for x in _SRCS)
    if force or stale:
        subprocess.check_call(["gcc", "-O3", "-mavx2", "-mbmi2", "-DRS_LANES=8", "-shared", "-fPIC", "-pthread", "-o", _SO8]
                              + _SRCS + ["-lm"])
    return _SO8


class _OrcMatch(C.Structure):
    _fields_ = [
        ("pattern_idx", C.c_uint64),
        ("text_start", C.c_uint64),
        ("text_end", C.c_uint64),
        ("pattern_start", C.c_uint64),
        ("pattern_end", C.c_uint64),
        ("cost", C.c_int32),
        ("strand", C.c_uint8),
        ("cigar_off", C.c_uint64),
        ("cigar_len", C.c_uint32),
    ]


_lib = None
_lib8 = None


def lib8():
    """The same oracle library compiled with RS_LANES=8: the reference's AVX-512 lane count (src/lib.rs:177-185)
    for the reference-shaped port; only rs_scan / rs_lanes / rs_free are used from it."""
    global _lib8
    if _lib8 is not None:
        return _lib8
    L = C.CDLL(_build_lanes8() if all(os.path.exists(x) for x in _SRCS) else _SO8)
    L.rs_scan.restype = C.c_size_t
    L.rs_scan.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int,
                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]
    L.rs_free.restype = None
    L.rs_free.argtypes = [C.c_void_p]
    L.rs_lanes.restype = C.c_int
    _lib8 = L
    return L


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u8p = C.c_char_p
    L.orc_search.restype = C.c_void_p
    L.orc_search.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.c_int32]
    L.orc_search_encoded.restype = C.c_void_p
    L.orc_search_encoded.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_size_t, C.c_size_t, u8p,
                                     C.c_size_t, C.c_int32]
    L.orc_search_overhang.restype = C.c_void_p
    L.orc_search_overhang.argtypes = [C.c_int, C.c_int, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.c_int32,
                                      C.c_float, C.c_long]
    L.orc_result_len.restype = C.c_size_t
    L.orc_result_len.argtypes = [C.c_void_p]
    L.orc_result_failed.restype = C.c_int
    L.orc_result_failed.argtypes = [C.c_void_p]
    L.orc_result_matches.restype = C.POINTER(_OrcMatch)
    L.orc_result_matches.argtypes = [C.c_void_p]
    L.orc_result_ops.restype = C.c_void_p
    L.orc_result_ops.argtypes = [C.c_void_p]
    L.orc_result_free.restype = None
    L.orc_result_free.argtypes = [C.c_void_p]
    L.orc_last_row.restype = None
    L.orc_last_row.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p]
    L.orc_find_ends.restype = C.c_size_t
    L.orc_find_ends.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_size_t]
    L.orc_valid_pattern.restype = C.c_int
    L.orc_valid_pattern.argtypes = [C.c_int, u8p, C.c_size_t]
    L.orc_complement.restype = None
    L.orc_complement.argtypes = [C.c_int, u8p, C.c_size_t, C.c_void_p]
    L.orc_reverse_complement.restype = None
    L.orc_reverse_complement.argtypes = [C.c_int, u8p, C.c_size_t, C.c_void_p]
    L.orc_generate_dna.restype = None
    L.orc_generate_dna.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.orc_make_plant.restype = C.c_size_t
    L.orc_make_plant.argtypes = [C.c_uint64, C.c_uint64, u8p, C.c_size_t, C.c_int, C.c_void_p]
    L.orc_plant_window.restype = C.c_size_t
    L.orc_plant_window.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, u8p,
                                   C.c_size_t, C.c_int, C.c_uint64]
    L.rs_scan.restype = C.c_size_t
    L.rs_scan.argtypes = [C.c_int, u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int,
                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]
    L.rs_scan_mt.restype = C.c_size_t
    L.rs_scan_mt.argtypes = [C.c_int, u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int, C.c_double,
                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                             C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.rs_free.restype = None
    L.rs_free.argtypes = [C.c_void_p]
    L.rs_lanes.restype = C.c_int
    L.rs_encode_block_test.restype = C.c_int
    L.rs_encode_block_test.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_void_p]
    _lib = L
    return L


@dataclass(frozen=True)
class Match:
    """Same fields as the reference's Match (src/search.rs:35-62); strand '+'/'-' as in
    src/python.rs:193-198; cigar in SAM text form as produced by pa_types::Cigar::to_string."""
    pattern_idx: int
    text_start: int
    text_end: int
    pattern_start: int
    pattern_end: int
    cost: int
    strand: str
    cigar: str

    def sort_key(self):
        # the key the reference's differential test sorts by (pattern_tiling/search.rs:748-757)
        return (self.pattern_idx, self.text_start, self.text_end, self.cost, self.strand, self.cigar)


def rle_cigar(ops: bytes) -> str:
    """pa_types::Cigar: push() merges runs, to_string() prints <count><op>."""
    out = []
    i = 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        out.append(f"{j - i}{chr(ops[i])}")
        i = j
    return "".join(out)


def _profile(p) -> int:
    return PROFILES[p.lower()] if isinstance(p, str) else int(p)


def _collect(res) -> List[Match]:
    L = lib()
    try:
        if L.orc_result_failed(res):
            raise RuntimeError("oracle traceback failed (the reference would panic here)")
        n = L.orc_result_len(res)
        ms = L.orc_result_matches(res)
        ops_ptr = L.orc_result_ops(res)
        out = []
        for i in range(n):
            m = ms[i]
            ops = C.string_at(ops_ptr + m.cigar_off, m.cigar_len) if m.cigar_len else b""
            out.append(Match(m.pattern_idx, m.text_start, m.text_end, m.pattern_start,
                             m.pattern_end, m.cost, "-" if m.strand else "+", rle_cigar(ops)))
        return out
    finally:
        L.orc_result_free(res)


def search(profile, pattern: bytes, text: bytes, k: int, rc: bool = False,
           all_minima: bool = False) -> List[Match]:
    """Searcher::<P>::search / search_all on the definition (naive DP)."""
    pattern, text = bytes(pattern), bytes(text)
    res = lib().orc_search(_profile(profile), int(rc), int(all_minima), pattern, len(pattern),
                           text, len(text), k)
    return _collect(res)


def search_overhang(profile, pattern: bytes, text: bytes, k: int, alpha: float, rc: bool = False,
                    all_minima: bool = False, max_overhang: Optional[int] = None) -> List[Match]:
    """Searcher::new_{fwd,rc}_with_overhang(alpha)[.with_max_overhang(mo)].search / search_all."""
    pattern, text = bytes(pattern), bytes(text)
    res = lib().orc_search_overhang(_profile(profile), int(rc), int(all_minima), pattern, len(pattern),
                                    text, len(text), k, alpha, -1 if max_overhang is None else max_overhang)
    return _collect(res)


def search_encoded(profile, patterns: Sequence[bytes], text: bytes, k: int, rc: bool = False,
                   all_minima: bool = False) -> List[Match]:
    """search_encoded_patterns semantics; result sorted by Match.sort_key()."""
    patterns = [bytes(p) for p in patterns]
    m = len(patterns[0])
    assert all(len(p) == m for p in patterns)
    flat = b"".join(patterns)
    text = bytes(text)
    res = lib().orc_search_encoded(_profile(profile), int(rc), int(all_minima), flat,
                                   len(patterns), m, text, len(text), k)
    return sorted(_collect(res), key=Match.sort_key)


def last_row(profile, pattern: bytes, text: bytes):
    import numpy as np
    out = np.empty(len(text) + 1, dtype=np.int32)
    lib().orc_last_row(_profile(profile), bytes(pattern), len(pattern), bytes(text), len(text),
                       out.ctypes.data)
    return out


def find_ends(costs, k: int, all_minima: bool = False) -> List[Tuple[int, int]]:
    import numpy as np
    costs = np.ascontiguousarray(costs, dtype=np.int32)
    n = len(costs) - 1
    cap = n + 2
    pos = np.empty(cap, dtype=np.uint64)
    cost = np.empty(cap, dtype=np.int32)
    cnt = lib().orc_find_ends(costs.ctypes.data, n, k, int(all_minima), pos.ctypes.data,
                              cost.ctypes.data, cap)
    return [(int(pos[i]), int(cost[i])) for i in range(cnt)]


def valid_pattern(profile, pattern: bytes) -> bool:
    return bool(lib().orc_valid_pattern(_profile(profile), bytes(pattern), len(pattern)))


def complement(profile, seq: bytes) -> bytes:
    buf = C.create_string_buffer(len(seq))
    lib().orc_complement(_profile(profile), bytes(seq), len(seq), buf)
    return buf.raw


def reverse_complement(profile, seq: bytes) -> bytes:
    buf = C.create_string_buffer(len(seq))
    lib().orc_reverse_complement(_profile(profile), bytes(seq), len(seq), buf)
    return buf.raw


def refstyle_ends(profile, pattern: bytes, text, k: int, all_minima: bool = False, lanes: int = 4):
    """The reference-shaped LANES-chunk scan (lanes = 4: AVX2, 8: AVX-512).  Returns ([(end_pos, cost)...], stats)
    where stats = {'word_rows': .., 'blocks': .., 'lanes': ..}.  `text` may be bytes or a numpy
    uint8 array (no copy)."""
    import numpy as np
    L = lib() if lanes == 4 else lib8()
    assert L.rs_lanes() == lanes
    if isinstance(text, (bytes, bytearray)):
        arr = np.frombuffer(text, dtype=np.uint8)
    else:
        arr = np.ascontiguousarray(text, dtype=np.uint8)
    op, oc = C.c_void_p(), C.c_void_p()
    stats = (C.c_uint64 * 2)()
    cnt = L.rs_scan(_profile(profile), bytes(pattern), len(pattern), arr.ctypes.data, arr.size, k,
                    int(all_minima), C.byref(op), C.byref(oc), stats)
    pos = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint64)), shape=(max(cnt, 1),))[:cnt].copy()
    cost = np.ctypeslib.as_array(C.cast(oc, C.POINTER(C.c_int32)), shape=(max(cnt, 1),))[:cnt].copy()
    L.rs_free(op)
    L.rs_free(oc)
    ends = [(int(p), int(c)) for p, c in zip(pos, cost)]
    return ends, {"word_rows": int(stats[0]), "blocks": int(stats[1]), "lanes": L.rs_lanes()}


def refstyle_ends_mt(profile, pattern: bytes, text, k: int, threads: int, min_seconds: float = 1.0):
    """The reference-shaped scan on `threads` persistent host threads (one shard each, m+k+1 bytes of
    overlap rounded up to whole blocks), repeated until `min_seconds` have passed; thread creation is
    outside the clock (oracle/sassy_refstyle.c: rs_scan_mt).  Returns (ends of the last pass, info)
    with info = {'passes', 'seconds', 'shards', 'busy_seconds'}."""
    import numpy as np
    L = lib()
    if isinstance(text, (bytes, bytearray)):
        arr = np.frombuffer(text, dtype=np.uint8)
    else:
        arr = np.ascontiguousarray(text, dtype=np.uint8)
    op, oc = C.c_void_p(), C.c_void_p()
    passes, shards, secs, busy = C.c_int(), C.c_int(), C.c_double(), C.c_double()
    cnt = L.rs_scan_mt(_profile(profile), bytes(pattern), len(pattern), arr.ctypes.data, arr.size, k, int(threads),
                       float(min_seconds), C.byref(op), C.byref(oc), C.byref(passes), C.byref(secs), C.byref(shards),
                       C.byref(busy))
    pos = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint64)), shape=(max(cnt, 1),))[:cnt].copy()
    cost = np.ctypeslib.as_array(C.cast(oc, C.POINTER(C.c_int32)), shape=(max(cnt, 1),))[:cnt].copy()
    L.rs_free(op)
    L.rs_free(oc)
    ends = [(int(p), int(c)) for p, c in zip(pos, cost)]
    return ends, {"passes": passes.value, "seconds": secs.value, "shards": shards.value, "busy_seconds": busy.value}


def generate_dna(seed: int, first: int, n: int):
    """Counter-based synthetic ACGT text as a numpy uint8 array."""
    import numpy as np
    out = np.empty(n, dtype=np.uint8)
    lib().orc_generate_dna(seed, first, n, out.ctypes.data)
    return out


def make_plant(seed: int, q: int, pattern: bytes, edits: int) -> bytes:
    buf = C.create_string_buffer(len(pattern) + edits + 1)
    n = lib().orc_make_plant(seed, q, bytes(pattern), len(pattern), edits, buf)
    return buf.raw[:n]


def plant_window(seed: int, total_n: int, first: int, arr, pattern: bytes, k: int,
                 stride: int = 1 << 20) -> int:
    """Overwrite numpy uint8 array `arr` (= text[first:first+len(arr)]) with its plants."""
    return lib().orc_plant_window(seed, total_n, first, arr.size, arr.ctypes.data, bytes(pattern),
                                  len(pattern), k, stride)


def profile_masks(profile, pattern: bytes, block64: bytes) -> List[int]:
    """Per-slot u64 equality masks of one 64-byte text block (Profile::encode_ref)."""
    assert len(block64) == 64
    out = (C.c_uint64 * 256)()
    n = lib().rs_encode_block_test(_profile(profile), bytes(pattern), len(pattern), bytes(block64), out)
    return [int(out[i]) for i in range(n)]


# ---------------------------------------------------------------- reporting modes of the Searcher
def _n_frac_ok(n_count: int, denom: int, max_n_frac: float) -> bool:
    import numpy as np
    return bool(np.float32(n_count) / np.float32(denom) <= np.float32(max_n_frac))


def _count_n(text: bytes, a: int, b: int) -> int:
    return sum(1 for c in text[a:b] if c | 0x20 == 0x6E)


def search_modes(profile, pattern: bytes, text: bytes, k: int, rc: bool = False, all_minima: bool = False,
                 end_filter=None, max_n_frac=None, only_best: bool = False, without_trace: bool = False,
                 alpha: Optional[float] = None):
    """What Searcher::search_one_strand does around the scan (src/search.rs:884-937), per strand:
    end-position callback (search_with_fn, :895-906), N-fraction pre-filter on the end position
    (src/n_filter.rs:38-52), only_best_match (:1392-1412: minimal cost, rightmost end), N-fraction
    filter on the traced span (src/n_filter.rs:54-60); the Rc strand sees complement(pattern) and
    the reversed text, results are mapped back (:813-878).
    end_filter(pattern_of_strand, text_till_end, strand '+'/'-') -> bool."""
    pattern, text = bytes(pattern), bytes(text)
    n, m = len(text), len(pattern)
    out = []
    strands = [("+", pattern, text)]
    if rc:
        strands.append(("-", complement(profile, pattern), text[::-1]))
    for strand, pat, txt in strands:
        if alpha is not None:
            ms = search_overhang(profile, pat, txt, k, alpha, rc=False, all_minima=all_minima)
        else:
            ms = search(profile, pat, txt, k, rc=False, all_minima=all_minima)
        if end_filter is not None:
            ms = [x for x in ms if end_filter(pat, txt[:min(x.text_end, n)], strand)]
        if max_n_frac is not None and max_n_frac != 1.0:
            keep = []
            for x in ms:
                end = min(x.text_end, n)
                start = end - min(end, max(0, m - k))
                if start >= n or start == end or _n_frac_ok(_count_n(txt, start, end), m + k, max_n_frac):
                    keep.append(x)
            ms = keep
        if only_best and ms:
            best = min(ms, key=lambda x: (x.cost, -x.text_end))
            ms = [best]
        if max_n_frac is not None and max_n_frac != 1.0 and not without_trace:
            ms = [x for x in ms if x.text_start >= n or x.text_end == x.text_start or
                  _n_frac_ok(_count_n(txt, x.text_start, x.text_end), x.text_end - x.text_start, max_n_frac)]
        for x in ms:
            if strand == "-":
                x = Match(x.pattern_idx, n - x.text_end, n - x.text_start, x.pattern_start, x.pattern_end,
                          x.cost, "-", x.cigar)
            out.append(x)
    return out
