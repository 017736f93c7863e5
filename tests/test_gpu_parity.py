"""GPU parity tests: the HIP path (through the C-ABI of libsassy_hip.so) against the CPU oracle
and the reference's known-answer tests.  Bit-exact: integer / byte / index work only.

Run on the MI355X box with `pytest -m gpu`."""
import ctypes as C
import json
import os
import random
import sys
import time

import numpy as np
import pytest

import oracle
from conftest import build_text, cigar_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sassy():
    import sassy_amd
    assert sassy_amd.device_count() > 0, "no HIP device: the GPU tests must not silently skip"
    return sassy_amd


def key(m):
    return (m.pattern_idx, m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost,
            m.strand, m.cigar)


def assert_same(got, want, ctx=None):
    g, w = [key(m) for m in got], [key(m) for m in want]
    assert g == w, (ctx, g[:5], w[:5], len(g), len(w))


def canon(r):
    """A Result's records as comparable bytes: the match rows and each row's cigar string (the bytes of the string
    pool behind a terminating NUL are unspecified)."""
    a, pool = r.array, r.pool
    return a.tobytes(), tuple(bytes(pool[int(o):int(o) + int(l)]) for o, l in zip(a["cigar_off"], a["cigar_len"]))


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, edits):
    s = bytearray(s)
    for _ in range(edits):
        t, p = rng.randrange(3), rng.randrange(len(s))
        if t == 0:
            s[p] = rng.choice(b"ACGT")
        elif t == 1:
            s.insert(p, rng.choice(b"ACGT"))
        elif len(s) > 1:
            del s[p]
    return bytes(s)


# ------------------------------------------------------------------ reference KATs via HIP
def test_reference_kats_through_hip(sassy, kats):
    for e in kats["search"]:
        s = sassy.Searcher(e["profile"], rc=e["rc"])
        text = build_text(e)
        pat = e["pattern"].encode()
        ms = s.search_all(pat, text, e["k"]) if e["mode"] == "search_all" else s.search(pat, text, e["k"])
        if "expect_len" in e:
            assert len(ms) == e["expect_len"], (e["id"], ms)
        for m, exp in zip(ms, e.get("expect", [])):
            for f, v in exp.items():
                assert getattr(m, f) == v, (e["id"], f, m)
        if "expect_first" in e:
            for f, v in e["expect_first"].items():
                if f == "path":
                    assert cigar_path(ms[0]) == [tuple(x) for x in v]
                else:
                    assert getattr(ms[0], f) == v, (e["id"], f, ms[0])
        if "expect_text_ends" in e:
            assert [m.text_end for m in ms] == e["expect_text_ends"]
        # and the oracle agrees field by field, cigar included
        want = oracle.search(e["profile"], pat, text, e["k"], rc=e["rc"], all_minima=(e["mode"] == "search_all"))
        assert_same(ms, want, e["id"])


def test_encoded_kats_through_hip(sassy, kats):
    for e in kats["encoded"]:
        s = sassy.Searcher(e["profile"], rc=e["rc"])
        enc = s.encode_patterns([p.encode() for p in e["patterns"]])
        ms = s.search_encoded_patterns(enc, e["text"].encode(), e["k"], all_minima=e.get("all", False))
        assert len(ms) == e["expect_len"], (e["id"], ms)
        if "expect_text_starts_in_order" in e:
            assert [m.text_start for m in ms] == e["expect_text_starts_in_order"]
        for pidx, exp in e.get("expect_by_pattern", {}).items():
            m = [x for x in ms if x.pattern_idx == int(pidx)][0]
            for f, v in exp.items():
                assert getattr(m, f) == v
        want = oracle.search_encoded(e["profile"], [p.encode() for p in e["patterns"]],
                                     e["text"].encode(), e["k"], rc=e["rc"], all_minima=e.get("all", False))
        assert_same(ms, want, e["id"])


def test_not_rev_invariant_through_hip(sassy, kats):
    e = kats["not_rev_invariant"]
    s = sassy.Searcher(e["profile"], rc=False)
    p, t = e["pattern"].encode(), e["text"].encode()
    assert len(s.search(p, t, e["k"])) != len(s.search(p[::-1], t[::-1], e["k"]))


def test_drop_in_c_abi_call_sequence(sassy):
    """The exact call sequence of the reference's c/example.c:14-29 against the drop-in symbols."""
    L = sassy.lib()
    s = L.sassy_searcher(b"dna", True, float("nan"))
    pat = b"AAGGGGA"
    text = b"CCCCCCCCCAAGGGGACCCCCAAGGCGACCCCCCCCC"
    out = C.POINTER(sassy.CMatch)()
    n = L.search(s, pat, len(pat), text, len(text), 1, C.byref(out))
    got = [(out[i].text_start, out[i].text_end, out[i].pattern_start, out[i].pattern_end,
            out[i].cost, out[i].strand) for i in range(n)]
    L.sassy_matches_free(out, n)
    want = oracle.search("dna", pat, text, 1, rc=True)
    assert got == [(m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost,
                    1 if m.strand == "-" else 0) for m in want]
    assert got[:2] == [(9, 16, 0, 7, 0, 0), (21, 28, 0, 7, 1, 0)]
    # zero matches: still a non-null pointer that sassy_matches_free accepts (src/c.rs:112-127)
    n0 = L.search(s, b"TTTTTTTT", 8, b"CCCCCCCCCCCCCCCC", 16, 0, C.byref(out))
    assert n0 == 0 and bool(out)
    L.sassy_matches_free(out, 0)
    L.sassy_searcher_free(s)


# ------------------------------------------------------------------ differential fuzz vs oracle
@pytest.mark.parametrize("profile", ["dna", "iupac"])
def test_fuzz_small_texts(sassy, profile):
    rng = random.Random(42 if profile == "dna" else 43)
    fwd = sassy.Searcher(profile, rc=False)
    both = sassy.Searcher(profile, rc=True)
    for it in range(400):
        m = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 23, 31, 32, 33, 40, 63, 64, 65, 100, 130])
        k = min(rng.choice([0, 0, 1, 2, 3, 3, 5, 8]), m - 1) if m > 1 else 0
        n = rng.choice([0, 1, 2, 31, 63, 64, 65, 100, 127, 128, 129, 200, 500, 511, 512, 513, 1500, 4000])
        pal = b"ACGT" if profile == "dna" else b"ACGTNRYSWKMBDHV"
        pat = rand_seq(rng, m, pal if rng.random() < 0.3 else b"ACGT")
        text = bytearray(rand_seq(rng, n, b"ACGT" if rng.random() < 0.7 else (b"ACGTNacgtn" if profile == "iupac" else b"ACGTacgt")))
        for _ in range(rng.randrange(5)):
            if n > m + 8:
                at = rng.randrange(0, n - m - 6)
                ins = mutate(rng, bytes(c if c in b"ACGT" else 65 for c in pat), rng.randrange(k + 2))
                text[at:at + len(ins)] = ins
        text = bytes(text[:n])
        rc = rng.random() < 0.4
        allm = rng.random() < 0.25
        s = both if rc else fwd
        got = s.search_all(pat, text, k) if allm else s.search(pat, text, k)
        want = oracle.search(profile, pat, text, k, rc=rc, all_minima=allm)
        assert_same(got, want, (it, profile, m, k, n, rc, allm, pat, text))


def test_low_complexity_and_seams(sassy):
    """Plateaus that straddle many lane chunks: poly-A, periodic text, long constant-cost runs.
    The oracle is the un-chunked definition; the HIP path must agree exactly (this exercises the
    conditional reports and the chunk exit-state chain)."""
    s = sassy.Searcher("dna", rc=False)
    cases = []
    cases.append((b"A" * 20, b"A" * 5000, 3))
    cases.append((b"A" * 20, b"A" * 92 + (b"C" + b"A" * 19) * 43 + b"G" * 51, 3))   # SURVEY A.5 shape
    cases.append((b"A" * 20, b"G" * 700 + b"A" * 3000 + b"G" * 900, 2))
    cases.append((b"A" * 20, b"A" * 3000 + b"C" + b"A" * 3000, 3))
    cases.append((b"ACAC" * 6, b"AC" * 4000, 2))
    cases.append((b"ACGT" * 8, (b"ACGT" * 8 + b"T") * 300, 3))
    cases.append((b"A" * 10, (b"A" * 9 + b"C") * 700, 1))
    cases.append((b"A" * 40, b"A" * 64 * 40, 5))
    rng = random.Random(5)
    for _ in range(60):
        per = rng.randrange(1, 12)
        unit = rand_seq(rng, per, b"AC")
        n = rng.randrange(600, 9000)
        text = bytearray((unit * (n // per + 1))[:n])
        for _ in range(rng.randrange(4)):
            text[rng.randrange(n)] = rng.choice(b"ACGT")
        m = rng.randrange(6, 45)
        pat = (unit * (m // per + 1))[:m]
        cases.append((pat, bytes(text), rng.randrange(0, 4)))
    for i, (pat, text, k) in enumerate(cases):
        k = min(k, len(pat) - 1)
        got = s.search(pat, text, k)
        want = oracle.search("dna", pat, text, k)
        assert_same(got, want, (i, pat, k, len(text)))
        got = s.search_all(pat, text[:1500], k)
        want = oracle.search("dna", pat, text[:1500], k, all_minima=True)
        assert_same(got, want, (i, "all"))
    assert s.stats()["cond_resolved"] >= 0


def test_long_pattern_iupac_config3_shape(sassy):
    """BASELINE config 3 shape at oracle-checkable size: |pattern|=200 with IUPAC letters, k=20."""
    rng = random.Random(44)
    pat = bytearray(oracle.generate_dna(44, 0, 200).tobytes())
    pat[50], pat[100], pat[150], pat[199] = ord("N"), ord("R"), ord("Y"), ord("W")
    pat = bytes(pat)
    n = 200_000
    text = oracle.generate_dna(42, 0, n)
    plain = bytes(c if c in b"ACGT" else 65 for c in pat)
    for q in range(12):
        ins = mutate(rng, plain, q * 2)
        at = 5000 + q * 15000
        text[at:at + len(ins)] = np.frombuffer(ins, dtype=np.uint8)
    tb = text.tobytes()
    s = sassy.Searcher("iupac", rc=False)
    got = s.search(pat, tb, 20)
    want = oracle.search("iupac", pat, tb, 20)
    assert len(want) >= 10
    assert_same(got, want)


@pytest.mark.parametrize("profile", ["dna", "iupac"])
def test_traceback_variants(sassy, profile):
    """Every traceback kernel variant against the oracle: k = 0..6 (band row in registers),
    k = 7..30 (one wavefront per report), k > 30 (generic per-thread band, LDS or global scratch),
    including text-start / text-end windows and search_all (dense, adjacent reports)."""
    rng = random.Random(7 if profile == "dna" else 8)
    cases = [(12, 0), (16, 1), (20, 2), (24, 4), (33, 5), (40, 6), (40, 7), (64, 9), (90, 13),
             (130, 20), (200, 30), (70, 31), (120, 35), (300, 30), (100, 9), (160, 15), (1000, 40), (2100, 12)]
    import os
    if os.environ.get("SASSY_HIP_PREFILTER") == "0":
        # the streaming DP keeps a pattern's carries in LDS: m up to about 1 800 (list mode: registers / more room)
        cases = [c for c in cases if c[0] <= 1500]
    for m, k in cases:
        pat = bytes(rng.choice(b"ACGT") for _ in range(m))
        if profile == "iupac" and m >= 20:
            p = bytearray(pat)
            p[3], p[m // 2], p[m - 2] = ord("N"), ord("R"), ord("y")
            pat = bytes(p)
        plain = bytes(c if c in b"ACGT" else 65 for c in pat.upper())
        n = 6000 + 10 * m
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        # a match cut by the text start, several inside, one cut by the text end
        head = mutate(rng, plain, min(k, 2))[m // 3:]
        text[0:len(head)] = head
        at = 300
        for q in range(6):
            ins = mutate(rng, plain, min(k, q * max(1, k // 5)))
            text[at:at + len(ins)] = ins
            at += len(ins) + 200 + 97 * q
        tail = mutate(rng, plain, min(k, 1))
        text[n - len(tail):] = tail
        tb = bytes(text[:n])
        s = sassy.Searcher(profile, rc=False)
        want = oracle.search(profile, pat, tb, k)
        assert len(want) >= 4, (m, k)
        assert_same(s.search(pat, tb, k), want)
        if m <= 130:
            assert_same(s.search_all(pat, tb, k), oracle.search(profile, pat, tb, k, all_minima=True))


def test_dense_reports(sassy):
    """Match-dense text: > 8192 reports (thread-per-report traceback instead of the wave-per-report
    kernel), > 32768 reports (host sort instead of the device ranking pass), and more reports than
    the initial candidate buffer holds (retry with a grown buffer)."""
    rng = random.Random(99)
    pat = b"ACGTTGCAAGGCTTACGATC"
    for reps, k, more_than in ((1500, 3, 8192), (6500, 2, 8192), (14500, 3, 73792)):
        unit = bytearray()
        for _ in range(reps):
            unit += mutate(rng, pat, rng.randrange(0, 3)) + bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(0, 4)))
        tb = bytes(unit)
        s = sassy.Searcher("dna", rc=False)
        want = oracle.search("dna", pat, tb, k, all_minima=True)
        assert len(want) > more_than, len(want)
        assert_same(s.search_all(pat, tb, k), want)
        assert_same(s.search(pat, tb, k), oracle.search("dna", pat, tb, k))


def test_qgram_table_filter_text_letters(sassy):
    """The q-gram table prefilter (Iupac profile, or > 8 pieces) must stay exact when the TEXT holds
    letters other than ACGT: ambiguity codes (match several bases), lower case, U, non-letters."""
    rng = random.Random(2024)
    for profile, m, k in (("iupac", 32, 3), ("iupac", 48, 4), ("iupac", 90, 9), ("dna", 100, 9)):
        pat = bytearray(rng.choice(b"ACGT") for _ in range(m))
        if profile == "iupac":
            pat[5], pat[m // 2], pat[m - 3] = ord("N"), ord("S"), ord("w")
        pat = bytes(pat)
        plain = bytes(c if c in b"ACGT" else 67 for c in pat.upper())
        n = 40_000
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        at = 100
        while at + 2 * m < n:
            ins = bytearray(mutate(rng, plain, rng.randrange(0, k + 1)))
            # replace some bases of the planted copy by codes that still (or no longer) match
            # (the Dna profile only defines ACGT text, any case: the reference panics otherwise)
            letters = b"NRYKMSWBDHVnacgtuUX-*@" if profile == "iupac" else b"acgtACGT"
            for _ in range(rng.randrange(0, 4)):
                ins[rng.randrange(len(ins))] = rng.choice(letters)
            text[at:at + len(ins)] = ins
            at += len(ins) + rng.randrange(50, 900)
        for _ in range(200):  # stray letters in the background, some right at block borders
            text[rng.randrange(n)] = rng.choice(b"NRYnu@-" if profile == "iupac" else b"acgt")
        for b in (63, 64, 127, 128, 4095, 4096):
            text[b] = ord("N") if profile == "iupac" else ord("g")
        tb = bytes(text)
        s = sassy.Searcher(profile, rc=False)
        want = oracle.search(profile, pat, tb, k)
        assert len(want) >= 20, (profile, m, k, len(want))
        assert_same(s.search(pat, tb, k), want)
        st = s.stats()
        import os
        if os.environ.get("SASSY_HIP_PREFILTER") != "0" and not os.environ.get("SASSY_HIP_FILTER_KIND"):
            assert st["filtered"] in (3, 4), st["filtered"]  # a q-gram table kernel really ran


def test_qgram_count_filter_worst_case_edits(sassy):
    """The q-gram counting prefilter at the edge of its lemma: copies of the pattern with exactly k
    edits spaced so that every edit destroys as many q-grams as it can (every q-th row), at the very
    start and end of the text, across block borders, for window sizes on either side of a 64 multiple."""
    import os
    rng = random.Random(77)
    cases = [("iupac", 32, 3), ("iupac", 33, 3), ("iupac", 24, 2), ("iupac", 20, 2), ("iupac", 64, 8),
             ("iupac", 69, 1), ("iupac", 70, 1), ("iupac", 133, 1), ("iupac", 134, 1), ("iupac", 200, 20),
             ("iupac", 500, 50), ("dna", 100, 10), ("dna", 90, 11), ("dna", 260, 30)]
    for profile, m, k in cases:
        pat = bytearray(rng.choice(b"ACGT") for _ in range(m))
        if profile == "iupac" and m >= 32:
            pat[7], pat[m // 2] = ord("R"), ord("N")
        pat = bytes(pat)
        plain = bytes(c if c in b"ACGT" else 65 for c in pat)
        n = 30_000
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))

        def worst(q):
            out = bytearray()
            edits = 0
            for i, c in enumerate(plain):
                if edits < k and i % q == q - 1:
                    edits += 1
                    kind = rng.randrange(3)
                    if kind == 0:
                        out.append(rng.choice([x for x in b"ACGT" if x != c]))   # substitution
                    elif kind == 1:
                        out.append(c); out.append(rng.choice(b"ACGT"))          # insertion
                    # kind == 2: deletion
                else:
                    out.append(c)
            return bytes(out)

        spots = [0, 64 * 7 - m // 2, 5000 + 63, 9000 + 64, 15000]
        for at in spots:
            for q in (5, 6, 7, max(2, m // (k + 1))):
                ins = worst(q)
                text[at:at + len(ins)] = ins
                at += 2 * m + 130
        tail = worst(6)
        text[n - len(tail):] = tail
        tb = bytes(text)
        s = sassy.Searcher(profile, rc=False)
        want = oracle.search(profile, pat, tb, k)
        assert len(want) >= 10, (profile, m, k, len(want))
        assert_same(s.search(pat, tb, k), want)
        assert_same(s.search_all(pat, tb, k), oracle.search(profile, pat, tb, k, all_minima=True))
        if os.environ.get("SASSY_HIP_PREFILTER") is None and not os.environ.get("SASSY_HIP_FILTER_KIND"):
            # (2: the Dna launch with its text check, taken by Iupac searches with plain patterns of <= 4 pieces on plain
            # text; the counting filter has these shapes under SASSY_HIP_IUPAC_PLANES=0 and FILTER_KIND=4)
            assert s.stats()["filtered"] in (2, 3, 4), (profile, m, k, s.stats()["filtered"])


def test_many_pieces_without_a_filter_stream(sassy):
    """Ascii with more piece rows than any prefilter takes (41 pieces of 9 rows): the search must fall
    back to the streaming DP, not refuse (found by tests/fuzz_gpu.py)."""
    rng = random.Random(9)
    m, k = 400, 40
    pat = bytes(rng.choice(b"ACGTXYZ acgt") for _ in range(m))
    text = bytearray(rng.choice(b"ACGT xyz") for _ in range(20_000))
    for at in (0, 5000, 12_345, 20_000 - m):
        text[at:at + m] = mutate(rng, pat, rng.randrange(0, k + 1))[:m]
    s = sassy.Searcher("ascii", rc=False)
    want = oracle.search("ascii", pat, bytes(text), k)
    assert len(want) >= 3
    assert_same(s.search(pat, bytes(text), k), want)


def test_rc_strand_all_text_alignments_and_resident_text_reuse(sassy):
    """The Rc strand scans a reversed copy of the text (vectorised reverse_kernel: aligned loads +
    v_perm): every text length mod 16 and mod 64, tiny texts included; then the same device-resident
    text searched again under SASSY_HIP_TEXT_UNCHANGED (reversed copy reused) with other patterns."""
    rng = random.Random(123)
    s = sassy.Searcher("dna", rc=True)
    for n in list(range(0, 70)) + [127, 128, 129, 1000, 1001, 1007, 1015, 1016, 4099, 65_537]:
        text = bytes(rng.choice(b"ACGT") for _ in range(n))
        pat = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([3, 8, 20])))
        if n > 40:
            ins = oracle.reverse_complement("dna", pat)
            at = rng.randrange(0, n - len(ins))
            text = text[:at] + ins + text[at + len(ins):]
        k = rng.randrange(0, 2)
        assert_same(s.search(pat, text, k), oracle.search("dna", pat, text, k, rc=True), (n, pat))
    # both strands from one forward pass: bit-plane filter with 1-4 pieces per strand (one launch),
    # 5-8 pieces (second launch for the Rc pieces), counting filter (Iupac / many pieces), long
    # patterns (word-pipelined chunk DP reading the text backwards), search_all
    for profile, m, k in (("dna", 32, 3), ("dna", 24, 0), ("dna", 60, 5), ("dna", 64, 7), ("dna", 120, 12),
                          ("iupac", 32, 3), ("iupac", 150, 15), ("iupac", 20, 1)):
        n = 50_003 + m
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        pat = bytes(rng.choice(b"ACGT") for _ in range(m))
        for t in range(12):
            src = oracle.reverse_complement(profile, pat) if t % 2 else pat
            ins = mutate(rng, src, rng.randrange(0, k + 1))
            at = [0, n - len(ins)][t] if t < 2 else rng.randrange(0, n - len(ins))
            text[at:at + len(ins)] = ins
        text = bytes(text)
        sb = sassy.Searcher(profile, rc=True)
        assert_same(sb.search(pat, text, k), oracle.search(profile, pat, text, k, rc=True), (profile, m, k))
        assert_same(sb.search_all(pat, text, k), oracle.search(profile, pat, text, k, rc=True, all_minima=True), (profile, m, k))
    # device-resident text, many patterns
    n = 200_003
    text = bytearray(rng.choice(b"ACGT") for _ in range(n))
    pats = [bytes(rng.choice(b"ACGT") for _ in range(24)) for _ in range(6)]
    for p in pats:
        for strand in (0, 1):
            ins = mutate(rng, oracle.reverse_complement("dna", p) if strand else p, rng.randrange(0, 3))
            at = rng.randrange(0, n - 40)
            text[at:at + len(ins)] = ins
    text = bytes(text)
    buf = sassy.DeviceBuffer(n + 64)
    buf.upload(text)
    dev = _DevText(buf.ptr, n)
    s2 = sassy.Searcher("dna", rc=True)
    assert_same(s2.search(pats[0], dev, 2), oracle.search("dna", pats[0], text, 2, rc=True))
    s2.text_unchanged(True)
    for p in pats:
        assert_same(s2.search(p, dev, 2), oracle.search("dna", p, text, 2, rc=True))
    # a changed text without the promise is picked up again
    text2 = bytes(reversed(text))
    buf.upload(text2)
    s2.text_unchanged(False)
    assert_same(s2.search(pats[1], dev, 2), oracle.search("dna", pats[1], text2, 2, rc=True))
    buf.free()


def test_reporting_modes(sassy, kats):
    """search_with_fn, only_best_match, max_n_frac: the reference's known
    answers, then seeded fuzz against the oracle's restatement of src/search.rs:884-937."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for e in kats["modes"]:
        pat, text = e["pattern"].encode(), build_text(e)
        s = sassy.Searcher(e["profile"], rc=e["rc"])
        if "end_filter" in e:
            fns = {
                "text_len_gt_10_plus_m": lambda q, t, strand: len(t) > 10 + len(q),
                "suffix_is_pattern_fwd_or_complement":
                    lambda q, t, strand: (t[len(t) - len(q):] if strand == "+" else t[len(t) - len(q):].translate(comp)) == pat,
            }
            ms = s.search_with_fn(pat, text, e["k"], e["mode"] == "search_all", fns[e["end_filter"]])
            assert [m.text_start for m in ms] == e["expect_text_start"], (e["id"], ms)
        else:
            s.with_max_n_frac(e["max_n_frac"])
            ms = s.search_all(pat, text, e["k"])
            assert [m.text_end for m in ms] == e["expect_text_end"], (e["id"], ms)
    rng = random.Random(31)
    for it in range(30):
        profile = "iupac"
        m, k = rng.randrange(8, 40), rng.randrange(0, 4)
        pat = bytes(rng.choice(b"ACGT") for _ in range(m))
        n = rng.randrange(200, 3000)
        text = bytearray(rng.choice(b"ACGTN") if rng.random() < 0.3 else rng.choice(b"ACGT") for _ in range(n))
        for _ in range(rng.randrange(1, 6)):
            ins = bytearray(mutate(rng, pat, rng.randrange(0, k + 1)))
            for _ in range(rng.randrange(0, 5)):
                ins[rng.randrange(len(ins))] = ord("N")
            at = rng.randrange(0, n - len(ins))
            text[at:at + len(ins)] = ins
        tb = bytes(text)
        rc = bool(it & 1)
        frac = rng.choice([0.0, 0.1, 0.25, 0.5])
        allm = bool(it & 2)
        # N-fraction filter
        s = sassy.Searcher(profile, rc=rc).with_max_n_frac(frac)
        got = s.search_all(pat, tb, k) if allm else s.search(pat, tb, k)
        assert_same(got, oracle.search_modes(profile, pat, tb, k, rc=rc, all_minima=allm, max_n_frac=frac))
        # only_best_match (+ the N filter in front of it)
        s = sassy.Searcher(profile, rc=rc).only_best_match()
        assert_same(s.search(pat, tb, k), oracle.search_modes(profile, pat, tb, k, rc=rc, only_best=True))
        s.with_max_n_frac(frac)
        assert_same(s.search(pat, tb, k),
                    oracle.search_modes(profile, pat, tb, k, rc=rc, only_best=True, max_n_frac=frac))
        # end-position callback: "the base before the match end is not G" (a PAM-like test)
        fn = lambda q, t, strand: len(t) >= 2 and t[-2] != ord("G")
        s = sassy.Searcher(profile, rc=rc)
        assert_same(s.search_with_fn(pat, tb, k, allm, fn),
                    oracle.search_modes(profile, pat, tb, k, rc=rc, all_minima=allm, end_filter=fn))


def test_search_many_and_tsv(sassy):
    """search_many / search_patterns / search_texts = independent searches with
    pattern_idx / text_idx, pattern-major; TSV rows as the reference CLI prints them (row 2)."""
    rng = random.Random(5)
    pats = [bytes(rng.choice(b"ACGT") for _ in range(16)) for _ in range(5)]
    texts = []
    for t in range(4):
        n = rng.randrange(300, 5000)
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for p in pats:
            ins = mutate(rng, p, rng.randrange(0, 3))
            if rng.random() < 0.5:
                ins = oracle.reverse_complement("dna", ins)
            at = rng.randrange(0, n - len(ins))
            text[at:at + len(ins)] = ins
        texts.append(bytes(text))
    texts.append(b"")  # an empty text has no matches
    for rc in (False, True):
        s = sassy.Searcher("dna", rc=rc)
        got = s.search_many(pats, texts, 2)
        want = []
        for pi, p in enumerate(pats):
            for ti, t in enumerate(texts):
                for m in oracle.search("dna", p, t, 2, rc=rc):
                    want.append((pi, ti, m.text_start, m.text_end, m.cost, m.strand, m.cigar))
        assert [(m.pattern_idx, m.text_idx, m.text_start, m.text_end, m.cost, m.strand, m.cigar) for m in got] == want
        assert len(want) >= 5
        one_text = s.search_patterns(pats, texts[0], 2)
        assert [(m.pattern_idx, m.text_start, m.cigar) for m in one_text] == \
            [(w[0], w[2], w[6]) for w in want if w[1] == 0]
        one_pat = s.search_texts(pats[1], texts, 2)
        assert [(m.text_idx, m.text_start, m.cigar) for m in one_pat] == [(w[1], w[2], w[6]) for w in want if w[0] == 1]
        # TSV rows (bin/grep.rs:710-757)
        for m in got:
            t = texts[m.text_idx]
            region = t[m.text_start:m.text_end]
            for sam in (False, True):
                row = s.format_tsv(m, f"pat{m.pattern_idx}", f"text{m.text_idx}", t, sam=sam)
                if m.strand == "-" and not sam:
                    region_out = oracle.reverse_complement("dna", region)
                else:
                    region_out = region
                cig = m.cigar
                if m.strand == "-" and sam:
                    import re
                    cig = "".join(reversed(re.findall(r"\d+[=XID]", cig)))
                assert row == f"pat{m.pattern_idx}\ttext{m.text_idx}\t{m.cost}\t{m.strand}\t{m.text_start}\t" \
                              f"{m.text_end}\t{region_out.decode()}\t{cig}\n"
    assert sassy.lib().sassy_hip_tsv_header() == b"pat_id\ttext_id\tcost\tstrand\tstart\tend\tmatch_region\tcigar\n"
    # the reference's own format expectations (bin/grep.rs:800-817)
    s = sassy.Searcher("dna", rc=True)
    m = sassy.Match(0, 0, 4, 0, 4, 0, "-", "2=1X3D")
    assert s.format_tsv(m, "p", "t", b"AAGT", sam=False).split("\t")[6:] == ["ACTT", "2=1X3D\n"]
    assert s.format_tsv(m, "p", "t", b"AAGT", sam=True).split("\t")[6:] == ["AAGT", "3D1X2=\n"]


def test_cli_search_tsv(sassy, tmp_path, capsys):
    """`python -m sassy_amd search`: the reference CLI's TSV table for FASTA input,
    defaults as in bin/grep.rs (iupac, rc on, max_n_frac 0.2), rows checked against the oracle."""
    import gzip
    import re
    from sassy_amd import cli
    rng = random.Random(77)
    pats = [("guideA", b"ACGTTGCAAGGCTTACGATC"), ("guideB", b"TTGACCAGTNACGGATCCAT")]
    recs = []
    for r in range(3):
        n = rng.randrange(500, 4000)
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for _, p in pats:
            plain = bytes(c if c in b"ACGT" else 67 for c in p)
            for _ in range(2):
                ins = mutate(rng, plain, rng.randrange(0, 3))
                if rng.random() < 0.5:
                    ins = oracle.reverse_complement("iupac", ins)
                at = rng.randrange(0, n - len(ins))
                text[at:at + len(ins)] = ins
        at = rng.randrange(0, n - 30)
        text[at:at + 12] = b"N" * 12  # an N run: matches over it are dropped by max_n_frac
        recs.append((f"rec{r} some description", bytes(text)))
    fa = tmp_path / "texts.fa.gz"
    with gzip.open(fa, "wb") as fh:
        for rid, seq in recs:
            fh.write(b">" + rid.encode() + b"\n")
            for i in range(0, len(seq), 60):
                fh.write(seq[i:i + 60] + b"\n")
    pf = tmp_path / "pats.fa"
    pf.write_bytes(b"".join(b">" + i.encode() + b"\n" + p + b"\n" for i, p in pats))
    for sam in (False, True):
        assert cli.main(["search", "-f", str(pf), "-k", "2", str(fa)] + (["--sam"] if sam else [])) == 0
        lines = capsys.readouterr().out.splitlines()
        assert lines[0] == "pat_id\ttext_id\tcost\tstrand\tstart\tend\tmatch_region\tcigar"
        want = []
        for rid, seq in recs:
            for pid, p in pats:
                for m in oracle.search_modes("iupac", p, seq, 2, rc=True, max_n_frac=0.2):
                    region = seq[m.text_start:m.text_end]
                    cig = m.cigar
                    if m.strand == "-" and not sam:
                        region = oracle.reverse_complement("iupac", region)
                    if m.strand == "-" and sam:
                        cig = "".join(reversed(re.findall(r"\d+[=XID]", cig)))
                    want.append(f"{pid}\t{rid}\t{m.cost}\t{m.strand}\t{m.text_start}\t{m.text_end}\t{region.decode()}\t{cig}")
        assert len(want) >= 8
        assert lines[1:] == want


def test_batched_texts(sassy):
    """Many short texts go through ONE buffer with 'X' separators (host.hip: search_many_batched).
    Every (pattern, text) pair must equal the independent search: matches cut by the text end
    (end-of-text rule -> a plateau that runs into the separator), matches at the text start, empty
    and tiny texts, search_all, without_trace, only_best_match, max_n_frac, both strands."""
    rng = random.Random(404)
    for profile in ("dna", "iupac"):
        pats = [bytes(rng.choice(b"ACGT") for _ in range(m)) for m in (12, 20, 20, 33)]
        if profile == "iupac":
            p = bytearray(pats[1]); p[4] = ord("N"); p[10] = ord("R"); pats[1] = bytes(p)
        texts = []
        for t in range(40):
            n = rng.choice([0, 1, 5, 19, 20, 21, 40, 63, 64, 65, 100, 300, 1000, 2500])
            text = bytearray(rng.choice(b"ACGT") for _ in range(n))
            for p in pats:
                plain = bytes(c if c in b"ACGT" else 65 for c in p)
                if n >= len(p) + 5 and rng.random() < 0.7:
                    ins = mutate(rng, plain, rng.randrange(0, 3))
                    at = rng.randrange(0, n - len(ins))
                    text[at:at + len(ins)] = ins
            if n >= 40:
                p = bytes(c if c in b"ACGT" else 65 for c in pats[t % 4])
                kind = t % 4
                if kind == 0:    # the text ends inside a match: pattern minus its last 1..2 chars
                    cut = p[:len(p) - 1 - (t // 4) % 2]
                    text[n - len(cut):] = cut
                elif kind == 1:  # the text starts inside a match
                    cut = p[1 + (t // 4) % 2:]
                    text[:len(cut)] = cut
                elif kind == 2:  # exact match flush with the end
                    text[n - len(p):] = p
                if profile == "iupac" and rng.random() < 0.5:
                    text[rng.randrange(n)] = ord("N")
            texts.append(bytes(text))
        for rc in (False, True):
            for k in (0, 2):
                s = sassy.Searcher(profile, rc=rc)
                for mode in ("search", "search_all"):
                    allm = mode == "search_all"
                    got = s.search_many(pats, texts, k, all_minima=allm)
                    want = []
                    for pi, p in enumerate(pats):
                        for ti, t in enumerate(texts):
                            for m in oracle.search(profile, p, t, k, rc=rc, all_minima=allm):
                                want.append((pi, ti, m.text_start, m.text_end, m.cost, m.strand, m.cigar))
                    assert [(m.pattern_idx, m.text_idx, m.text_start, m.text_end, m.cost, m.strand, m.cigar)
                            for m in got] == want, (profile, rc, k, mode)
                    assert len(want) > 10
                    # one scan per pattern and strand, not one per (pattern, text) pair
                    assert s.stats()["scan_launches"] == len(pats) * (2 if rc else 1)
                # modes on top of the batched path
                s2 = sassy.Searcher(profile, rc=rc).only_best_match().with_max_n_frac(0.1 if profile == "iupac" else None)
                got = s2.search_many(pats, texts, k)
                want = []
                for pi, p in enumerate(pats):
                    for ti, t in enumerate(texts):
                        for m in oracle.search_modes(profile, p, t, k, rc=rc, only_best=True,
                                                     max_n_frac=0.1 if profile == "iupac" else None):
                            want.append((pi, ti, m.text_start, m.text_end, m.cost, m.strand, m.cigar))
                assert [(m.pattern_idx, m.text_idx, m.text_start, m.text_end, m.cost, m.strand, m.cigar)
                        for m in got] == want, (profile, rc, k, "best")


def test_overhang(sassy, kats):
    """Overhang: the reference's known answers, then seeded fuzz of
    Searcher::<Iupac>::new_{fwd,rc}_with_overhang(alpha)[.with_max_overhang(mo)] against the oracle --
    matches hanging over the text start / end, alpha in {0, 0.25, 0.5, 1}, search and search_all,
    k small (wave-shape traceback) and > 30 (thread-shape traceback), N-fraction filter on top."""
    for e in kats["overhang"]:
        pat, text = e["pattern"].encode(), e["text"].encode()
        s = sassy.Searcher(e["profile"], rc=e["rc"], alpha=e["alpha"])
        if "max_n_frac" in e:
            s.with_max_n_frac(e["max_n_frac"])
        ms = s.search_all(pat, text, e["k"]) if e["mode"] == "search_all" else s.search(pat, text, e["k"])
        assert_same(ms, oracle.search_overhang(e["profile"], pat, text, e["k"], e["alpha"], rc=e["rc"],
                                               all_minima=e["mode"] == "search_all"))
        if "expect" in e:
            for m, x in zip(ms, e["expect"]):
                for f, v in x.items():
                    assert getattr(m, f) == v, (e["id"], f, m)
        if "expect_len" in e:
            assert len(ms) == e["expect_len"]
    rng = random.Random(515)
    for it in range(120):
        m = rng.choice([4, 8, 12, 20, 33, 40, 70])
        k = rng.randrange(0, min(6, m // 2) + 1)
        if it % 40 == 39:
            m, k = 80, 34  # band wider than a wavefront: thread-shape traceback
        alpha = rng.choice([0.0, 0.25, 0.5, 0.5, 1.0])
        mo = rng.choice([None, None, 0, 3, m // 2])
        pat = bytearray(rng.choice(b"ACGT") for _ in range(m))
        if rng.random() < 0.3:
            pat[rng.randrange(m)] = rng.choice(b"NRYW")
        pat = bytes(pat)
        plain = bytes(c if c in b"ACGT" else 65 for c in pat)
        n = rng.choice([3, 10, 40, 63, 64, 65, 130, 500, 2000])
        text = bytearray(rng.choice(b"ACGTN") if rng.random() < 0.05 else rng.choice(b"ACGT") for _ in range(n))
        # a suffix of the pattern at the text start, a prefix at the text end, something in the middle
        cut = rng.randrange(1, m)
        head = mutate(rng, plain, rng.randrange(0, 2))[cut:][:n]
        text[:len(head)] = head
        cut = rng.randrange(1, m)
        tail = mutate(rng, plain, rng.randrange(0, 2))[:cut][-n:]
        text[n - len(tail):] = tail
        if n > 3 * m:
            mid = mutate(rng, plain, rng.randrange(0, k + 1))
            at = rng.randrange(m, n - 2 * m)
            text[at:at + len(mid)] = mid
        tb = bytes(text)
        rc = bool(it & 1)
        allm = bool(it & 2)
        s = sassy.Searcher("iupac", rc=rc, alpha=alpha).with_max_overhang(mo)
        got = s.search_all(pat, tb, k) if allm else s.search(pat, tb, k)
        want = oracle.search_overhang("iupac", pat, tb, k, alpha, rc=rc, all_minima=allm, max_overhang=mo)
        assert_same(got, want), (it, m, k, alpha, mo, n)


def test_overhang_many_patterns_in_one_pass(sassy):
    """search_many with an overhang searcher and several patterns of one length: one pass per strand over the batch
    (tiled_pertext_kernel: every text from its own overhang column to its last virtual column; the reference's v2 scan takes
    overhang in its tiled loop, src/pattern_tiling/search.rs:222-323) -- against oracle.search_overhang pair by pair and
    against the launch-per-pattern path (SASSY_HIP_OVERHANG_TILED=0): barcodes hanging over read starts and ends, texts
    shorter than a pattern, empty texts, lengths around the block size, alpha in {0, 0.25, 0.5, 1}, max_overhang, both
    strands, search_all, without_trace; then the reference's own overhang vectors, each as a batch of texts."""
    import os
    rng = random.Random(616)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    seeded_cases = 0
    for it in range(14):
        m = rng.choice([8, 16, 24, 24, 32, 40, 64])
        k = rng.randrange(0, min(5, m // 3) + 1)
        alpha = rng.choice([0.0, 0.25, 0.5, 0.5, 1.0])
        mo = rng.choice([None, None, 0, 3, m // 2])
        npat = rng.choice([4, 7, 70])
        pats = []
        for _ in range(npat):
            p_ = bytearray(rand_seq(rng, m))
            if rng.random() < 0.3:
                p_[rng.randrange(m)] = rng.choice(b"NRYW")
            pats.append(bytes(p_))
        texts = []
        for _ in range(rng.choice([3, 40, 300])):
            n = rng.choice([0, 1, 5, m - 1, m, 62, 63, 64, 65, 127, 128, 129, 300, 1000])
            # (every other case a batch of plain bases: the seeded search then lists the inside of the texts, the per-text
            # tiled scan their two edges)
            t = bytearray(rng.choice(b"ACGTN") if (it % 2 and rng.random() < 0.03) else rng.choice(b"ACGT") for _ in range(n))
            if n:
                plain = bytes(c if c in b"ACGT" else 65 for c in rng.choice(pats))
                if rng.random() < 0.5:
                    plain = plain.translate(comp)[::-1]
                cut = rng.randrange(1, m)
                head = mutate(rng, plain, rng.randrange(0, 2))[cut:][:n]
                if rng.random() < 0.6:
                    t[:len(head)] = head
                cut = rng.randrange(1, m)
                tail = mutate(rng, plain, rng.randrange(0, 2))[:cut][-n:]
                if rng.random() < 0.6:
                    t[n - len(tail):] = tail
                if n > 3 * m and rng.random() < 0.7:
                    mid = mutate(rng, plain, rng.randrange(0, k + 1))
                    at = rng.randrange(m, n - 2 * m)
                    t[at:at + len(mid)] = mid
            texts.append(bytes(t))
        rc, allm = bool(it & 1), it % 5 == 4
        want = []
        for pi, p_ in enumerate(pats):
            for ti, t in enumerate(texts):
                for x in oracle.search_overhang("iupac", p_, t, k, alpha, rc=rc, all_minima=allm, max_overhang=mo):
                    want.append((pi, ti) + key(x)[1:])
        keyf = lambda x: (x.pattern_idx, x.text_idx) + key(x)[1:]
        res = {}
        paths = set()
        for env in ("1", "tiled", "0"):
            os.environ["SASSY_HIP_OVERHANG_TILED"] = "0" if env == "0" else "1"
            if env == "tiled":
                os.environ["SASSY_HIP_OVERHANG_SEEDED"] = "0"
            s = sassy.Searcher("iupac", rc=rc, alpha=alpha).with_max_overhang(mo)
            got = s.search_many(pats, texts, k, all_minima=allm)
            os.environ.pop("SASSY_HIP_OVERHANG_SEEDED", None)
            res[env] = [keyf(x) for x in got]
            assert sorted(res[env]) == sorted(want), (it, env, m, k, alpha, mo, npat, len(texts), rc, allm, len(got), len(want))
            if env != "0" and len(texts) >= 2:
                assert s.stats()["filtered"] in ((5, 6) if env == "1" else (5,)), s.stats()   # one pass took the batch
                paths.add(s.stats()["filtered"])
        os.environ.pop("SASSY_HIP_OVERHANG_TILED")
        seeded_cases = seeded_cases + (1 if 6 in paths else 0)
        # pattern-major, text by text, the forward strand's records in front of the Rc strand's -- as the other path orders them
        assert [x[:2] for x in res["1"]] == [x[:2] for x in res["0"]], (it,)
    assert seeded_cases >= 2, seeded_cases   # (the seeded search + edge segments took some of the plain batches)
    # the reference's overhang vectors: each text between decoys, four copies of the pattern (a batch the one pass takes)
    for e in kats_overhang():
        pat, text = e["pattern"].encode(), e["text"].encode()
        if len(pat) > 64:
            continue
        s = sassy.Searcher(e["profile"], rc=e["rc"], alpha=e["alpha"])
        texts = [b"ACGTACGTAC", text, b"", text[::-1], text]
        got = s.search_many([pat] * 4, texts, e["k"], all_minima=e["mode"] == "search_all")
        want = []
        for pi in range(4):
            for ti, t in enumerate(texts):
                for x in oracle.search_overhang(e["profile"], pat, t, e["k"], e["alpha"], rc=e["rc"], all_minima=e["mode"] == "search_all"):
                    want.append((pi, ti) + key(x)[1:])
        assert sorted((x.pattern_idx, x.text_idx) + key(x)[1:] for x in got) == sorted(want), e["id"]


@pytest.mark.gpu
def test_overhang_encoded_patterns_on_one_long_text(sassy):
    """search_encoded_patterns of an overhang searcher on ONE long text: one pass (the text as a batch of one: the seeded search
    for the inside, tiled_pertext_kernel's edge segments for [0, m + k] and the virtual columns; the reference's v2 scan takes
    overhang in its tiled loop, src/pattern_tiling/search.rs:222-323) -- against oracle.search_overhang pattern by pattern
    (forward searchers) and against the chain-per-pattern path (switch overhang_seeded = 0; both strands too).  Patterns
    hanging over the text's start and end, planted inside, host and device-resident texts of odd lengths."""
    rng = random.Random(717)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for it in range(8):
        m = rng.choice([20, 24, 32])
        k = rng.randrange(1, 4)
        alpha = rng.choice([0.25, 0.5, 0.5, 1.0])
        mo = rng.choice([None, None, 5])
        npat = rng.choice([4, 9, 70])
        pats = []
        for _ in range(npat):
            p_ = bytearray(rand_seq(rng, m))
            if rng.random() < 0.2:
                p_[rng.randrange(m)] = rng.choice(b"NRYW")
            pats.append(bytes(p_))
        n = rng.choice([3_001, 70_001, 200_003])
        text = bytearray(rand_seq(rng, n))
        for _ in range(12):
            plain = bytes(c if c in b"ACGT" else 65 for c in rng.choice(pats))
            if rng.random() < 0.4:
                plain = plain.translate(comp)[::-1]
            ins = mutate(rng, plain, rng.randrange(0, k + 1))
            at = rng.randrange(2 * m, n - 3 * m)
            text[at:at + len(ins)] = ins
        for side in (0, 1):  # a pattern hanging over the text's start / end
            plain = bytes(c if c in b"ACGT" else 65 for c in rng.choice(pats))
            cut = rng.randrange(2, m - 2)
            if side == 0:
                head = mutate(rng, plain, rng.randrange(0, 2))[cut:]
                text[:len(head)] = head
            else:
                tail = mutate(rng, plain, rng.randrange(0, 2))[:cut]
                text[n - len(tail):] = tail
        text = bytes(text[:n])
        rc, allm = bool(it & 1), it % 4 == 3
        keyp = lambda x: (x.pattern_idx,) + key(x)[1:]
        s = sassy.Searcher("iupac", rc=rc, alpha=alpha).with_max_overhang(mo)
        enc = s.encode_patterns(pats)
        if it % 3 == 2:  # device-resident, the buffer as long as the text
            buf = sassy.DeviceBuffer(n)
            buf.upload(text)
            got = s.search_encoded_patterns(enc, _DevText(buf.ptr, n), k, all_minima=allm)
        else:
            got = s.search_encoded_patterns(enc, text, k, all_minima=allm)
        assert s.stats()["filtered"] == 6, s.stats()  # the one pass took it
        s0 = sassy.Searcher("iupac", rc=rc, alpha=alpha).with_max_overhang(mo)
        s0.set_option("overhang_seeded", 0)
        got0 = s0.search_encoded_patterns(s0.encode_patterns(pats), text, k, all_minima=allm)
        assert s0.stats()["filtered"] != 6
        assert sorted(keyp(x) for x in got) == sorted(keyp(x) for x in got0), (it, m, k, alpha, mo, npat, n, rc, allm, len(got), len(got0))
        if not rc:
            want = []
            for pi, p_ in enumerate(pats):
                for x in oracle.search_overhang("iupac", p_, text, k, alpha, all_minima=allm, max_overhang=mo):
                    want.append((pi,) + key(x)[1:])
            assert sorted(keyp(x) for x in got) == sorted(want), (it, m, k, alpha, mo, npat, n, allm, len(got), len(want))


def kats_overhang():
    import json
    root = os.path.dirname(os.path.abspath(__file__))
    return [e for e in json.load(open(os.path.join(root, "golden", "kats.json")))["overhang"] if "max_n_frac" not in e]


def test_encoded_many_patterns(sassy):
    """search_encoded_patterns with many plain-ACGT patterns on an Iupac searcher (BASELINE config 4
    shape): on plain-ACGT text the scans run with the Dna kernels, on text with other letters with the
    Iupac kernels -- both must equal the oracle's search_encoded."""
    import os
    rng = random.Random(45)
    pats = [bytes(rng.choice(b"ACGT") for _ in range(20)) for _ in range(12)]
    os.environ["SASSY_HIP_TILED"] = "0"  # one scan per pattern here; the one-pass paths: test_encoded_pattern_tiled,
    os.environ["SASSY_HIP_SEEDED"] = "0"  # test_encoded_seeded
    for variant in ("plain", "lower", "with_n", "plain_multi", "lower_multi"):
        # *_multi: force the multi-pattern prefilter (one filter_dna_multi_kernel pass per batch of
        # patterns) that long texts get by default
        if variant.endswith("_multi"):
            os.environ["SASSY_HIP_MULTI_MIN_TEXT"] = "1"
        else:
            os.environ.pop("SASSY_HIP_MULTI_MIN_TEXT", None)
        n = 30_000 + (7 if variant.startswith("lower") else 0)  # also a length that is not a multiple of 16
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for p in pats:
            for _ in range(2):
                ins = mutate(rng, p, rng.randrange(0, 3))
                if rng.random() < 0.5:
                    ins = oracle.reverse_complement("iupac", ins)
                at = rng.randrange(0, n - len(ins))
                text[at:at + len(ins)] = ins
        if variant.startswith("lower"):
            for _ in range(300):
                i = rng.randrange(n); text[i] = text[i] | 0x20
        if variant == "with_n":
            for _ in range(50):
                text[rng.randrange(n)] = rng.choice(b"NRY")
            text[n - 1] = ord("N")  # in the tail bytes behind the last full 16-byte chunk
        tb = bytes(text)
        for rc in (False, True):
            s = sassy.Searcher("iupac", rc=rc)
            enc = s.encode_patterns(pats)
            got = s.search_encoded_patterns(enc, tb, 2)
            want = oracle.search_encoded("iupac", pats, tb, 2, rc=rc)
            assert len(want) >= 6
            assert sorted(key(m) for m in got) == sorted(key(m) for m in want), (variant, rc)
            if variant.endswith("_multi"):
                assert s.stats()["filtered"] == 2  # chunk lists from the multi-pattern prefilter's bitmaps
            elif not os.environ.get("SASSY_HIP_PREFILTER"):
                # 20-mers at k=2: pieces too short for the plain bit-plane filter -- the paired filter's fused launch (round 5),
                # else q-gram counting or the streaming DP
                st_ = s.stats()
                assert st_["filtered"] in (0, 4) or (st_["filtered"] == 2 and st_["pair"] == 2), st_
    os.environ.pop("SASSY_HIP_MULTI_MIN_TEXT", None)
    os.environ.pop("SASSY_HIP_TILED", None)
    os.environ.pop("SASSY_HIP_SEEDED", None)


def test_encoded_pattern_tiled(sassy):
    """search_encoded_patterns through the pattern-tiled scan (tiled_kernel.hip, the reference's v2 shape: one
    pattern per lane, all patterns in one pass; src/pattern_tiling/search.rs:326-425) against the oracle:
    both profiles, both strands, every word shape (m <= 32, m <= 64), k = 0, m <= k, texts shorter than the
    pattern, IUPAC letters in patterns and text, search_all, without_trace, more than 64 patterns (several
    groups), chunk seams (texts longer than one wave's chunk)."""
    import os
    rng = random.Random(4711)
    os.environ.pop("SASSY_HIP_TILED", None)
    os.environ["SASSY_HIP_SEEDED"] = "0"  # (the library would seed some of these shapes by itself)
    shapes = [  # (profile, m, k, npat, n, alphabet of the text, all_minima)
        ("dna", 20, 2, 12, 30_000, b"ACGT", False),
        ("dna", 20, 2, 150, 5_000, b"ACGT", False),
        ("dna", 32, 3, 70, 20_000, b"ACGT", False),
        ("dna", 33, 4, 9, 20_000, b"ACGT", False),
        ("dna", 64, 8, 65, 9_000, b"ACGT", False),
        ("dna", 5, 0, 3, 4_000, b"ACGT", False),
        ("dna", 3, 3, 4, 300, b"ACGT", False),       # m <= k: every position is a report candidate
        ("dna", 4, 6, 2, 200, b"ACGT", True),
        ("dna", 12, 2, 5, 7, b"ACGT", False),        # text shorter than the patterns
        ("dna", 16, 1, 200, 3_000, b"ACGTN", True),
        ("iupac", 20, 2, 12, 30_000, b"ACGT", False),
        ("iupac", 20, 3, 100, 8_000, b"ACGTNRYacgtn-", False),
        ("iupac", 24, 2, 40, 6_000, b"ACGTN", True),
        ("iupac", 64, 12, 7, 4_000, b"ACGTRYKM", False),
        ("iupac", 1, 0, 3, 500, b"ACGTN", False),
        ("dna", 24, 3, 20, 600_000, b"ACGT", False),  # many chunks per group of patterns
    ]
    for (profile, m, k, npat, n, alpha, allm) in shapes:
        pal = b"ACGT" if profile == "dna" else b"ACGTNRYSWKM"
        pats = [bytes(rng.choice(pal if rng.random() < 0.3 else b"ACGT") for _ in range(m)) for _ in range(npat)]
        text = bytearray(rng.choice(alpha) for _ in range(n))
        for p in pats[:40]:
            for _ in range(2):
                ins = mutate(rng, p, rng.randrange(0, k + 1))
                if profile == "iupac":
                    ins = bytes(c if chr(c) in "ACGT" else rng.choice(b"ACGT") for c in ins)
                if rng.random() < 0.5:
                    ins = oracle.reverse_complement("iupac", ins)
                if len(ins) < n:
                    at = rng.randrange(0, n - len(ins))
                    text[at:at + len(ins)] = ins
        if n > 100:
            ins = mutate(rng, pats[0], 0)  # a match that ends at the text end, one at its start
            text[n - len(ins):] = ins
            text[:m] = pats[-1]
        tb = bytes(text)
        for rc in (False, True):
            s = sassy.Searcher(profile, rc=rc)
            enc = s.encode_patterns(pats)
            got = s.search_encoded_patterns(enc, tb, k, all_minima=allm)
            st = s.stats()
            want = oracle.search_encoded(profile, pats, tb, k, rc=rc, all_minima=allm)
            assert st["filtered"] == 5, (profile, m, k, npat, n, st)  # the pattern-tiled scan took the call
            assert sorted(key(x) for x in got) == sorted(key(x) for x in want), (profile, m, k, npat, n, rc, allm)
            assert len(want) >= (1 if n > 100 else 0)
            got_wo = s.search_encoded_patterns(enc, tb, k, all_minima=allm, without_trace=True)
            assert sorted((x.pattern_idx, x.text_end, x.cost, x.strand) for x in got_wo) == \
                sorted((x.pattern_idx, x.text_end, x.cost, x.strand) for x in want)
    # a device-resident text that does not start on a 64-byte boundary (the kernel loads aligned blocks and
    # skips the bytes in front of the text; the bytes around it are poison that would match)
    pats = [bytes(rng.choice(b"ACGT") for _ in range(20)) for _ in range(70)]
    n = 5_000
    for off in (16, 48):
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        text[:20] = pats[3]
        text[n - 20:] = pats[69]
        text[1000:1020] = pats[5]
        buf = sassy.DeviceBuffer(n + 256)
        buf.upload(pats[3] * 3 + pats[3][:4], 0)           # in front of the text: more of pattern 3
        buf.upload(bytes(text), off)
        buf.upload(pats[69] * 3, off + n)                   # behind it: more of pattern 69
        s = sassy.Searcher("dna", rc=False)
        got = s.search_encoded_patterns(s.encode_patterns(pats), _DevText(buf.ptr + off, n), 1)
        assert s.stats()["filtered"] == 5
        want = oracle.search_encoded("dna", pats, bytes(text), 1)
        assert sorted(key(x) for x in got) == sorted(key(x) for x in want) and len(want) >= 3, off
    # the searcher's report filters are applied per pattern: equal to the one-scan-per-pattern path
    _encoded_filters_agree(sassy, rng, "SASSY_HIP_TILED", 5)
    os.environ.pop("SASSY_HIP_SEEDED", None)


def test_seeded_test_geometry_sweep(sassy):
    """The sub-piece test in front of the seeded search's verification reads ONE text window for all pieces, laid out from
    the pattern's shape (host.hip: reach, win_left, offsets within 48 characters, lengths capped by the shifts;
    seed_kernels.hip: test_issue / test_finish).  A wrong offset shows as a LOST match, so: every pattern length 8 .. 32
    with every k the path takes, matches planted with exactly k edits (substitutions, insertions and deletions at random
    rows: next to the seed, at sub-piece borders, at the pattern's ends), plain patterns and patterns with ambiguity
    letters (care words), against the oracle."""
    import os
    rng = random.Random(4242)
    os.environ["SASSY_HIP_SEEDED"] = "1"
    os.environ.pop("SASSY_HIP_TILED", None)
    seeded = 0
    try:
        for m in range(8, 33):
            for k in range(0, 8):
                if m // (k + 1) < 4 or k * 4 > m:
                    continue
                for profile in ("dna", "iupac"):
                    npat, n = 24, 24_000
                    pats = []
                    for _ in range(npat):
                        p = bytearray(rng.choice(b"ACGT") for _ in range(m))
                        if profile == "iupac" and rng.random() < 0.5:
                            for _ in range(rng.randrange(1, 3)):
                                p[rng.randrange(m)] = rng.choice(b"NRYKMSW")
                        pats.append(bytes(p))
                    text = bytearray(rng.choice(b"ACGT") for _ in range(n))
                    at = 64
                    for p in pats:
                        conc = bytes(c if c in b"ACGT" else rng.choice(b"ACGT") for c in p)
                        for e_ in (k, k, max(0, k - 1), k):
                            ins = mutate(rng, conc, e_)
                            text[at:at + len(ins)] = ins
                            at += len(ins) + rng.randrange(m + 2 * k + 2, 3 * m + 40)
                    assert at < n
                    tb = bytes(text)
                    s = sassy.Searcher(profile, rc=False)
                    enc = s.encode_patterns(pats)
                    got = s.search_encoded_patterns(enc, tb, k)
                    seeded += s.stats()["filtered"] == 6
                    want = oracle.search_encoded(profile, pats, tb, k, rc=False)
                    assert sorted(key(x) for x in got) == sorted(key(x) for x in want), (profile, m, k, len(got), len(want))
                    assert len(want) >= 2 * npat
    finally:
        os.environ.pop("SASSY_HIP_SEEDED", None)
    assert seeded >= 60, seeded


def test_search_many_takes_a_text_batch(sassy):
    """search_many with the texts as ONE buffer + offsets (sassy_amd.TextBatch: what a FASTA / FASTQ reader holds) returns
    what the list of bytes returns -- empty texts, texts shorter than the patterns, an unused gap in the buffer."""
    rng = random.Random(77)
    pats = [bytes(rng.choice(b"ACGT") for _ in range(24)) for _ in range(8)]
    texts = []
    for i in range(300):
        t = bytearray(rng.choice(b"ACGT") for _ in range(rng.choice([0, 5, 200, 700])))
        if len(t) > 100:
            ins = mutate(rng, pats[i % len(pats)], rng.randrange(0, 4))
            at = rng.randrange(0, len(t) - len(ins))
            t[at:at + len(ins)] = ins
        texts.append(bytes(t))
    s = sassy.Searcher("iupac", rc=True)
    want = s.search_many(pats, texts, 3)
    got = s.search_many(pats, sassy.TextBatch.from_list(texts), 3)
    assert [key(x) + (x.text_idx,) for x in got] == [key(x) + (x.text_idx,) for x in want] and len(want) >= 100
    import numpy as np
    buf = b"#" * 17 + b"".join(texts)  # (a header in front: starts need not begin at 0)
    lens = np.array([len(t) for t in texts], dtype=np.uint64)
    starts = 17 + np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    got2 = s.search_many(pats, sassy.TextBatch(buf, starts, lens), 3)
    assert [key(x) + (x.text_idx,) for x in got2] == [key(x) + (x.text_idx,) for x in want]
    with pytest.raises(sassy.SassyHipError):
        sassy.TextBatch(b"ACGT", [2], [5])


def _encoded_filters_agree(sassy, rng, env, kind):
    import os
    pats = [bytes(rng.choice(b"ACGT") for _ in range(20)) for _ in range(30)]
    # (the seeded search is for Dna codes: no other letters, or the reference's traceback panics)
    text = bytearray(rng.choice(b"ACGTN" if kind == 5 else b"ACGT") for _ in range(20_000))
    for p in pats:
        for _ in range(3):
            ins = mutate(rng, p, rng.randrange(0, 3))
            at = rng.randrange(0, len(text) - len(ins))
            text[at:at + len(ins)] = ins
    tb = bytes(text)
    for cfg in ("best", "nfrac", "both"):
        res = []
        for on in ("1", "0"):
            os.environ[env] = on
            os.environ["SASSY_HIP_TILED"] = os.environ.get("SASSY_HIP_TILED", "0")
            s = sassy.Searcher("iupac" if kind == 5 else "dna", rc=True)
            if cfg in ("best", "both"):
                s.only_best_match()
            if cfg in ("nfrac", "both"):
                s.with_max_n_frac(0.1)
            enc = s.encode_patterns(pats)
            res.append(sorted(key(x) for x in s.search_encoded_patterns(enc, tb, 3)))
            assert (s.stats()["filtered"] == kind) == (on == "1")
        assert res[0] == res[1] and len(res[0]) >= 10, cfg
    os.environ.pop(env, None)
    os.environ.pop("SASSY_HIP_TILED", None)


def test_encoded_seeded(sassy):
    """search_encoded_patterns through seed -> verify -> report (seed_kernels.hip: one pass over the text looks every
    L-gram up in a table of all patterns' pigeonhole pieces, one lane per hit runs the pattern around it) against
    the oracle.  Shapes: both word widths, seeds of one and of two lengths, seeds cut to 10 rows, 1 .. 7 pieces,
    matches at both ends of the text, texts shorter than a pattern, lower case, the same pattern twice, both
    strands, search_all, without_trace, a candidate list that overflows (segments are cut smaller and repeated)."""
    import os
    rng = random.Random(99)
    os.environ["SASSY_HIP_SEEDED"] = "1"
    os.environ.pop("SASSY_HIP_TILED", None)
    shapes = [  # (searcher profile, m, k, npat, n, all_minima)
        ("dna", 20, 2, 300, 40_000, False),
        ("iupac", 20, 2, 70, 30_000, False),   # plain ACGT patterns and text: the Dna path after the text check
        ("dna", 32, 3, 100, 50_000, False),
        ("dna", 24, 1, 64, 30_000, True),
        ("dna", 40, 2, 30, 20_000, False),      # 64-bit pattern words
        ("dna", 12, 0, 40, 20_000, False),      # one piece of 12 rows: the seed is its last 10
        ("dna", 35, 6, 10, 3_000, False),       # seven pieces of 5 rows: many hits
        ("dna", 60, 1, 9, 8_000, True),         # the longest window: m + 3k + 1 = 64 characters
        ("dna", 20, 2, 5, 7, False),            # text shorter than the patterns
        ("dna", 20, 2, 12, 45, False),          # every window leaves the text
        ("dna", 23, 3, 3000, 6_000, False),
    ]
    for (profile, m, k, npat, n, allm) in shapes:
        pats = [bytes(rng.choice(b"ACGT") for _ in range(m)) for _ in range(npat)]
        pats[-1] = pats[0]  # the same pattern twice: two entries under every seed
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for p in pats[:60]:
            for _ in range(2):
                ins = mutate(rng, p, rng.randrange(0, k + 1))
                if rng.random() < 0.5:
                    ins = oracle.reverse_complement("iupac", ins)
                if len(ins) < n:
                    at = rng.randrange(0, n - len(ins))
                    text[at:at + len(ins)] = ins
        if n > 100:
            text[n - m:] = pats[1]
            text[:m] = pats[2]
            for _ in range(50):
                i = rng.randrange(n); text[i] |= 0x20
        tb = bytes(text)
        for rc in (False, True):
            s = sassy.Searcher(profile, rc=rc)
            enc = s.encode_patterns(pats)
            got = s.search_encoded_patterns(enc, tb, k, all_minima=allm)
            st = s.stats()
            want = oracle.search_encoded(profile, pats, tb, k, rc=rc, all_minima=allm)
            if n >= 16:
                assert st["filtered"] == 6, (profile, m, k, npat, n, st)
            assert sorted(key(x) for x in got) == sorted(key(x) for x in want), (profile, m, k, npat, n, rc, allm)
            assert len(want) >= (3 if n > 100 else 0)
            got_wo = s.search_encoded_patterns(enc, tb, k, all_minima=allm, without_trace=True)
            assert sorted((x.pattern_idx, x.text_end, x.cost, x.strand) for x in got_wo) == \
                sorted((x.pattern_idx, x.text_end, x.cost, x.strand) for x in want)
    # patterns with ambiguity letters on a plain-ACGT text (Iupac searcher): CRISPR guides with their NGG, letters that
    # stand for two or three bases inside seeds and sub-pieces, an X (matches nothing); a seed with such letters has one
    # table entry per concrete string it matches
    for (m, k, npat, n) in [(23, 3, 120, 60_000), (20, 2, 40, 30_000), (32, 3, 20, 30_000)]:
        pats = []
        for i in range(npat):
            p = bytearray(rng.choice(b"ACGT") for _ in range(m))
            if m == 23:
                p[20:23] = b"NGG"
            else:
                for _ in range(rng.randrange(0, 4)):
                    p[rng.randrange(m)] = rng.choice(b"NRYKMSWBDHV")
            pats.append(bytes(p))
        pats[1] = pats[1][:5] + b"X" + pats[1][6:]
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for p in pats[:60]:
            conc = bytes(c if c in b"ACGT" else rng.choice(b"ACGT") for c in p)
            for _ in range(2):
                ins = mutate(rng, conc, rng.randrange(0, k + 1))
                if rng.random() < 0.5:
                    ins = oracle.reverse_complement("iupac", ins)
                at = rng.randrange(0, n - len(ins))
                text[at:at + len(ins)] = ins
        tb = bytes(text)
        for rc in (False, True):
            s = sassy.Searcher("iupac", rc=rc)
            enc = s.encode_patterns(pats)
            got = s.search_encoded_patterns(enc, tb, k)
            assert s.stats()["filtered"] == 6, (m, k, s.stats())
            want = oracle.search_encoded("iupac", pats, tb, k, rc=rc)
            assert sorted(key(x) for x in got) == sorted(key(x) for x in want), (m, k, npat, rc, len(got), len(want))
            assert len(want) >= 10
    # texts with other letters than ACGT (Iupac searcher): the seeded pass is exact where the window in front of an
    # end position is plain; around every run of other letters the pattern-tiled scan fills in, and a long run of N
    # is left out of its copy (every cost is constant inside) -- runs of all lengths around that threshold
    # m + 2, at both ends of the text, close to each other, with a letter inside that is not a wildcard
    for (m, k, npat, allm) in [(20, 2, 60, False), (23, 3, 40, False), (20, 2, 30, True), (32, 3, 25, False)]:
        C = m + k
        pats = [bytes(rng.choice(b"ACGT") for _ in range(m)) for _ in range(npat)]
        pats[0] = pats[0][:4] + b"N" + pats[0][5:]
        pats[1] = pats[1][:9] + b"X" + pats[1][10:]
        n = 40_000
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for p in pats:
            conc = bytes(c if c in b"ACGT" else 65 for c in p)
            for _ in range(3):
                ins = mutate(rng, conc, rng.randrange(0, k + 1))
                at = rng.randrange(0, n - len(ins))
                text[at:at + len(ins)] = ins
        at = 0
        for ln in [5, 1, m, m + 1, m + 2, m + 3, 2 * C + 2, 3 * C, 500, 4000, 7, m + 2]:
            text[at:at + ln] = b"N" * ln
            at += ln + rng.choice([1, 3, C - 1, C, C + 1, 2 * C, 900, 2500])
        text[at + 100:at + 100 + 300] = b"N" * 300
        text[at + 250] = ord("R")            # not a wildcard: this run is copied whole
        for _ in range(40):
            text[rng.randrange(n)] = rng.choice(b"NRYKMSWn-*x")
        text[n - 3 * C:] = b"N" * (3 * C)    # the text ends inside a run
        for p in pats[:10]:                  # matches that lean into runs
            conc = bytes(c if c in b"ACGT" else 65 for c in p)
            i = bytes(text).find(b"N" * 50)
            text[i - m // 2:i] = conc[:m // 2]
        tb = bytes(text)
        for rc in (False, True):
            s = sassy.Searcher("iupac", rc=rc)
            enc = s.encode_patterns(pats)
            got = s.search_encoded_patterns(enc, tb, k, all_minima=allm)
            st = s.stats()
            assert st["filtered"] == 6 and st["cond_resolved"] >= 10, (m, k, st)
            want = oracle.search_encoded("iupac", pats, tb, k, rc=rc, all_minima=allm)
            assert sorted(key(x) for x in got) == sorted(key(x) for x in want), (m, k, npat, rc, allm, len(got), len(want))
            assert len(want) >= 100
    # a text of one repeated unit: every position hits the seed tables of the patterns cut from it; the candidate
    # list overflows its expectation-sized capacity and the segments are cut smaller
    unit = bytes(rng.choice(b"ACGT") for _ in range(37))
    tb = (unit * 30_000)[:1_000_000]
    pats = [tb[i:i + 20] for i in range(0, 37)] * 3 + [bytes(rng.choice(b"ACGT") for _ in range(20)) for _ in range(20)]
    pats = [bytes(mutate(rng, p, rng.randrange(0, 2))[:20].ljust(20, b"A")) for p in pats]
    s = sassy.Searcher("dna", rc=False)
    enc = s.encode_patterns(pats)
    r = s.search_encoded_patterns(enc, tb, 1, as_result=True)
    assert s.stats()["filtered"] in (6, 5, 0, 2, 4)  # (the library may give up on seeding this text)
    sub = tb[:20_000]
    got = [x for x in s.search_encoded_patterns(enc, sub, 1)]
    want = oracle.search_encoded("dna", pats, sub, 1)
    assert sorted(key(x) for x in got) == sorted(key(x) for x in want) and len(want) > 1000
    assert len(r) > 50 * len(want) * 0.5
    _encoded_filters_agree(sassy, rng, "SASSY_HIP_SEEDED", 6)


def test_encoded_dense_results_leave_in_a_pinned_block(sassy):
    """search_encoded_patterns with 10^3 .. 10^4 matches per call (a guide set on a genome makes 10^7): the rows get their
    pattern index and strand on the device and leave with their cigars in one pinned block the result keeps
    (finish_pattern_list) -- record by record equal to the host's way (SASSY_HIP_ENCODED_PIN=0) and, sorted, to the
    oracle; seeded search and pattern-tiled scan, forward and both strands, guides with their NGG on a text with N."""
    import os
    rng = random.Random(131)
    cases = [("dna", 20, 2, 50, 120_000, False), ("iupac", 23, 3, 40, 150_000, True), ("iupac", 20, 2, 30, 80_000, True)]
    for (profile, m, k, npat, n, ngg) in cases:
        pats = [rand_seq(rng, m - 3) + b"NGG" if ngg else rand_seq(rng, m) for _ in range(npat)]
        text = bytearray(rand_seq(rng, n))
        for _ in range(n // 12):
            p_ = rng.choice(pats)
            ins = mutate(rng, bytes(c if c in b"ACGT" else 71 for c in p_), rng.randrange(0, k + 1))
            if rng.random() < 0.5:
                ins = oracle.reverse_complement("iupac", ins)
            at = rng.randrange(0, n - len(ins))
            text[at:at + len(ins)] = ins
        if profile == "iupac":
            for _ in range(40):
                text[rng.randrange(n)] = rng.choice(b"NRYn")
        tb = bytes(text)
        for rc in (False, True):
            want = sorted(key(x) for x in oracle.search_encoded(profile, pats, tb, k, rc=rc))
            assert len(want) >= 1100, len(want)
            for env in ({"SASSY_HIP_SEEDED": "1"}, {"SASSY_HIP_SEEDED": "0", "SASSY_HIP_TILED": "1"}):
                res = []
                for pin in ("1", "0", "threads"):
                    os.environ.update(env)
                    os.environ["SASSY_HIP_ENCODED_PIN"] = "1" if pin == "threads" else pin
                    if pin == "threads":  # the thread-per-report traceback (dense lists: from 65 536 reports on) with a pattern per report
                        os.environ["SASSY_HIP_ENCODED_TRACE_THREADS"] = "512"
                    s = sassy.Searcher(profile, rc=rc)
                    r = s.search_encoded_patterns(s.encode_patterns(pats), tb, k, as_result=True)
                    for k_ in list(env) + ["SASSY_HIP_ENCODED_PIN", "SASSY_HIP_ENCODED_TRACE_THREADS"]:
                        os.environ.pop(k_, None)
                    res.append(r)
                a, b, c = res
                assert canon(a) == canon(c), (profile, rc, env, "thread-per-report traceback")
                assert len(a) == len(b) == len(want), (profile, rc, env, len(a), len(b), len(want))
                # the same records in the same order (the strings' places in the two pools differ: compared per record)
                for f in a.array.dtype.names:
                    if f != "cigar_off":
                        assert (a.array[f] == b.array[f]).all(), (profile, rc, env, f)
                assert canon(a)[1] == canon(b)[1], (profile, rc, env)
                assert sorted(key(x) for x in a.matches) == want, (profile, rc, env)


def test_encoded_patterns_on_long_plateaus(sassy):
    """The report rule on the sorted list of all end positions (sort_kernels.hip): a plateau of 300 000 equal costs
    (poly-A text, patterns cut from it or one edit away) has ONE report, decided by the entry in front of the plateau
    -- found by a scan over the list, not by a thread that walks it.  Seeded and pattern-tiled search, both strands,
    against the oracle."""
    import os
    rng = random.Random(17)
    flank = lambda n: bytes(rng.choice(b"ACGT") for _ in range(n))
    tb = flank(3_000) + b"A" * 300_000 + flank(2_000) + b"AC" * 40_000 + flank(3_000) + b"T" * 50_000
    pats = [b"A" * 20, b"A" * 19 + b"C", b"C" + b"A" * 19, b"AC" * 10, b"CA" * 10, b"A" * 10 + b"G" + b"A" * 9,
            b"T" * 20, flank(20), flank(20)]
    try:
        for env in ("SASSY_HIP_SEEDED", "SASSY_HIP_TILED"):
            os.environ.pop("SASSY_HIP_SEEDED", None)
            os.environ.pop("SASSY_HIP_TILED", None)
            os.environ[env] = "1"
            for rc in (False, True):
                for allm in (False, True):
                    if allm and rc:
                        continue
                    s = sassy.Searcher("dna", rc=rc)
                    enc = s.encode_patterns(pats)
                    got = s.search_encoded_patterns(enc, tb, 2, all_minima=allm, without_trace=True)
                    want = oracle.search_encoded("dna", pats, tb, 2, rc=rc, all_minima=allm)
                    a = sorted((x.pattern_idx, x.text_end, x.cost, x.strand) for x in got)
                    b = sorted((x.pattern_idx, x.text_end, x.cost, x.strand) for x in want)
                    assert a == b, (env, rc, allm, len(a), len(b))
                    assert len(b) >= (300_000 if allm else 10)
    finally:
        os.environ.pop("SASSY_HIP_SEEDED", None)
        os.environ.pop("SASSY_HIP_TILED", None)


def test_dense_search_then_long_cigars_on_one_searcher(sassy):
    """A searcher keeps its device buffers across calls: after a search with millions of reports (the report buffer
    grows with them) a search whose cigars are long must not size its cigar pool by that capacity (2.1 M reports
    x 2 208 B of cigar text would pass the pool's 4 GiB and used to fail with EUNSUPPORTED)."""
    rng = random.Random(5)
    s = sassy.Searcher("dna", rc=False)
    dense = b"A" * 2_100_000
    got = s.search_all(b"AAAAAAAA", dense, 0)
    assert len(got) == len(dense) - 7
    m, k = 1000, 100
    pat = rand_seq(rng, m)
    text = bytearray(rand_seq(rng, 60_000))
    ins = mutate(rng, pat, 40)
    text[20_000:20_000 + len(ins)] = ins
    got = s.search(pat, bytes(text), k)
    want = oracle.search("dna", pat, bytes(text), k)
    assert_same(got, want)
    assert len(want) >= 1


def test_pack_result_for_gather(sassy):
    """multigpu.pack_result (vectorised wire format of the match gather) against the per-match packer."""
    from sassy_amd import multigpu
    rng = random.Random(12)
    pat = bytes(rng.choice(b"ACGT") for _ in range(32))
    n = 200_000
    text = bytearray(rng.choice(b"ACGT") for _ in range(n))
    for q in range(40):
        ins = mutate(rng, pat, q % 4)
        text[1000 + q * 4000:1000 + q * 4000 + len(ins)] = ins
    buf = sassy.DeviceBuffer(n + 64)
    buf.upload(bytes(text))
    s = sassy.Searcher("dna", rc=False)
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
    assert len(r) >= 40
    packed = multigpu.pack_result(r)
    assert (packed.rows == multigpu.rows_from_matches(r.matches)).all()
    back = multigpu.matches_from_rows(packed.rows, sassy.Match)
    assert [key(m) for m in back] == [key(m) for m in r.matches]


def test_per_text_lanes(sassy):
    """search_many in "one lane per text" mode (host.hip: search_many_pertext): overhang searches,
    the Ascii profile and unfilterable patterns over many short texts must equal the pair-by-pair
    searches -- incl. overhang at every text's own start and end."""
    rng = random.Random(808)
    # ---- overhang over many reads ----
    pats = [bytes(rng.choice(b"ACGT") for _ in range(m)) for m in (10, 16, 24)]
    texts = []
    for t in range(60):
        n = rng.choice([0, 1, 3, 9, 30, 63, 64, 65, 100, 129, 400, 1500])
        text = bytearray(rng.choice(b"ACGT") for _ in range(n))
        p = pats[t % 3]
        if n >= 8:
            cut = rng.randrange(1, len(p))
            if t % 4 == 0:
                head = p[cut:][:n]; text[:len(head)] = head              # pattern hangs over the start
            elif t % 4 == 1:
                tail = p[:cut][-n:]; text[n - len(tail):] = tail         # ... over the end
            elif n > 2 * len(p):
                ins = mutate(rng, p, rng.randrange(0, 3)); at = rng.randrange(0, n - len(ins)); text[at:at + len(ins)] = ins
        texts.append(bytes(text))
    for alpha, mo in ((0.5, None), (0.25, 4), (1.0, None)):
        for rc in (False, True):
            for allm in (False, True):
                s = sassy.Searcher("iupac", rc=rc, alpha=alpha).with_max_overhang(mo)
                got = s.search_many(pats, texts, 2, all_minima=allm)
                assert s.stats()["scan_launches"] == len(pats) * (2 if rc else 1)
                want = []
                for pi, p in enumerate(pats):
                    for ti, t in enumerate(texts):
                        for m in oracle.search_overhang("iupac", p, t, 2, alpha, rc=rc, all_minima=allm, max_overhang=mo):
                            want.append((pi, ti) + key(m)[1:])
                assert [(m.pattern_idx, m.text_idx) + key(m)[1:] for m in got] == want, (alpha, mo, rc, allm)
                assert len(want) > 20
    # ---- Ascii ----
    words = [b"hello", b"world!", b"needle"]
    docs = []
    for t in range(40):
        n = rng.choice([0, 2, 7, 64, 65, 200, 900])
        doc = bytearray(rng.choice(b"abcdefghij lmnopqrstuvwxyz!") for _ in range(n))
        if n > 20:
            w = words[t % 3]; ins = bytearray(w)
            if t % 2: ins[rng.randrange(len(ins))] = ord("#")
            at = rng.randrange(0, n - len(ins)); doc[at:at + len(ins)] = ins
        docs.append(bytes(doc))
    s = sassy.Searcher("ascii", rc=False)
    got = s.search_many(words, docs, 1)
    want = []
    for pi, p in enumerate(words):
        for ti, t in enumerate(docs):
            for m in oracle.search("ascii", p, t, 1):
                want.append((pi, ti) + key(m)[1:])
    assert [(m.pattern_idx, m.text_idx) + key(m)[1:] for m in got] == want
    assert len(want) > 5 and s.stats()["scan_launches"] == len(words)


def test_config1_shape_1mib(sassy):
    """BASELINE config 1: 'ATCG'x8, k=3, 1 MiB random ACGT (+ plants), Dna, forward only."""
    pat = b"ATCG" * 8
    n = 1 << 20
    text = oracle.generate_dna(42, 0, n)
    oracle.plant_window(42, n, 0, text, pat, 3, stride=1 << 14)
    tb = text.tobytes()
    s = sassy.Searcher("dna", rc=False)
    got = s.search(pat, tb, 3)
    want = oracle.search("dna", pat, tb, 3)
    assert len(want) >= 64
    assert_same(got, want)
    # the reference-shaped CPU port reports the same end positions
    ends, _ = oracle.refstyle_ends("dna", pat, tb, 3)
    assert [(m.text_end, m.cost) for m in got] == ends
    # rc searcher on the same text
    got = sassy.Searcher("dna", rc=True).search(pat, tb, 3)
    assert_same(got, oracle.search("dna", pat, tb, 3, rc=True))


def test_without_trace(sassy):
    s = sassy.Searcher("dna", rc=True)
    pat, text = b"ATCGATCG", b"GGGGATCGATCGTTTT"
    full = s.search(pat, text, 1)
    wt = s.search_without_trace(pat, text, 1)
    assert len(wt) == len(full) == 3
    U = sassy.UINT64_MAX
    # reference: src/search.rs:1464-1475 (fwd) and :868-873 (rc: text_end = usize::MAX)
    assert (wt[0].text_start, wt[0].text_end, wt[0].pattern_start, wt[0].cost, wt[0].cigar) == (U, 12, U, 0, "")
    assert wt[1].strand == "-" and wt[1].text_end == U and wt[1].cost == full[1].cost


def test_ascii_profile(sassy):
    s = sassy.Searcher("ascii", rc=False)
    text = b"the quick brown fox jumps over the lazy dog, The Quick Brown Fox" * 20
    for pat, k in [(b"quick", 0), (b"quick", 1), (b"brwn fox", 2), (b"Quick", 0)]:
        assert_same(s.search(pat, text, k), oracle.search("ascii", pat, text, k), (pat, k))


def test_ascii_many_distinct_bytes(sassy):
    """The reference's Ascii profile has 256 slots (src/profiles/ascii.rs:13-29): ordinary text patterns
    with more than 16 distinct bytes must search, not abort -- here 17 .. 64 distinct pattern bytes
    (32- and 64-slot kernels), one text and many texts (per-text lanes), against the oracle (more than 64 distinct
    bytes: test_ascii_binary_patterns_and_very_long_patterns)."""
    rng = random.Random(31)
    sent = b"The quick brown fox jumps over the lazy dog, 1234567890 times; WHY? (because: it_can!)"
    assert len(set(sent)) > 32
    text = bytearray(rng.choice(b"abcdefghij klmnop") for _ in range(30_000))
    for at, e in ((0, 0), (1000, 2), (7777, 5), (15000, 8), (30_000 - len(sent), 1)):
        ins = mutate(rng, sent, e)[:len(sent)]
        text[at:at + len(ins)] = ins
    text = bytes(text)
    s = sassy.Searcher("ascii", rc=False)
    for pat, k in [(sent[:24], 2), (sent[:40], 4), (sent, 8), (sent[10:70], 0), (bytes(range(60, 124)), 3)]:
        assert 16 < len(set(pat)) <= 64, len(set(pat))
        want = oracle.search("ascii", pat, text, k)
        assert_same(s.search(pat, text, k), want, (pat, k))
        assert_same(s.search_all(pat, text, 2), oracle.search("ascii", pat, text, 2, all_minima=True), (pat, "all"))
    assert len(oracle.search("ascii", sent, text, 8)) >= 4
    # the drop-in symbol too (it aborted for such patterns)
    L = sassy.lib()
    h = L.sassy_searcher(b"ascii", False, float("nan"))
    out = C.POINTER(sassy.CMatch)()
    n = L.search(h, sent, len(sent), text, len(text), 8, C.byref(out))
    want = oracle.search("ascii", sent, text, 8)
    assert [(out[i].text_start, out[i].text_end, out[i].cost) for i in range(n)] == \
           [(m.text_start, m.text_end, m.cost) for m in want]
    L.sassy_matches_free(out, n)
    L.sassy_searcher_free(h)
    # many short texts: one lane per text
    texts = [text[i:i + 700] for i in range(0, 30_000, 650)]
    got = s.search_many([sent[:40], sent], texts, 4)
    want = []
    for pi, pat in enumerate([sent[:40], sent]):
        for ti, t in enumerate(texts):
            want += [(pi, ti, m.text_start, m.text_end, m.cost, m.cigar) for m in oracle.search("ascii", pat, t, 4)]
    assert [(m.pattern_idx, m.text_idx, m.text_start, m.text_end, m.cost, m.cigar) for m in got] == want
    assert len(want) >= 3


def test_ascii_binary_patterns_and_very_long_patterns(sassy):
    """The two cliffs the drop-in search() fell off: Ascii patterns with more than 64 distinct bytes (the reference's
    profile has 256 slots, src/profiles/ascii.rs:13-29: byte mode here -- bit planes compared with the row's byte)
    and patterns whose per-row carries exceed four waves' LDS (m = 2 500 .. 4 096: fewer waves per workgroup).
    Against the oracle, single and many texts, and through the drop-in symbol."""
    rng = random.Random(33)
    n = 40_000
    text = bytearray(rng.randrange(256) for _ in range(n))
    pats = [bytes(range(40, 140)), bytes(rng.sample(range(256), 200)), bytes(range(256)), bytes(rng.randrange(256) for _ in range(300))]
    for i, p in enumerate(pats):
        for e in (0, 3, 9):
            ins = mutate(rng, p, e)
            at = 1000 + 9000 * i + 2000 * (e % 4)
            text[at:at + len(ins)] = ins
    text = bytes(text[:n])
    s = sassy.Searcher("ascii", rc=False)
    for p, k in zip(pats, (3, 10, 5, 12)):
        assert len(set(p)) > 64
        want = oracle.search("ascii", p, text, k)
        assert len(want) >= 2, (len(p), k)
        assert_same(s.search(p, text, k), want, ("bytes", len(p), k))
        assert_same(s.search_all(p, text[:12000], 2), oracle.search("ascii", p, text[:12000], 2, all_minima=True), ("bytes all", len(p)))
    texts = [text[i:i + 3000] for i in range(0, n, 2900)]
    got = s.search_many([pats[0]], texts, 3)
    want = [(ti, m.text_start, m.text_end, m.cost, m.cigar) for ti, t in enumerate(texts) for m in oracle.search("ascii", pats[0], t, 3)]
    assert [(m.text_idx, m.text_start, m.text_end, m.cost, m.cigar) for m in got] == want and len(want) >= 1
    L = sassy.lib()
    h = L.sassy_searcher(b"ascii", False, float("nan"))
    out = C.POINTER(sassy.CMatch)()
    cnt = L.search(h, pats[2], len(pats[2]), text, len(text), 5, C.byref(out))
    want = oracle.search("ascii", pats[2], text, 5)
    assert [(out[i].text_start, out[i].text_end, out[i].cost) for i in range(cnt)] == [(m.text_start, m.text_end, m.cost) for m in want]
    L.sassy_matches_free(out, cnt)
    L.sassy_searcher_free(h)
    # very long patterns: Dna / Iupac, with and without a prefilter
    for profile, m, k, pre in (("dna", 2500, 5, -1), ("dna", 4096, 12, -1), ("iupac", 3000, 30, -1), ("dna", 4096, 12, 0), ("ascii", 2600, 4, -1)):
        pat = rand_seq(rng, m) if profile != "ascii" else bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ") for _ in range(m))
        nt = 60_000
        t = bytearray(rand_seq(rng, nt) if profile != "ascii" else bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ") for _ in range(nt)))
        for at, e in ((100, 0), (20_000, k // 2), (45_000, k)):
            ins = mutate(rng, pat, e)
            t[at:at + len(ins)] = ins
        t = bytes(t[:nt])
        sl = sassy.Searcher(profile, rc=False).set_prefilter(pre)
        want = oracle.search(profile, pat, t, k)
        assert len(want) >= 2, (profile, m, k)
        assert_same(sl.search(pat, t, k), want, ("long", profile, m, k, pre))


@pytest.mark.gpu
def test_patterns_beyond_the_lds_carry_store(sassy):
    """The reference's row loop has no cap (src/search.rs:1060-1061, 1100-1163).  Here the per-row carries of a lane live in
    its wave's LDS -- 512 bytes per 32 rows -- which ends near 9 800 rows; longer patterns keep them in global memory
    (scan_kernel / list_kernel <.., GC>).  With and without a prefilter, Dna / Iupac / Ascii, against the oracle."""
    rng = random.Random(61)
    for profile, m, k, pre in (("dna", 9_900, 40, -1), ("dna", 12_000, 60, 0), ("iupac", 12_500, 100, -1), ("iupac", 10_100, 30, 0),
                               ("ascii", 11_000, 25, -1), ("dna", 20_000, 150, -1)):
        abc = b"abcdefghijklmnopqrstuvwxyz " if profile == "ascii" else b"ACGT"
        pat = rand_seq(rng, m, abc)
        if profile == "iupac":
            pb = bytearray(pat)
            for at in rng.sample(range(m), 12):
                pb[at] = rng.choice(b"NRYWSKM")
            pat = bytes(pb)
        nt = 3 * m + 30_000
        t = bytearray(rand_seq(rng, nt, abc))
        plain = bytes(rng.choice(b"ACGT") if c not in abc else c for c in pat) if profile == "iupac" else pat
        for at, e in ((50, 0), (m + 10_000, k // 2), (2 * m + 20_000, k)):
            ins = bytearray(mutate(rng, plain, e)) if profile != "ascii" else bytearray(plain)
            if profile == "ascii":
                for _ in range(e):
                    ins[rng.randrange(len(ins))] = rng.choice(abc)
            t[at:at + len(ins)] = ins
        t = bytes(t[:nt])
        sl = sassy.Searcher(profile, rc=False).set_prefilter(pre)
        want = oracle.search(profile, pat, t, k)
        assert len(want) >= 2, (profile, m, k)
        assert_same(sl.search(pat, t, k), want, ("beyond LDS", profile, m, k, pre))


# ------------------------------------------------------------------ device-resident text
def test_device_generator_matches_cpu_twin(sassy):
    n = 1 << 20
    buf = sassy.DeviceBuffer(n + 256)
    for first in (0, 12345, 1 << 32):
        sassy.generate_dna(buf.ptr, n, 42, first)
        assert buf.download(n) == oracle.generate_dna(42, first, n).tobytes()
    pat = bytes(oracle.generate_dna(43, 0, 32))
    sassy.generate_dna(buf.ptr, n, 42, 0)
    planted = sassy.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 14)
    want = oracle.generate_dna(42, 0, n)
    assert oracle.plant_window(42, n, 0, want, pat, 3, stride=1 << 14) == planted == 64
    assert buf.download(n) == want.tobytes()


def test_device_resident_search_and_shards(sassy):
    """Text generated on the device, searched in place; then the same text searched as 3 shards
    with halos (the multi-GPU decomposition): union of shard results == whole."""
    pat = bytes(oracle.generate_dna(43, 0, 32))
    n = (1 << 22) + 1000  # not a multiple of 64
    buf = sassy.DeviceBuffer(n + 256)
    sassy.generate_dna(buf.ptr, n, 42, 0)
    sassy.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 16)
    host = buf.download(n)
    want = oracle.search("dna", pat, host, 3)
    assert len(want) >= 60
    s = sassy.Searcher("dna", rc=False)
    got = s._search(pat, _DevText(buf.ptr, n), 3, sassy.TEXT_ON_DEVICE).matches
    assert_same(got, want)
    # both strands, search_all and without_trace on the resident text (reverse_kernel on the device)
    both = sassy.Searcher("dna", rc=True)
    rcpat = oracle.reverse_complement("dna", pat)  # its Rc strand finds the planted copies
    assert_same(both._search(rcpat, _DevText(buf.ptr, n), 3, sassy.TEXT_ON_DEVICE).matches,
                oracle.search("dna", rcpat, host, 3, rc=True))
    assert_same(both._search(rcpat, _DevText(buf.ptr, n), 2, sassy.TEXT_ON_DEVICE | sassy.ALL_MINIMA).matches,
                oracle.search("dna", rcpat, host, 2, rc=True, all_minima=True))
    wo = both._search(rcpat, _DevText(buf.ptr, n), 3, sassy.TEXT_ON_DEVICE | sassy.WITHOUT_TRACE).matches
    ref = oracle.search("dna", rcpat, host, 3, rc=True)
    # without_trace (src/search.rs:1464-1475, :868-873): Fwd keeps text_end, Rc keeps text_start
    assert [(m.text_end if m.strand == "+" else m.text_start, m.cost, m.strand) for m in wo] == \
           [(m.text_end if m.strand == "+" else m.text_start, m.cost, m.strand) for m in ref]
    halo = sassy.required_halo(len(pat), 3)
    bounds = [0, 1 << 20, (1 << 21) + 64 * 777, n]
    allm = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        h = 0 if a == 0 else halo
        r = s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, 3)
        assert r.conditional_index == -1
        allm += r.matches
    assert_same(allm, want)


class _DevText:
    """Minimal stand-in for a CUDA tensor: data_ptr/numel/is_cuda (torch is not needed here)."""

    def __init__(self, ptr, n):
        self._p, self._n = ptr, n
        self.is_cuda = True

        class _DT:
            itemsize = 1
        self.dtype = _DT()

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n

    def is_contiguous(self):
        return True


def test_shard_seam_plateau_chain(sassy):
    """A constant-cost plateau that runs across shard borders: each shard reports its exit state
    and flags the report that depends on the previous shard (conditional_index); resolving the
    chain on the host reproduces the un-sharded answer."""
    pat = b"A" * 20
    text = b"G" * 1000 + b"A" * 90 + b"C" + b"A" * 20000 + b"G" * 3000
    n = len(text)
    buf = sassy.DeviceBuffer(n + 256)
    buf.upload(text)
    s = sassy.Searcher("dna", rc=False)
    want = oracle.search("dna", pat, text, 3)
    halo = sassy.required_halo(len(pat), 3)
    bounds = [0, 64 * 40, 64 * 100, 64 * 200, n]
    prev_state = 1  # decreasing = TRUE at column 0
    allm = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        h = 0 if a == 0 else halo
        r = s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, 3)
        ms = list(r.matches)
        if r.conditional_index >= 0 and prev_state != 1:
            del ms[r.conditional_index]
        allm += ms
        if r.exit_state != 2:
            prev_state = r.exit_state
    assert_same(allm, want)


# ------------------------------------------------------------------ full-size properties
def test_full_size_3gb_properties(sassy):
    """BASELINE config 2 at full size (3 GB, |P|=32, k=3, Dna): no CPU oracle can run this, so
    check size-independent properties: every plant is found once, at its planted position,
    with cost <= its edit count, and a 2 MiB slice of the same device text agrees bit-exactly
    with the oracle."""
    n = 3_000_000_000
    pat = bytes(oracle.generate_dna(43, 0, 32))
    try:
        buf = sassy.DeviceBuffer(n + 4096)
    except sassy.SassyHipError:
        pytest.skip("cannot allocate 3 GB on this device")
    sassy.generate_dna(buf.ptr, n, 42, 0)
    planted = sassy.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 20)
    assert planted == n // (1 << 20)
    s = sassy.Searcher("dna", rc=False)
    got = s._search(pat, _DevText(buf.ptr, n), 3, sassy.TEXT_ON_DEVICE).matches
    # A plant with indels near its ends can produce two rightmost-of-plateau reports, so the
    # count is >= the number of plants; every report must sit on a plant and every plant must be
    # reported (random ACGT has ~0 natural matches at m=32, k=3).
    assert planted <= len(got) <= planted + planted // 20
    seen = set()
    for m in got:
        q = m.text_start >> 20
        assert abs(m.text_start - (q * (1 << 20) + (1 << 19))) <= 6, m
        assert m.cost <= 3 and 26 <= m.text_end - m.text_start <= 38
        seen.add(q)
    assert len(seen) == planted
    assert [m.text_end for m in got] == sorted(m.text_end for m in got)
    off = 1_500_000_000 - 4096
    sl = buf.download(1 << 21, off)
    want = oracle.search("dna", pat, sl, 3)
    sub = [m for m in got if off + 64 <= m.text_start and m.text_end <= off + (1 << 21)]
    assert [(m.text_start - off, m.text_end - off, m.cost, m.cigar) for m in sub] == \
           [(m.text_start, m.text_end, m.cost, m.cigar) for m in want if m.text_start >= 64]
    st = s.stats()
    assert st["scan_launches"] in (1, 2, 3, 4) and st["text_bytes"] == n  # one launch per sub-shard lane
    buf.free()


def _cfg4_patterns(npat):
    """BASELINE config 4's pattern set: seeded random ACGT 20-mers, seed 45."""
    flat = oracle.generate_dna(45, 0, 20 * npat).tobytes()
    return [flat[20 * i:20 * i + 20] for i in range(npat)]


def _rows_of(arr, pool, sel):
    """(pattern_idx, text_start, text_end, cost, strand, cigar) of the selected records."""
    out = []
    for r in arr[sel]:
        out.append((int(r["pattern_idx"]), int(r["text_start"]), int(r["text_end"]), int(r["cost"]), int(r["strand"]),
                    pool[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["cigar_len"])].decode()))
    return out


def test_config4_full_size_10000_patterns_against_the_oracle(sassy):
    """BASELINE config 4 at its own size: search_encoded_patterns with 10 000 pre-encoded 20-mers, k = 2,
    Iupac searcher, on the 3 GB text -- checked against the oracle, not against another HIP path:
      (a) 240 sampled patterns: on a 4 MiB slice around one of their matches and on a second, fixed 4 MiB
          slice, the complete set of matches inside the slice equals oracle.search_encoded on the slice;
      (b) EVERY reported match (all ~1.8e5) is re-derived by the oracle on a window around it: same end,
          cost, start and cigar;
      (c) for 24 patterns the full list equals a single-pattern search that streams the DP over every
          block (prefilter off, sassy_hip_set_prefilter(0)) -- a path that shares no filter with (a);
      (d) the result is sorted by the reference's comparison key and every pattern index is in range.
    Then the same text with N runs and other IUPAC letters scattered in (Iupac semantics: N matches
    everything, src/pattern_tiling/tqueries.rs:101-110): 600 patterns, slices with N runs against the oracle."""
    n = 3_000_000_000
    npat, m, k = 10_000, 20, 2
    try:
        buf = sassy.DeviceBuffer(n + 4096)
        host = np.empty(n, dtype=np.uint8)
    except (sassy.SassyHipError, MemoryError):
        pytest.skip("cannot allocate 3 GB on this device / host")
    sassy.generate_dna(buf.ptr, n, 42, 0)
    buf.download_into(host)
    pats = _cfg4_patterns(npat)
    s = sassy.Searcher("iupac", rc=False)
    enc = s.encode_patterns(pats)
    r = s.search_encoded_patterns(enc, _DevText(buf.ptr, n), k, as_result=True)
    st = s.stats()
    arr, pool = r.array, r.pool
    # (one pass for all patterns: seed -> verify -> report, a pair of launches per text segment -- not one per pattern)
    assert st["filtered"] == 6 and st["scan_launches"] < 1000 and len(arr) > 100_000
    pidx = arr["pattern_idx"].astype(np.int64)
    # (d) order and ranges
    assert int(pidx.min()) >= 0 and int(pidx.max()) < npat
    keyc = np.stack([pidx, arr["text_start"].astype(np.int64), arr["text_end"].astype(np.int64)], axis=1)
    assert (np.lexsort((keyc[:, 2], keyc[:, 1], keyc[:, 0])) == np.arange(len(arr))).all()
    assert int(arr["cost"].max()) <= k and int(arr["text_end"].max()) <= n and (arr["strand"] == 0).all()
    first_of = np.searchsorted(pidx, np.arange(npat + 1))

    # (b) every match, re-derived on its own window
    LEFT, RIGHT = 96, 8
    bad = 0
    for i in range(len(arr)):
        rec = arr[i]
        e = int(rec["text_end"])
        o = max(0, e - LEFT)
        hi = min(n, e + RIGHT)
        win = host[o:hi].tobytes()
        want = (int(rec["text_start"]) - o, e - o, int(rec["cost"]),
                pool[int(rec["cigar_off"]):int(rec["cigar_off"]) + int(rec["cigar_len"])].decode())
        found = [(w.text_start, w.text_end, w.cost, w.cigar) for w in oracle.search("iupac", pats[int(rec["pattern_idx"])], win, k)]
        if want not in found:
            bad += 1
            assert bad < 5, (i, want, found)
    assert bad == 0

    # (a) sampled patterns on slices, complete sets
    rng = random.Random(4)
    SL = 4 << 20
    sample = rng.sample(range(npat), 240)
    checked = 0
    for p in sample:
        lo, hi = int(first_of[p]), int(first_of[p + 1])
        starts = [((p * 7919) % 700) * SL]  # a fixed slice (usually without a match of this pattern)
        if hi > lo:
            anchor = int(arr["text_start"][lo + (hi - lo) // 2])
            starts.append(max(0, min(n - SL, (anchor - SL // 2) // 64 * 64)))
        for a in starts:
            sl = host[a:a + SL].tobytes()
            want = [(w.text_start + a, w.text_end + a, w.cost, w.cigar)
                    for w in oracle.search_encoded("iupac", [pats[p]], sl, k) if w.text_start >= 64 and w.text_end <= SL - 64]
            sel = [i for i in range(lo, hi) if int(arr["text_start"][i]) >= a + 64 and int(arr["text_end"][i]) <= a + SL - 64]
            got = [(ts, te, c, cg) for (_, ts, te, c, _, cg) in _rows_of(arr, pool, sel)]
            assert sorted(got) == sorted(want), (p, a)
            checked += len(want)
    assert checked >= 200

    # (c) against the streaming DP (no prefilter at all)
    plain = sassy.Searcher("iupac", rc=False).set_prefilter(0)
    for p in rng.sample(range(npat), 24):
        rr = plain._search(pats[p], _DevText(buf.ptr, n), k, sassy.TEXT_ON_DEVICE)
        assert plain.stats()["filtered"] == 0
        want = sorted((int(x["text_start"]), int(x["text_end"]), int(x["cost"])) for x in rr.array)
        lo, hi = int(first_of[p]), int(first_of[p + 1])
        got = sorted((int(x["text_start"]), int(x["text_end"]), int(x["cost"])) for x in arr[lo:hi])
        assert got == want, p

    # ---- the same text with N runs and other IUPAC letters: Iupac semantics for the text ----
    patch_rng = random.Random(9)
    spots = []
    for q in range(40):
        at = patch_rng.randrange(1 << 20, n - (1 << 20))
        ln = patch_rng.choice([1, 2, 5, 17, 23, 60, 400])
        run = b"N" * ln if q % 10 < 7 else bytes(patch_rng.choice(b"NNNNNNRYKMSWBDHVn") for _ in range(ln))
        buf.upload(run, at)
        host[at:at + len(run)] = np.frombuffer(run, dtype=np.uint8)
        spots.append(at)
    sub = pats[:600]
    enc2 = s.encode_patterns(sub)
    r2 = s.search_encoded_patterns(enc2, _DevText(buf.ptr, n), k, as_result=True)
    st2 = s.stats()  # (the seeded search again, with the pattern-tiled scan around the 40 patches of other letters)
    assert st2["filtered"] == 6 and 20 <= st2["cond_resolved"] <= 80, st2
    arr2, pool2 = r2.array, r2.pool
    pidx2 = arr2["pattern_idx"].astype(np.int64)
    first2 = np.searchsorted(pidx2, np.arange(len(sub) + 1))
    assert len(arr2) > 600 * 5  # the long N runs match every pattern
    SL2 = 1 << 16
    for p in patch_rng.sample(range(len(sub)), 40):
        lo, hi = int(first2[p]), int(first2[p + 1])
        for at in patch_rng.sample(spots, 6):
            a = (at - SL2 // 2) // 64 * 64
            sl = host[a:a + SL2].tobytes()
            want = [(w.text_start + a, w.text_end + a, w.cost, w.cigar)
                    for w in oracle.search_encoded("iupac", [sub[p]], sl, k) if w.text_start >= 64 and w.text_end <= SL2 - 64]
            sel = [i for i in range(lo, hi) if int(arr2["text_start"][i]) >= a + 64 and int(arr2["text_end"][i]) <= a + SL2 - 64]
            got = [(ts, te, c, cg) for (_, ts, te, c, _, cg) in _rows_of(arr2, pool2, sel)]
            assert sorted(got) == sorted(want), (p, at)
    buf.free()


def test_full_size_config_3_properties(sassy):
    """BASELINE config 3 at full text size (3 GB; Iupac, |P|=200 with N/R/Y/W, k=20): every plant found at
    its place, sorted, and a 1 MiB slice equal to the oracle."""
    n = 3_000_000_000
    try:
        buf = sassy.DeviceBuffer(n + 4096)
    except sassy.SassyHipError:
        pytest.skip("cannot allocate 3 GB on this device")
    sassy.generate_dna(buf.ptr, n, 42, 0)
    # ---- config 3 ----
    p = bytearray(oracle.generate_dna(44, 0, 200).tobytes())
    p[50], p[100], p[150], p[199] = ord("N"), ord("R"), ord("Y"), ord("W")
    pat = bytes(p)
    # the planted copies spell the ambiguity letters with a base they contain (N, R, W -> A; Y -> C)
    plain = bytes({ord("N"): 65, ord("R"): 65, ord("W"): 65, ord("Y"): 67}.get(c, c) for c in pat)
    planted = sassy.plant(buf.ptr, n, 0, n, 42, plain, 20, stride=1 << 20)
    s = sassy.Searcher("iupac", rc=False)
    got = s._search(pat, _DevText(buf.ptr, n), 20, sassy.TEXT_ON_DEVICE).matches
    import os
    if os.environ.get("SASSY_HIP_PREFILTER") != "0":
        assert s.stats()["filtered"] in (3, 4)  # a q-gram prefilter (counting by default)
    assert planted <= len(got) <= planted + planted // 5
    seen = set()
    for m in got:
        q = m.text_start >> 20
        assert abs(m.text_start - (q * (1 << 20) + (1 << 19))) <= 40, m
        assert m.cost <= 20
        seen.add(q)
    assert len(seen) == planted
    assert [m.text_end for m in got] == sorted(m.text_end for m in got)
    off = 2_000_000_000 - 4096
    sl = buf.download(1 << 20, off)
    want = oracle.search("iupac", pat, sl, 20)
    sub = [m for m in got if off + 256 <= m.text_start and m.text_end <= off + (1 << 20)]
    assert [(m.text_start - off, m.text_end - off, m.cost, m.cigar) for m in sub] == \
           [(m.text_start, m.text_end, m.cost, m.cigar) for m in want if m.text_start >= 256]
    buf.free()



def test_text_beyond_4gib_both_strands(sassy):
    """A 5 GB text (positions beyond 2^32), both strands in one pass: every plant found at its place and
    the slice that straddles the 2^32 byte border equal to the oracle's answer (with its two plants)."""
    n = 5_000_000_000
    try:
        buf = sassy.DeviceBuffer(n + 4096)
    except sassy.SassyHipError:
        pytest.skip("cannot allocate 5 GB on this device")
    sassy.generate_dna(buf.ptr, n, 42, 0)
    pat = bytes(oracle.generate_dna(43, 0, 32))
    planted = sassy.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 20)
    for profile in ("dna", "iupac"):
        s = sassy.Searcher(profile, rc=True)
        ms = s.search(pat, _DevText(buf.ptr, n), 3)
        assert [m.text_end for m in ms if m.strand == "+"] == sorted(m.text_end for m in ms if m.strand == "+")
        slots = {m.text_start >> 20 for m in ms if m.strand == "+"}
        assert len(slots) >= planted - 1, (len(slots), planted)
        off = (1 << 32) - (1 << 20)
        sl = buf.download(1 << 21, off)
        want = oracle.search(profile, pat, sl, 3, rc=True)
        inner = lambda a, b: a >= 64 and b <= (1 << 21) - 64
        got = sorted((m.text_start - off, m.text_end - off, m.cost, m.strand, m.cigar) for m in ms
                     if m.text_start >= off and inner(m.text_start - off, m.text_end - off))
        exp = sorted((m.text_start, m.text_end, m.cost, m.strand, m.cigar) for m in want if inner(m.text_start, m.text_end))
        assert len(exp) >= 2 and got == exp
    buf.free()


@pytest.mark.parametrize("profile,k", [("dna", 3), ("iupac", 3), ("dna", 8)])
def test_geometry_tuner_trials_are_exact(sassy, profile, k):
    """With the (opt-in) geometry tuner the library tries ~20 lane-chunk lengths on a resident text during its first searches:
    every trial must return the same matches (bit-plane filter, counting filter, streaming DP at k = 8)."""
    n = 300_000_000
    buf = sassy.DeviceBuffer(n + 4096)
    sassy.generate_dna(buf.ptr, n, 42, 0)
    pat = bytes(oracle.generate_dna(43, 0, 32))
    planted = sassy.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 20)
    s = sassy.Searcher(profile, rc=False).set_geometry_tuner(True)
    first = None
    for it in range(45):
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        got = canon(r)
        if first is None:
            first = got
            assert len(r) >= planted
        assert got == first, it
    sl = buf.download(1 << 20, 64 << 20)
    want = oracle.search(profile, pat, sl, k)
    ms = [m for m in r.matches if (64 << 20) + 64 <= m.text_start and m.text_end <= (65 << 20)]
    assert [(m.text_start - (64 << 20), m.text_end - (64 << 20), m.cost, m.cigar) for m in ms] == \
           [(m.text_start, m.text_end, m.cost, m.cigar) for m in want if m.text_start >= 64]
    buf.free()


# ------------------------------------------------------------------ Dna profile on real-genome letters
def test_dna_profile_text_with_other_letters(sassy):
    """Every real genome holds N runs, soft-masked (lower-case) stretches and a few other IUPAC letters.
    The reference's Dna profile scans them through the 2-bit code (c >> 1) & 3 -- N reads as G, R as C,
    Y as A, ... (src/profiles/dna.rs:19-40) -- but its traceback compares bytes case-insensitively
    (:48-50), so a match that runs over such a letter either gets an alignment with an X where the scan
    saw a match, or the walk finds no ancestor and the reference panics (src/trace.rs:360-388).  The
    HIP path must agree with the oracle's restatement of exactly that, case by case: same matches, or
    a loud failure where the reference would panic -- through the streaming DP and the prefilters alike."""
    rng = random.Random(77)
    outcomes = {"same": 0, "both_fail": 0}
    s_fwd = sassy.Searcher("dna", rc=False)
    s_rc = sassy.Searcher("dna", rc=True)
    for case in range(60):
        m = rng.choice([12, 20, 32, 32, 48, 90])
        k = rng.choice([0, 1, 2, 3]) if m < 40 else rng.choice([3, 5, 8])
        pat = rand_seq(rng, m)
        n = rng.choice([300, 5000, 70_000])
        text = bytearray(rand_seq(rng, n))
        # N runs, a soft-masked stretch, scattered ambiguity letters
        for _ in range(rng.randrange(1, 4)):
            a = rng.randrange(0, n)
            ln = rng.choice([1, 3, 40, 500])
            text[a:a + ln] = b"N" * len(text[a:a + ln])
        a = rng.randrange(0, n)
        text[a:a + 200] = bytes(text[a:a + 200]).lower()
        for _ in range(n // 2000):
            text[rng.randrange(0, n)] = rng.choice(b"RYKMSWnryx-")
        # near-copies of the pattern, some with their letters replaced by what aliases to them
        alias = {ord("G"): b"NnKg", ord("C"): b"RrSc", ord("A"): b"YyXa", ord("T"): b"Ut"}
        for _ in range(rng.randrange(2, 6)):
            ins = bytearray(mutate(rng, pat, rng.randrange(0, k + 1)))
            if rng.random() < 0.12:
                x = rng.randrange(0, len(ins))
                ins[x] = rng.choice(alias.get(ins[x], bytes([ins[x]])))
            at = rng.randrange(0, max(1, n - len(ins)))
            text[at:at + len(ins)] = ins
        text = bytes(text[:n])
        for s, rc in ((s_fwd, False), (s_rc, True)):
            try:
                want = oracle.search("dna", pat, text, k, rc=rc)
            except RuntimeError:
                want = None
            if want is None:
                with pytest.raises(sassy.SassyHipError, match="traceback failed"):
                    s.search(pat, text, k)
                outcomes["both_fail"] += 1
            else:
                assert_same(s.search(pat, text, k), want, (case, m, k, n, rc))
                outcomes["same"] += 1
            # end positions and costs never depend on the traceback
            wo = s.search_without_trace(pat, text, k)
            ends = oracle.find_ends(oracle.last_row("dna", pat, text), k)
            assert [(x.text_end, x.cost) for x in wo if x.strand == "+"] == ends, (case, rc)
    assert outcomes["same"] >= 30 and outcomes["both_fail"] >= 10, outcomes


def test_search_many_pattern_tiled(sassy):
    """search_many with patterns of one length through the one-pass kernels over the whole batch of texts --
    the pattern-tiled scan (SASSY_HIP_MANY_TILED=1) and, for plain-ACGT patterns and texts, the seeded search
    (SASSY_HIP_MANY_SEEDED=1; host.hip: search_many_batched + finish_pattern_list with text tables) -- against
    one oracle search per (pattern, text) pair: both strands (the Rc strand runs complement(pattern) over the
    reversed batch, coordinates mapped back), matches at the first and last characters of a text, empty and
    one-character texts, search_all (reports inside separators are dropped, else moved to the text end)."""
    import os
    rng = random.Random(17)
    os.environ["SASSY_HIP_MANY_TILED"] = "1"
    try:
        for mode in ("tiled", "seeded"):
            os.environ["SASSY_HIP_MANY_SEEDED"] = "1" if mode == "seeded" else "0"
            for (profile, m, k, npat, allm) in [("iupac", 20, 2, 70, False), ("dna", 24, 3, 5, False),
                                                ("iupac", 16, 1, 3, True), ("iupac", 64, 5, 2, False)]:
                pats = [rand_seq(rng, m) for _ in range(npat)]
                pats[-1] = b"A" * m  # (its pieces also "occur" in the separators, whose bytes read as code A)
                if profile == "iupac" and mode == "tiled":
                    pats[0] = pats[0][:3] + b"N" + pats[0][4:7] + b"R" + pats[0][8:]
                texts = []
                for t in range(120):
                    n = rng.choice([0, 1, m - 1, m, m + 3, 100, 333, 1000])
                    tx = bytearray(rand_seq(rng, n))
                    if n >= m + 3:
                        p = bytes(c if c in b"ACGT" else 65 for c in rng.choice(pats))
                        ins = mutate(rng, p, rng.randrange(0, k + 1))
                        if rng.random() < 0.5:
                            ins = oracle.reverse_complement("iupac", ins)
                        if len(ins) <= n:
                            at = rng.choice([0, n - len(ins), rng.randrange(0, n - len(ins) + 1)])
                            tx[at:at + len(ins)] = ins
                    texts.append(bytes(tx))
                for rc in (False, True):
                    s = sassy.Searcher(profile, rc=rc)
                    got = s.search_many(pats, texts, k, all_minima=allm)
                    seedable = mode == "seeded" and m + 3 * k + 1 <= 64
                    assert s.stats()["filtered"] == (6 if seedable else 5), (mode, m, k, s.stats())
                    gk = sorted((x.pattern_idx, x.text_idx, x.text_start, x.text_end, x.pattern_start, x.pattern_end, x.cost,
                                 x.strand, x.cigar) for x in got)
                    wk = []
                    for pi, p in enumerate(pats):
                        for ti, tx in enumerate(texts):
                            for x in oracle.search(profile, p, tx, k, rc=rc, all_minima=allm):
                                wk.append((pi, ti, x.text_start, x.text_end, x.pattern_start, x.pattern_end, x.cost, x.strand, x.cigar))
                    assert gk == sorted(wk), (mode, profile, m, k, npat, allm, rc, len(gk), len(wk))
                    assert len(wk) >= 15
                    if not allm:
                        # the records were put in result order on the device (host.hip: assemble_many): the same
                        # records in the same order as the host's way (per-strand copies, append, stable sort)
                        s.set_option("many_assemble", 0)
                        try:
                            host = s.search_many(pats, texts, k, all_minima=allm)
                        finally:
                            s.set_option("many_assemble", 1)
                        row = lambda x: (x.pattern_idx, x.text_idx, x.text_start, x.text_end, x.pattern_start,
                                         x.pattern_end, x.cost, x.strand, x.cigar)
                        assert [row(x) for x in got] == [row(x) for x in host], (mode, profile, m, k, rc)
    finally:
        os.environ.pop("SASSY_HIP_MANY_TILED", None)
        os.environ.pop("SASSY_HIP_MANY_SEEDED", None)


def test_search_many_on_device_resident_texts(sassy):
    """search_many with texts that already live in HBM (SASSY_HIP_TEXT_ON_DEVICE): forward searchers keep
    several (pattern, text) pairs in flight on the searcher's lanes; the result is the pair-by-pair answer in
    the reference's pattern-major order, for every profile, search_all and without_trace, an empty text and
    texts of very different lengths; both-strand searchers (pair loop) give the oracle's answer too."""
    rng = random.Random(23)
    lens = [70_000, 0, 3_000, 1 << 20, 333, 64, 250_000]
    offs, total = [], 0
    for ln in lens:
        offs.append(total)
        total += (ln + 15) // 16 * 16 + 64
    buf = sassy.DeviceBuffer(total + 256)
    pats = [rand_seq(rng, 32), rand_seq(rng, 20), rand_seq(rng, 48)]
    texts = []
    for ln, off in zip(lens, offs):
        t = bytearray(rand_seq(rng, ln))
        for p in pats:
            for _ in range(max(1, ln // 40_000)):
                if ln > len(p) + 10:
                    at = rng.randrange(0, ln - len(p) - 5)
                    ins = mutate(rng, p, rng.randrange(0, 3))
                    t[at:at + len(ins)] = ins
        t = bytes(t[:ln])
        texts.append(t)
        if ln:
            buf.upload(t, off)
    dev = [_DevText(buf.ptr + off, ln) for ln, off in zip(lens, offs)]
    for profile in ("dna", "iupac", "ascii"):
        s = sassy.Searcher(profile, rc=False)
        for allm in (False, True):
            got = s.search_many(pats, dev, 2, all_minima=allm)
            want = []
            for pi, p in enumerate(pats):
                for ti, t in enumerate(texts):
                    want += [(pi, ti) + key(m)[1:] for m in oracle.search(profile, p, t, 2, all_minima=allm)]
            assert [(m.pattern_idx, m.text_idx) + key(m)[1:] for m in got] == want, (profile, allm)
            assert len(want) > 10
    both = sassy.Searcher("dna", rc=True)
    got = both.search_many(pats[:2], dev, 2)
    want = []
    for pi, p in enumerate(pats[:2]):
        for ti, t in enumerate(texts):
            want += [(pi, ti) + key(m)[1:] for m in oracle.search("dna", p, t, 2, rc=True)]
    assert [(m.pattern_idx, m.text_idx) + key(m)[1:] for m in got] == want
    buf.free()


def test_reference_lane_reports_mode(sassy):
    """sassy_hip_set_reference_lanes(4): the reports of the reference binary built for AVX2 -- 4 lanes that each
    start with decreasing = true (src/search.rs:1016-1056, 1202-1240) -- against the reference-shaped port
    oracle.refstyle_ends (RS_LANES = 4) on the periodic / low-complexity fixtures where they differ from the
    definition (the documented artefact: A^20, k = 3 -> an extra (953, 1)), on random text (no difference), with
    traced records equal to the oracle's traceback of every reported end, through the drop-in symbol with
    SASSY_HIP_REF_LANES semantics set per searcher, and for both strands."""
    assert oracle.lib().rs_lanes() == 4
    rng = random.Random(3)
    s_def = sassy.Searcher("dna", rc=False)
    s_ref = sassy.Searcher("dna", rc=False).set_reference_lanes(4)
    s_ref8 = sassy.Searcher("dna", rc=False).set_reference_lanes(8)
    differ8 = 0
    # the documented case (tests/test_oracle_diff.py::test_lane_seam_artefact_documented)
    pat = b"A" * 20
    text = b"A" * 92 + (b"C" + b"A" * 19) * 43 + b"G" * 51
    assert [(m.text_end, m.cost) for m in s_def.search(pat, text, 3)] == [(92, 0)]
    assert [(m.text_end, m.cost) for m in s_ref.search(pat, text, 3)] == [(92, 0), (953, 1)]
    differ = 0
    for it in range(400):
        m = rng.choice([8, 12, 20, 32, 40, 60, 70, 90])
        k = min(rng.choice([1, 2, 3, 5, 8]), m - 1)
        style = 0 if it % 2 == 0 else 1 + (it // 2) % 2
        if style == 0:
            sep = rng.choice([b"C", b"CG", b"CCC"])
            per = rng.choice([m - 1, m, m + 1, m // 2, 2 * m])
            text = b"A" * rng.randrange(0, 200) + (sep + b"A" * per) * rng.randrange(5, 120) + b"G" * rng.randrange(0, 100)
            pat = b"A" * m
        elif style == 1:
            unit = rand_seq(rng, rng.choice([1, 2, 3, 5]))
            n = rng.choice([100, 257, 1003, 5000])
            text, pat = (unit * (n // len(unit) + 1))[:n], (unit * (m // len(unit) + 1))[:m]
        else:
            pat = rand_seq(rng, m)
            text = bytearray(rand_seq(rng, rng.choice([64, 500, 3000, 20000])))
            for _ in range(3):
                if len(text) > m + 5:
                    at = rng.randrange(0, len(text) - m)
                    text[at:at + m] = pat
            text = bytes(text)
        want, _ = oracle.refstyle_ends("dna", pat, text, k)
        got = s_ref.search(pat, text, k)
        assert [(x.text_end, x.cost) for x in got] == want, (it, m, k, len(text))
        want8, _ = oracle.refstyle_ends("dna", pat, text, k, lanes=8)  # the AVX-512 build of the reference
        assert [(x.text_end, x.cost) for x in s_ref8.search(pat, text, k)] == want8, (it, m, k, len(text), 8)
        differ8 += want8 != want
        plain = [(x.text_end, x.cost) for x in s_def.search(pat, text, k)]
        differ += plain != want
        # every record is the traceback of its end position, as the definition's oracle traces that end
        by_end = {x.text_end: x for x in oracle.search("dna", pat, text, k, all_minima=True)}
        for x in got:
            w = by_end[x.text_end]
            assert (x.text_start, x.cost, x.cigar) == (w.text_start, w.cost, w.cigar), (it, x)
        wo = s_ref.search_without_trace(pat, text, k)
        assert [(x.text_end, x.cost) for x in wo] == want
    assert differ >= 3, (differ, differ8)  # the mode is exercised where it matters (a few % of the periodic fixtures)
    # both strands: the Rc strand is the same lane scheme on the reversed text with complement(pattern)
    both = sassy.Searcher("dna", rc=True).set_reference_lanes(4)
    text = b"A" * 92 + (b"C" + b"A" * 19) * 43 + b"G" * 51
    rc_text = oracle.reverse_complement("dna", text)
    got = both.search(b"A" * 20, rc_text, 3)
    fw = [(m.text_end, m.cost) for m in got if m.strand == "+"]
    rc = [(len(rc_text) - m.text_start, m.cost) for m in got if m.strand == "-"]
    want_fw, _ = oracle.refstyle_ends("dna", b"A" * 20, rc_text, 3)
    assert fw == want_fw and rc == [(92, 0), (953, 1)]
    with pytest.raises(sassy.SassyHipError, match="0, 4 or 8"):
        s_def.set_reference_lanes(5)


# ------------------------------------------------------------------ texts that are not i.i.d.
def _check_slices(sassy, buf, n, profile, pat, k, r, starts, SL=1 << 20):
    """the matches of a whole-text search that lie inside a few slices, against the oracle on each slice"""
    arr, pool = r.array, r.pool
    ts, te = arr["text_start"].astype(np.int64), arr["text_end"].astype(np.int64)
    assert (np.diff(te) > 0).all() and (arr["cost"] <= k).all()
    compared = 0
    for a in starts:
        a = min(max(0, a // 64 * 64), n - SL)
        sl = buf.download(SL, a)
        want = [(m.text_start + a, m.text_end + a, m.cost, m.cigar) for m in oracle.search(profile, pat, sl, k)
                if m.text_start >= 256 and m.text_end <= SL - 256]
        sel = np.nonzero((ts >= a + 256) & (te <= a + SL - 256))[0]
        got = [(int(ts[i]), int(te[i]), int(arr["cost"][i]),
                pool[int(arr["cigar_off"][i]):int(arr["cigar_off"][i]) + int(arr["cigar_len"][i])].decode()) for i in sel]
        assert got == want, (profile, pat, a, len(got), len(want))
        compared += len(want)
    return compared


@pytest.mark.parametrize("case", ["dense_plants", "periodic_pattern", "repeats_family", "repeats_microsatellite",
                                  "repeats_polyA", "repeats_with_N_iupac"])
def test_texts_that_are_not_iid(sassy, case):
    """SURVEY 8(d)'s dense-plant variant (a near-match every 4 KiB: the output path), the periodic BASELINE
    pattern 'ATCG'x8, and the repeat-rich synthetic text (microsatellites, repeat families, soft-masked
    stretches, N runs) on which a prefilter finds far more candidate blocks than on i.i.d. letters: whole-text
    search on the device, matches sorted and <= k, and the matches inside 1 MiB slices equal to the oracle's
    -- through the searches-in-flight entry points as well.  (tools/bench_texts.py times the same cases at 3 GB.)"""
    n = 160 << 20
    buf = sassy.DeviceBuffer(n + 4096)
    rnd32 = bytes(oracle.generate_dna(43, 0, 32))
    profile, k = "dna", 3
    if case == "dense_plants":
        pat = rnd32
        sassy.generate_dna(buf.ptr, n, 42, 0)
        planted = sassy.plant(buf.ptr, n, 0, n, 42, pat, k, stride=4096)
        assert planted == n // 4096
    elif case == "periodic_pattern":
        pat = b"ATCG" * 8
        sassy.generate_dna(buf.ptr, n, 42, 0)
        buf.upload(b"ATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCGATCG", 5 << 20)  # a plateau of its own
        buf.upload(b"ATCGATCGATCGATCGTTCGATCGATCGATCG", (40 << 20) + 17)
    else:
        sassy.generate_genome_like(buf.ptr, n, 42, 0, with_n=case.endswith("iupac"))
        first_region = buf.download(1 << 16, 0)
        if case == "repeats_family":
            # a 32-mer of the consensus of family 0: read it off a copy in the text itself (the longest exact
            # repeat between two repeat regions would do; simpler: the generator's rule, restated in the tool)
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
            from bench_texts import consensus_32mer
            pat = consensus_32mer(0, 1000)
        elif case == "repeats_microsatellite":
            pat = b"AC" * 16
        elif case == "repeats_polyA":
            pat = b"A" * 32
        else:
            profile, pat = "iupac", rnd32
        assert len(first_region) == 1 << 16
    s = sassy.Searcher(profile, rc=False)
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
    assert s.stats()["filtered"] != 0 or os.environ.get("SASSY_HIP_PREFILTER") == "0"
    nm = len(r)
    expect_min = {"dense_plants": n // 4096, "periodic_pattern": 2, "repeats_family": 200, "repeats_microsatellite": 20000,
                  "repeats_polyA": 1000, "repeats_with_N_iupac": 500}[case]
    assert nm >= expect_min, (case, nm)
    starts = [0, n // 3, n - (1 << 20), 5 << 20, 40 << 20]
    if nm:
        starts.append(int(r.array["text_start"][nm // 2]) - (1 << 19))
    compared = _check_slices(sassy, buf, n, profile, pat, k, r, starts)
    assert compared >= min(5, expect_min)
    # the same search as one of two in flight: identical records
    t1 = s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, k)
    t2 = s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, k)
    r1, r2 = s.search_finish(t1), s.search_finish(t2)
    assert canon(r1) == canon(r) == canon(r2)
    buf.free()


# ------------------------------------------------------------------ searches in flight
def test_searches_in_flight_begin_finish(sassy):
    """sassy_hip_search_shard_begin / sassy_hip_search_finish: two searches in flight on one searcher give
    exactly the results of the one-at-a-time calls (different patterns, different k, shards with halos,
    finished in either order), a third begin is refused, a NULL result pointer discards, and a searcher
    can be freed with a ticket still open."""
    rng = random.Random(5)
    n = (1 << 22) + 777
    pats = [bytes(oracle.generate_dna(43 + i, 0, m)) for i, m in enumerate([32, 20, 32, 64, 48, 24])]
    ks = [3, 1, 2, 5, 3, 2]
    text = bytearray(oracle.generate_dna(42, 0, n).tobytes())
    for p, k in zip(pats, ks):
        for _ in range(40):
            ins = mutate(rng, p, rng.randrange(0, k + 1))
            at = rng.randrange(0, n - 100)
            text[at:at + len(ins)] = ins
    buf = sassy.DeviceBuffer(n + 256)
    buf.upload(bytes(text[:n]))
    s = sassy.Searcher("dna", rc=False)
    ref = sassy.Searcher("dna", rc=False)
    want = [ref.search_shard(p, buf.ptr, 0, n, 0, n, k).matches for p, k in zip(pats, ks)]
    assert all(len(w) > 20 for w in want[:3])
    # a stream of searches, two in flight, finished oldest first
    s.set_option("pipe_depth", 2)  # (the default; a forced-switch run may have set another)
    got, pending = [], []
    for p, k in zip(pats, ks):
        pending.append(s.search_shard_begin(p, buf.ptr, 0, n, 0, n, k))
        if len(pending) == 2:
            got.append(s.search_finish(pending.pop(0)).matches)
    while pending:
        got.append(s.search_finish(pending.pop(0)).matches)
    for g, w in zip(got, want):
        assert_same(g, w)
    # newest first, and shards with halos
    halo = sassy.required_halo(64, 5)
    a = 1 << 21
    t1 = s.search_shard_begin(pats[0], buf.ptr, 0, a, 0, n, 3)
    t2 = s.search_shard_begin(pats[3], buf.ptr + a - halo, halo, n - a, a, n, 5)
    with pytest.raises(sassy.SassyHipError, match="in flight"):
        s.search_shard_begin(pats[1], buf.ptr, 0, n, 0, n, 1)
    r2 = s.search_finish(t2)
    r1 = s.search_finish(t1)
    assert_same(r1.matches, ref.search_shard(pats[0], buf.ptr, 0, a, 0, n, 3).matches)
    assert_same(r2.matches, ref.search_shard(pats[3], buf.ptr + a - halo, halo, n - a, a, n, 5).matches)
    assert (r2.exit_state, r2.conditional_index) == (1, -1)
    # discard, then the lane is free again; one-at-a-time calls still work on the same searcher
    L = sassy.lib()
    t = s.search_shard_begin(pats[2], buf.ptr, 0, n, 0, n, 2)
    assert L.sassy_hip_search_finish(s._h, t, None) == 0
    assert_same(s.search_shard(pats[2], buf.ptr, 0, n, 0, n, 2).matches, want[2])
    # a searcher freed with a ticket still open
    s2 = sassy.Searcher("dna", rc=False)
    s2.search_shard_begin(pats[0], buf.ptr, 0, n, 0, n, 3)
    del s2
    assert_same(s.search_shard(pats[0], buf.ptr, 0, n, 0, n, 3).matches, want[0])
    buf.free()


# ------------------------------------------------------------------ N > 1 on one GPU
def _run_ranks(script_args, world, timeout=600):
    """torch.distributed.run with `world` ranks on this box (they share the GPU)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_share_one_gpu_seam_plateau(sassy):
    """Two ranks (one process each, sharing the GPU; gloo for the exchange) search their shards of texts
    whose <=k plateau crosses the rank seam, exchange the match rows in one collective (capacity grows
    from 2 rows, cigars of a 200-row pattern included) and rank 0's merged list equals the oracle's
    search of the whole text (tests/helpers/shard_ranks.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = _run_ranks([os.path.join(root, "tests", "helpers", "shard_ranks.py")], 2)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["ok"] and rep["world"] == 2 and len(rep["cases"]) == 3
    assert all(c["same"] and c["matches"] > 0 for c in rep["cases"])
    assert rep["cases"][2]["longest_cigar"] > 40 and rep["cases"][1]["regrown"] >= 1


def test_bench_launches_its_own_ranks(sassy):
    """`python bench.py --gpus 2` starts two ranks by itself, prints n_gpus = 2 and a whole-job value;
    with one visible GPU it refuses unless told that the ranks may share it; --gpus must equal the
    launcher's world size."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    small = ["--steps", "3", "--warmup", "1", "--text-bytes", str(64 << 20), "--tune-searches", "0"]
    # (two ranks sharing ONE GPU over gloo is a debugging configuration; it hung once in some forty runs of this suite --
    # never reproduced, 8 of 8 afterwards -- so a run that exceeds 200 s (it takes 15) is started once more instead of costing the suite
    # its 900 s and its verdict; a second hang fails the test)
    import signal
    import types
    for attempt in (0, 1):
        proc = subprocess.Popen([sys.executable, bench, "--gpus", "2", "--allow-shared-gpu"] + small, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True, start_new_session=True)  # (its own process group: the ranks too)
        try:
            so, se = proc.communicate(timeout=200)
            p = types.SimpleNamespace(returncode=proc.returncode, stdout=so, stderr=se)
            break
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)  # (exactly the group this test started)
            proc.communicate()
            if attempt == 1:
                raise
            print("[test_bench_launches_its_own_ranks] the two-rank run exceeded 200 s: started once more", flush=True)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["total_text_bytes"] == 2 * (64 << 20)
    assert out["matches"] >= 2 * 63 and out["value"] > 0 and "cpu_baseline" not in out
    import torch
    if torch.cuda.device_count() < 2:
        p = subprocess.run([sys.executable, bench, "--gpus", "2"] + small, capture_output=True, text=True, timeout=900)
        assert p.returncode != 0 and "one rank per GPU" in (p.stdout + p.stderr)
    p = _run_ranks([bench, "--gpus", "3", "--allow-shared-gpu"] + small, 2)
    assert p.returncode != 0 and "must equal --gpus" in (p.stdout + p.stderr)


# ------------------------------------------------------------------ the boundary from C
def test_compiled_c_client_links_and_runs(sassy, tmp_path):
    """A C translation unit written against include/sassy.h only (tests/c/dropin_client.c: the call
    order of the reference's c/example.c:14-29), compiled with gcc, linked with -lsassy_hip, run as its
    own process with two host threads that each own a searcher: every match it prints equals the
    oracle's on the same bytes."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dropin_client")
    libdir = os.path.join(root, "sassy_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-std=c11", os.path.join(root, "tests", "c", "dropin_client.c"),
                           "-I" + os.path.join(root, "include"), "-L" + libdir, "-lsassy_hip", "-lpthread", "-lm", "-o", exe])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + ":" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe], env=env, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    texts, got = {}, {}
    for line in p.stdout.split(b"\n"):
        if line.startswith(b"text "):
            _, t, body = line.split(b" ", 2)
            texts[int(t)] = body
        elif line and line[:1].isdigit():
            f = line.split()
            got.setdefault((int(f[0]), int(f[1])), []).append((int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), f[7].decode()))
    assert p.stdout.rstrip().endswith(b"done 0") and len(texts) == 2 and len(texts[0]) == 200000
    jobs = {0: ("dna", True, [(b"ACGGTCAGGTTACGATCGGATCAGTTAGCAAT", 3), (b"TTGACCAGTA", 1), (b"GATTACAGATTACA", 2)]),
            1: ("iupac", False, [(b"ACGNTCAGGTYACGATCGRATCAGTTAGCWAT", 3), (b"CCATGGCATGCCATGG", 2), (b"A" * 20, 3)])}
    total = 0
    for t, (profile, rc, searches) in jobs.items():
        for q, (pat, k) in enumerate(searches):
            want = [(m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand)
                    for m in oracle.search(profile, pat, texts[t], k, rc=rc)]
            assert got.get((t, q), []) == want, (t, q)
            total += len(want)
    assert total >= 20


# ------------------------------------------------------------------ the fused filter (one launch: filter + chunk DP)
def _fused_cases(rng):
    """(pattern, text, k) for which the Dna bit-plane prefilter applies (k + 1 <= 8 pieces of >= 7 rows)."""
    cases = []
    for it in range(60):
        k = rng.choice([0, 1, 2, 3, 3, 3, 4, 5, 7])
        m = rng.choice([7 * (k + 1), 8 * (k + 1), 8 * (k + 1) + 3, 12 * (k + 1), 12 * (k + 1) + 40, 32 if k <= 3 else 9 * (k + 1)])
        n = rng.choice([64, 100, 1000, 4096, 4097, 30_000, 70_000, 300_000])
        pat = rand_seq(rng, m)
        t = bytearray(rand_seq(rng, n))
        dense = rng.random() < 0.3
        for _ in range(n // 300 if dense else rng.randrange(12)):
            ins = mutate(rng, pat, rng.randrange(k + 2))
            if len(ins) < n:
                at = rng.randrange(0, n - len(ins) + 1)
                t[at:at + len(ins)] = ins
        if rng.random() < 0.2:  # a low-complexity stretch: plateaus, long runs of candidate blocks
            ln = min(n // 2, rng.choice([100, 1000, 5000]))
            at = rng.randrange(0, n - ln + 1)
            t[at:at + ln] = (pat[:rng.randrange(1, 5)] * ln)[:ln]
        cases.append((pat, bytes(t[:n]), k))
    # plants right at the ends of the text, and a pattern made of one letter (every block is a candidate)
    p = rand_seq(rng, 32)
    cases.append((p, p + rand_seq(rng, 5000) + p, 3))
    cases.append((p, rand_seq(rng, 63) + p[:31], 3))
    cases.append((b"A" * 32, b"A" * 20_000, 3))
    cases.append((b"AC" * 16, b"AC" * 9000 + b"G" * 77 + b"CA" * 3000, 2))
    # at most four pieces of six rows (a 20-mer with k = 2, m = 24 .. 27 with k = 3): a window in every sixteenth
    # block of random text -- many passes of the wave over its queue
    for (m, k) in ((20, 2), (24, 3), (27, 3), (12, 1), (6, 0), (11, 1), (15, 2)):  # (the last two: 5-row pieces, the streaming DP)
        p = rand_seq(rng, m)
        t = bytearray(rand_seq(rng, 200_000))
        for _ in range(300):
            ins = mutate(rng, p, rng.randrange(k + 2))
            at = rng.randrange(0, len(t) - len(ins))
            t[at:at + len(ins)] = ins
        cases.append((p, bytes(t), k))
    return cases


def _env_allows_fusing():
    """False in the forced-path processes whose switches take the fused launch out of the picture."""
    e = os.environ
    return not (e.get("SASSY_HIP_FUSED") == "0" or e.get("SASSY_HIP_PREFILTER") == "0" or e.get("SASSY_HIP_FILTER_KIND", "2") != "2" or
                e.get("SASSY_HIP_FILTER_LINEAR") or e.get("SASSY_HIP_SELF_RANK") == "0" or e.get("SASSY_HIP_TRACE_WAVE") == "0" or
                e.get("SASSY_HIP_LANES"))


def test_fused_filter_equals_classic_chain_and_oracle(sassy):
    """filter_dna_kernel<.., FUSED> (filter + chunk DP in one launch, reports deduplicated where they are ranked)
    against the classic chain (hit bitmap -> chunk list -> list kernel) and the oracle: whole texts, search_all,
    shards with halos (exit states), searches in flight."""
    rng = random.Random(77)
    can_fuse = _env_allows_fusing()
    fused = sassy.Searcher("dna", rc=False).set_fused(True)
    classic = sassy.Searcher("dna", rc=False).set_fused(False)
    ran_fused = 0
    for i, (pat, text, k) in enumerate(_fused_cases(rng)):
        want = oracle.search("dna", pat, text, k)
        got = fused.search(pat, text, k)
        st = fused.stats()
        ran_fused += st["fused"]
        assert_same(got, want, ("fused", i, len(pat), k, len(text), st["filtered"], st["fused"]))
        got = classic.search(pat, text, k)
        assert classic.stats()["fused"] == 0
        assert_same(got, want, ("classic", i, len(pat), k, len(text)))
        if i % 4 == 0:
            assert_same(fused.search_all(pat, text[:3000], k), oracle.search("dna", pat, text[:3000], k, all_minima=True), ("all", i))
    assert ran_fused >= (30 if os.environ.get("SASSY_HIP_SHORT_PIECES") == "0" else 33) or not can_fuse, ran_fused
    # shards with halos over a resident text; plants across the seams; both searchers give the same shard results
    pat = bytes(oracle.generate_dna(43, 0, 32))
    n = (1 << 21) + 333
    text = bytearray(oracle.generate_dna(42, 0, n).tobytes())
    bounds = [0, 64 * 1000, 64 * 1001, 64 * 5000, 64 * 9000, 1 << 20, n]
    for b in bounds[1:-1]:
        if b in (64 * 5000, 64 * 9000):
            continue
        for off in (-40, -3, 10):
            ins = mutate(rng, pat, rng.randrange(3))
            text[b + off:b + off + len(ins)] = ins
    for _ in range(200):
        ins = mutate(rng, pat, rng.randrange(4))
        at = rng.randrange(0, n - 64)
        text[at:at + len(ins)] = ins
    # matches that END exactly on a shard border (the border position belongs to the shard on its left), and one
    # position to either side of it
    text[64 * 5000 - 32:64 * 5000] = pat
    text[64 * 9000 - 31:64 * 9000 + 1] = pat
    text[64 * 9000 - 1000 - 33:64 * 9000 - 1000 - 1] = pat
    text = bytes(text[:n])
    buf = sassy.DeviceBuffer(n + 256)
    buf.upload(text)
    want = oracle.search("dna", pat, text, 3)
    halo = sassy.required_halo(len(pat), 3)
    fused = sassy.Searcher("dna", rc=False).set_fused(True)  # (the poly-A cases above made the old one back off)
    for s in (fused, classic):
        allm = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            h = 0 if a == 0 else halo
            r = s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, 3)
            # (no report hangs on the previous shard; the exit state only matters to a shard that has such a report)
            assert r.conditional_index == -1 and r.exit_state in (0, 1)
            allm += r.matches
        assert_same(allm, want, "shards")
        assert s is classic or fused.stats()["fused"] == 1 or not can_fuse
        allm = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            h = 0 if a == 0 else halo
            allm += s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, 2, sassy.ALL_MINIMA).matches
        assert_same(allm, oracle.search("dna", pat, text, 2, all_minima=True), "shards, search_all")
    t1 = fused.search_shard_begin(pat, buf.ptr, 0, n, 0, n, 3)
    t2 = fused.search_shard_begin(pat, buf.ptr, 0, n, 0, n, 2)
    assert_same(fused.search_finish(t2).matches, oracle.search("dna", pat, text, 2), "in flight k=2")
    assert_same(fused.search_finish(t1).matches, want, "in flight k=3")
    buf.free()


def test_iupac_searcher_plain_pattern_takes_the_dna_launch_on_any_text(sassy):
    """An Iupac searcher whose pattern holds A C G T only runs the fused Dna launch with a check of the text
    (filter_dna_kernel<.., CHECK>) on ANY text: other letters -- inside a match, where N matches every base, in the
    text's last partial 16-byte piece, in its first block, whole runs of N -- are handled where they lie (the owning lane
    queues the columns a match touching them can end in, the chunk DP builds the Iupac masks): same launch, exact."""
    rng = random.Random(123)
    can_fuse = _env_allows_fusing() and os.environ.get("SASSY_HIP_IUPAC_PLANES", "1") != "0"
    pat = rand_seq(rng, 32)
    k = 3
    for n in (70_001, 300_000, 4096, 2_000_003):
        base = bytearray(rand_seq(rng, n))
        for _ in range(25):
            ins = mutate(rng, pat, rng.randrange(k + 1))
            at = rng.randrange(0, n - len(ins))
            base[at:at + len(ins)] = ins
        for i in range(0, n, 97):
            base[i] |= 0x20  # lower case is plain
        plant = n // 2
        base[plant:plant + 32] = pat
        variants = {"plain": bytes(base)}
        t = bytearray(base); t[plant + 10] = ord("N"); t[plant + 11] = ord("N"); t[plant + 12] = ord("R"); t[plant + 13] = ord("n")
        variants["N inside a match"] = bytes(t)
        t = bytearray(base); t[n - 1] = ord("N")
        variants["N as the last byte"] = bytes(t)
        t = bytearray(base); t[0] = ord("-")
        variants["a non-letter as the first byte"] = bytes(t)
        t = bytearray(base); t[n - 40] = ord("U")
        variants["U near the end"] = bytes(t)
        t = bytearray(base)
        for a, ln in ((n // 5, 3000), (n // 5 + 3100, 31), (n // 3, 29), (n - 200, 200), (0, 50), (n // 2 + 5000, 150_000),
                      (n // 2 + 160_000, 64 * 40), (n // 2 + 170_001, 64 * 3 + 5)):
            if a + ln <= n:
                t[a:a + ln] = b"N" * ln
        for i in range(0, min(n, 50_000), 211):
            t[i] = ord("RYKMSWBDHVNX-"[i % 13])
        variants["runs of N (one longer than two windows), ambiguity letters every 211 bytes"] = bytes(t)
        for name, text in variants.items():
            s = sassy.Searcher("iupac", rc=False)
            want = oracle.search("iupac", pat, text, k)
            for rep in range(3):
                got = s.search(pat, text, k)
                st = s.stats()
                assert_same(got, want, (name, n, rep, st["fused"], st["filtered"]))
                if can_fuse and n > 4096:
                    assert st["fused"] == 1, (name, n, rep, st)  # every variant, every time
            # a pattern with an ambiguity letter never takes that launch
            pat2 = pat[:7] + b"N" + pat[8:]
            assert_same(s.search(pat2, text, k), oracle.search("iupac", pat2, text, k), (name, n, "N in the pattern"))
    # device-resident texts, searches in flight: a plain text and one with a letter that is no base, alternating -- the
    # fall-back happens inside search_finish, on the ticket's own lane
    n = (1 << 21) + 17
    clean = bytearray(oracle.generate_dna(42, 0, n).tobytes())
    for _ in range(30):
        ins = mutate(rng, pat, rng.randrange(k + 1))
        at = rng.randrange(0, n - 64)
        clean[at:at + len(ins)] = ins
    dirty = bytearray(clean)
    dirty[n // 3] = ord("N")
    bufs = []
    for t in (clean, dirty):
        b = sassy.DeviceBuffer(n + 256)
        b.upload(bytes(t))
        bufs.append(b)
    wants = [oracle.search("iupac", pat, bytes(clean), k), oracle.search("iupac", pat, bytes(dirty), k)]
    s = sassy.Searcher("iupac", rc=False)
    pending, got = [], []
    for i in range(8):
        pending.append((i & 1, s.search_shard_begin(pat, bufs[i & 1].ptr, 0, n, 0, n, k)))
        if len(pending) == 2:
            which, t = pending.pop(0)
            got.append((which, s.search_finish(t).matches))
    while pending:
        which, t = pending.pop(0)
        got.append((which, s.search_finish(t).matches))
    assert len(got) == 8
    for which, g in got:
        assert_same(g, wants[which], ("in flight", which))


def test_fused_filter_dense_runs_and_its_one_fall_back(sassy):
    """More candidate runs than a wave's LDS queue holds (a near-match every 192 bytes) and long plateaus of cost 0: the
    wave runs its chunk DP whenever the queue fills and splits long runs into windows -- one launch, the oracle's
    results.  What the fused launch still hands to the classic chain: a long FLAT plateau of cost > 0 (no window sees
    how it was entered, nothing inside settles it) -- the conditional report sends the search there, the lane backs off
    for its next searches, and the results are the oracle's."""
    rng = random.Random(78)
    can_fuse = _env_allows_fusing()
    pat = rand_seq(rng, 32)
    n = 400_000
    t = bytearray(rand_seq(rng, n))
    for at in range(0, n - 64, 192):
        ins = mutate(rng, pat, rng.randrange(3))
        t[at:at + len(ins)] = ins
    text = bytes(t[:n])
    s = sassy.Searcher("dna", rc=False)
    got = s.search(pat, text, 3)
    assert (s.stats()["fused"] == 1 and s.stats()["filtered"] == 2) or not can_fuse
    want = oracle.search("dna", pat, text, 3)
    assert len(want) > 1500
    assert_same(got, want)
    assert_same(s.search_all(pat, text[:100_000], 2), oracle.search("dna", pat, text[:100_000], 2, all_minima=True), "dense, search_all")
    # plateaus of cost 0 far longer than a window (poly-A against poly-A), one of them to the end of the text
    p2 = b"A" * 28
    for t2 in (b"G" * 5000 + b"A" * 40_000 + b"G" * 5000, b"G" * 777 + b"A" * 300_000, b"A" * 70_000 + b"C" + b"A" * 9000 + b"G" * 40):
        assert_same(s.search(p2, t2, 3), oracle.search("dna", p2, t2, 3), ("cost-0 plateau", len(t2)))
        assert s.stats()["fused"] == 1 or not can_fuse
    # a flat plateau of cost 1, 40 000 columns: conditional report -> classic chain, and the lane backs off
    p3 = b"A" * 16 + b"C" + b"A" * 15
    t3 = b"G" * 5000 + b"A" * 40_000 + b"G" * 5000
    assert_same(s.search(p3, t3, 3), oracle.search("dna", p3, t3, 3), "flat plateau of cost 1")
    assert s.stats()["fused"] == 0
    if not can_fuse:
        return
    sparse = bytes(rand_seq(rng, 100_000))
    seen = []
    for _ in range(20):
        assert s.search(pat, sparse, 3) == []
        seen.append(s.stats()["fused"])
    assert (seen[0] == 0 and seen[-1] == 1) or not can_fuse, seen


# ------------------------------------------------------------------ fuzz failures, replayed
@pytest.mark.parametrize("name", sorted(os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_regressions"))))
def test_fuzz_regressions(sassy, name):
    """Every case tests/fuzz_gpu.py ever failed on, kept as data: description line, patterns joined by '|', text."""
    import ast
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_regressions", name), "rb") as fh:
        head, pats, text = fh.read().split(b"\n", 2)
    d = ast.literal_eval(head.decode())
    if d["mode"] in ("bytes_long", "fused"):  # one pattern, one text, forward search
        assert len(pats) == d["m"] and len(text) == d["n"]
        want = oracle.search(d["profile"], pats, text, d["k"], all_minima=d["all_minima"])
        assert len(want) == d["matches"]
        for pre in (-1, 0):
            s = sassy.Searcher(d["profile"], rc=False).set_prefilter(pre)
            got = s.search_all(pats, text, d["k"]) if d["all_minima"] else s.search(pats, text, d["k"])
            assert_same(got, want, (name, pre, s.stats()["filtered"]))
        return
    pats = pats.split(b"|")
    assert d["mode"] == "encoded" and len(pats) == d["npat"] and len(text) == d["n"]
    want = sorted(key(m) for m in oracle.search_encoded(d["profile"], pats, text, d["k"], rc=d["rc"], all_minima=d["all_minima"]))
    assert len(want) == d["matches"]
    for env in ({"SASSY_HIP_SEEDED": "1"}, {"SASSY_HIP_SEEDED": "0", "SASSY_HIP_TILED": "1"}, {"SASSY_HIP_SEEDED": "0", "SASSY_HIP_TILED": "0"}, {}):
        os.environ.update(env)
        try:
            s = sassy.Searcher(d["profile"], rc=d["rc"])
            got = s.search_encoded_patterns(s.encode_patterns(pats), text, d["k"], all_minima=d["all_minima"])
            assert sorted(key(m) for m in got) == want, (env, s.stats()["filtered"], len(got), len(want))
        finally:
            for k_ in env:
                os.environ.pop(k_, None)


# ------------------------------------------------------------------ the paired filter (round 5)
_PAIR_SHAPES = [(23, 3), (20, 3), (24, 3), (27, 3), (32, 4), (34, 4), (32, 5), (35, 5), (41, 5), (10, 1), (12, 1), (13, 1), (20, 2),
                (40, 6), (48, 6), (47, 7), (55, 7)]


def _pair_geometry(m, k):
    """(super-pieces S, rows per sub-piece Q) of the paired filter for a shape (host.hip: pair_s / pair_q)."""
    s_ = (k + 2) // 2
    return s_, m // (2 * s_)


def _pair_variants(rng, pat, k):
    """Mutated copies of `pat`, each within k edits, built against the paired filter's pigeonhole: ONE super-piece keeps a
    single edit -- every row of both of its halves, every kind of edit -- and the other super-pieces take the rest of the
    budget (two edits each where it reaches), so that the pair the filter must find is the one with the edit in it."""
    m = len(pat)
    S, Q = _pair_geometry(m, k)
    out = []
    other = lambda c: bytes([rng.choice([x for x in b"ACGT" if x != c])])
    for t in range(S):
        for j in range(2 * Q):
            for kind in range(3):
                edits = {t * 2 * Q + j: kind}
                budget = k - 1
                for u in range(S):  # two edits in every other super-piece while the budget lasts, far from their borders
                    if u == t:
                        continue
                    for off in (1, Q + 1):
                        if budget > 0:
                            edits[u * 2 * Q + off] = 0
                            budget -= 1
                v = bytearray()
                for i, c in enumerate(pat):
                    if i in edits:
                        if edits[i] == 0:
                            v += other(c)
                        elif edits[i] == 1:
                            v += other(c) + bytes([c])   # an extra text character in front of row i
                        # kind 2: row i has no text character
                    else:
                        v.append(c)
                out.append(bytes(v))
    return out


def _env_allows_pairing(profile="dna"):
    e = os.environ
    return _env_allows_fusing() and e.get("SASSY_HIP_PAIR", "1") != "0" and e.get("SASSY_HIP_PREFILTER", "-1") == "-1" and \
        not (profile == "iupac" and e.get("SASSY_HIP_IUPAC_PLANES", "1") == "0")


@pytest.mark.parametrize("profile", ["dna", "iupac"])
def test_paired_filter_against_oracle(sassy, profile):
    """filter_dna_kernel<.., PAIR>: shapes whose pigeonhole pieces are 5 or 6 rows (the reference's benchmark shape m = 23,
    k = 3, benches/perf.rs:46-48) -- every single-edit placement in every super-piece with the rest of the budget spent
    elsewhere, plants at the text's first and last columns (the A-type sub-piece that would be detected behind the last
    block), ragged and block-aligned lengths, search_all, shards, an Iupac searcher on a text with other letters, tiny texts."""
    rng = random.Random(505 if profile == "dna" else 506)
    s = sassy.Searcher(profile, rc=False)
    both = sassy.Searcher(profile, rc=True)
    ran = 0
    for m, k in _PAIR_SHAPES:
        S, Q = _pair_geometry(m, k)
        if profile == "iupac" and S > 3:
            continue
        pat = rand_seq(rng, m)
        if profile == "iupac" and m - 2 * S * Q >= 1 and (m, k) != (20, 3):
            # ambiguity letters BEHIND the filter's rows -- a CRISPR guide's NGG (benches/perf.rs:46-48 with its PAM): the
            # same launch, the chunk DP with the letters' slot masks
            tail = m - 2 * S * Q
            pat = pat[:m - tail] + (b"NGG"[-tail:] if tail <= 3 else rand_seq(rng, tail - 3) + b"NRG")
        variants = _pair_variants(rng, bytes(c if c in b"ACGT" else 71 for c in pat), k)
        rng.shuffle(variants)
        variants = variants[:60]
        # over budget: one more edit than k, spread
        variants += [mutate(rng, pat, k + 1) for _ in range(6)]
        for tail in (0, 1, 64):  # the text ends inside a block / on a block border
            text = bytearray()
            for v in variants:
                text += rand_seq(rng, rng.randrange(40, 200)) + v
            text += rand_seq(rng, 100)
            n = (len(text) + 63) // 64 * 64 + tail
            text += rand_seq(rng, n - len(text))
            # the text begins with a copy that lost its first rows, and ends with one that lost its last rows (the pair in
            # front of them is found behind the last block's columns: the tail the launch always searches)
            j0, j1 = rng.randrange(0, k + 1), rng.randrange(0, k + 1)
            text[:m - j0] = pat[j0:]
            text[n - (m - j1):] = pat[:m - j1]
            text = bytes(text)
            want = oracle.search(profile, pat, text, k)
            got = s.search(pat, text, k)
            st = s.stats()
            ran += 1 if st["pair"] else 0
            assert_same(got, want, ("pair", profile, m, k, tail, st["filtered"], st["fused"], st["pair"]))
            assert len(want) >= 20
            if tail == 1:
                cut = text[:5000]
                assert_same(s.search_all(pat, cut, k), oracle.search(profile, pat, cut, k, all_minima=True), ("pair all", m, k))
                # both strands (each strand a fused launch of its own, the Rc strand's on the reversed copy)
                rtext = text[:3000] + oracle.reverse_complement(profile, text[3000:9000]) + text[9000:12000]
                assert_same(both.search(pat, rtext, k), oracle.search(profile, pat, rtext, k, rc=True), ("pair rc", m, k))
        # tiny texts: shorter than the pattern, than a block, than the look-back
        for n in (0, 1, Q, 2 * Q + 1, m - k, m, m + k, 63, 64, 65, 129):
            text = (pat * 3)[:n] if n % 2 else rand_seq(rng, n)
            assert_same(s.search(pat, text, k), oracle.search(profile, pat, text, k), ("pair tiny", m, k, n))
    if _env_allows_pairing(profile):
        assert ran >= 3 * (len(_PAIR_SHAPES) - (5 if profile == "iupac" else 0)) - 2, ran
    # a low-complexity pattern on a text of its own units: sub-piece occurrences everywhere, every sibling passes
    for unit, m, k in ((b"AC", 23, 3), (b"A", 32, 4), (b"ACG", 32, 5), (b"AAC", 23, 3)):
        pat = (unit * m)[:m]
        text = bytearray(rand_seq(rng, 30000))
        for _ in range(4):
            at, ln = rng.randrange(0, 28000), rng.choice([50, 300, 1500])
            text[at:at + ln] = (unit * ln)[:ln]
        for _ in range(30):
            text[rng.randrange(len(text))] = rng.choice(b"ACGT")
        text = bytes(text)
        assert_same(s.search(pat, text, k), oracle.search(profile, pat, text, k), ("pair low complexity", unit, m, k))
    # shards with halos over a resident text, plants across the seams and at the seams' first / last columns
    m, k = 23, 3
    pat = rand_seq(rng, m)
    n = (1 << 20) + 77
    text = bytearray(oracle.generate_dna(52, 0, n).tobytes())
    bounds = [0, 64 * 700, 64 * 701, 64 * 4000, 1 << 19, n]
    for b in bounds[1:-1]:
        for off in (-m - 2, -m, -12, -3, 0, 5):
            ins = mutate(rng, pat, rng.randrange(k + 1))
            text[b + off:b + off + len(ins)] = ins
    for v in _pair_variants(rng, pat, k)[::3]:
        at = rng.randrange(0, n - 64)
        text[at:at + len(v)] = v
    if profile == "iupac":
        for _ in range(300):
            text[rng.randrange(n)] = rng.choice(b"NRYKMnX-")
        for _ in range(3):
            at, ln = rng.randrange(0, n - 3000), rng.choice([70, 900, 2500])
            text[at:at + ln] = b"N" * ln
    text = bytes(text[:n])
    buf = sassy.DeviceBuffer(n + 256)
    buf.upload(text)
    want = oracle.search(profile, pat, text, k)
    halo = sassy.required_halo(m, k)
    allm = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        h = 0 if a == 0 else halo
        allm.append(s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, k))
    assert_same(sassy.merge_shards(allm, 1).matches, want, "pair shards")
    assert s.stats()["pair"] == 2 or not _env_allows_pairing(profile)
    t1 = s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, k)
    t2 = s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, k - 1)
    assert_same(s.search_finish(t2).matches, oracle.search(profile, pat, text, k - 1), "pair in flight k-1")
    assert_same(s.search_finish(t1).matches, want, "pair in flight")
    buf.free()



# ------------------------------------------------------------------ every kernel path, forced
_CORE = ("test_fuzz_small_texts or test_low_complexity_and_seams or test_long_pattern_iupac_config3_shape or "
         "test_traceback_variants or test_dna_profile_text_with_other_letters or test_device_resident_search_and_shards or "
         "test_shard_seam_plateau_chain or test_fused_filter_equals_classic_chain_and_oracle or test_dense_reports or "
         "test_qgram_count_filter_worst_case_edits or test_searches_in_flight_begin_finish or "
         "test_iupac_searcher_plain_pattern_takes_the_dna_launch_on_any_text or test_fused_filter_dense_runs_and_its_one_fall_back or "
         "test_paired_filter_against_oracle")
_FORCED = [
    {"SASSY_HIP_PREFILTER": "0"},                    # streaming DP over every block (scan_kernel), also multi-word
    {"SASSY_HIP_PREFILTER": "0", "SASSY_HIP_ROW_CUT": "0"},   # ... every row of every block
    {"SASSY_HIP_PREFILTER": "0", "SASSY_HIP_STAGE_BLOCKS": "2"},
    {"SASSY_HIP_PREFILTER": "1"},                    # prefilter even with 2-row pieces
    {"SASSY_HIP_PREFILTER": "1", "SASSY_HIP_FUSED": "0"},
    {"SASSY_HIP_FILTER_KIND": "1"},                  # filter_kernel (slot masks in LDS)
    {"SASSY_HIP_FILTER_KIND": "3"},                  # filter_table_kernel
    {"SASSY_HIP_FILTER_KIND": "4"},                  # filter_count_kernel
    {"SASSY_HIP_FILTER_KIND": "4", "SASSY_HIP_COUNT_STAGE_BLOCKS": "1"},  # (half lines per step; the default is whole lines)
    {"SASSY_HIP_ROW_CUT": "0"},                      # list kernels without bounded rows
    {"SASSY_HIP_FUSED": "0"},                        # classic chain: bitmap -> chunk list -> list kernel
    {"SASSY_HIP_FILTER_LINEAR": "64"},               # filter_dna_linear_kernel
    {"SASSY_HIP_TRACE_WAVE": "0"},                   # thread-per-report traceback only
    {"SASSY_HIP_SELF_RANK": "0"},                    # rank_count / rank_scatter kernels
    {"SASSY_HIP_RC_FUSED": "0"},                     # Rc strand from a reversed copy
    {"SASSY_HIP_LIST_WORDS": "0"},                   # multi-word chunk DP by the lane-per-chunk kernel
    {"SASSY_HIP_LANES": "3", "SASSY_HIP_SUBSHARD_MIN": "2048"},  # one search cut into sub-shards on several streams
    {"SASSY_HIP_IUPAC_PLANES": "0"},                 # Iupac searches with plain patterns through the Iupac chain only
    {"SASSY_HIP_FILTER_KIND": "4", "SASSY_HIP_COUNT_WPG": "4"},  # the counting filter with four waves per workgroup
    {"SASSY_HIP_BIG_PIN": "0", "SASSY_HIP_SHORT_PIECES": "0"},   # dense results through the host's vectors; no 5- / 6-row pieces
    {"SASSY_HIP_FUSED_PRESS": "8", "SASSY_HIP_EXT_EVENTS": "0"},  # a pass of the fused launch's waves every 8 queued windows
    {"SASSY_HIP_PAIR": "0"},                         # no paired filter: 5- / 6-row shapes through the paths of round 4
    {"SASSY_HIP_STRANDS_IN_FLIGHT": "0"},            # two strands that are two searches: one after the other
    # (round 6: the switches of csrc/switches.h that had no forced run)
    {"SASSY_HIP_PAIR_RC": "0", "SASSY_HIP_COMPACT_CIGARS": "0", "SASSY_HIP_ADOPT": "0", "SASSY_HIP_TRACE_THREADS": "256"},
    {"SASSY_HIP_FUSED_PROBE": "1", "SASSY_HIP_TRACE_PROBE": "1", "SASSY_HIP_TIMING": "2", "SASSY_HIP_TUNE": "1", "SASSY_HIP_PIPE_DEPTH": "3"},
    {"SASSY_HIP_COUNT_FUSED": "0", "SASSY_HIP_CTL_TWIN": "0"},  # the counting filter's classic chain: bitmap + build_chunks_kernel; a memset in front of every search
]


@pytest.mark.parametrize("env", _FORCED, ids=lambda e: ",".join(f"{k[10:]}={v}" for k, v in e.items()))
def test_forced_kernel_paths(sassy, env):
    """The environment switches of DESIGN 5.7 are read once per process, so each forced configuration runs the core
    differential tests (oracle comparisons: fuzz, seams, long patterns, tracebacks, shards, in flight) in a
    process of its own -- inside the one `pytest -m gpu` run the driver makes."""
    import subprocess
    e = dict(os.environ)
    for k_ in list(e):
        if k_.startswith("SASSY_HIP_"):
            del e[k_]
    e.update(env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", _CORE], env=e, cwd=root, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + r.stderr[-1500:])
    assert r.returncode == 0, (env, tail)
    assert " passed" in r.stdout and "failed" not in r.stdout, (env, tail)
    print(env, r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("env", [{"SASSY_HIP_SEED_NARROW": "0"}, {"SASSY_HIP_SEED_POS64": "1"}, {"SASSY_HIP_SEED_SUBTEST": "0"},
                                 {"SASSY_HIP_SEED_LAYOUT": "0"}],
                         ids=lambda e: ",".join(f"{k[10:]}={v}" for k, v in e.items()))
def test_seeded_search_forced_test_layouts(sassy, env):
    """The seeded search's sub-piece test picks its layout by shape and text (seed_kernels.hip: test_issue -- narrow,
    wide, with care words, positions beyond 32 bits) or is off; the switches are read once per process, so the seeded
    tests run once more under each forced layout in a process of their own."""
    import subprocess
    e = {k_: v for k_, v in os.environ.items() if not k_.startswith("SASSY_HIP_")}
    e.update(env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    which = "test_encoded_seeded or test_encoded_kats_through_hip or test_overhang_many_patterns_in_one_pass"
    if "SASSY_HIP_SEED_SUBTEST" not in env:  # (no test: no layout to sweep)
        which += " or test_seeded_test_geometry_sweep"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", which], env=e, cwd=root, capture_output=True, text=True, timeout=900)
    tail = (r.stdout[-3000:] + r.stderr[-1500:])
    assert r.returncode == 0, (env, tail)
    assert " passed" in r.stdout and "failed" not in r.stdout, (env, tail)


def test_synchronous_calls_are_refused_while_a_ticket_is_open(sassy):
    """A search begun with search_shard_begin owns a lane's stream and result buffers until it is finished: the
    synchronous entry points of the same searcher must not run in between (they would overwrite what finish() reads)."""
    pat = bytes(oracle.generate_dna(43, 0, 32))
    n = 1 << 20
    text = bytearray(oracle.generate_dna(42, 0, n).tobytes())
    text[5000:5032] = pat
    text = bytes(text)
    buf = sassy.DeviceBuffer(n + 256)
    buf.upload(text)
    s = sassy.Searcher("dna", rc=False)
    want = oracle.search("dna", pat, text, 3)
    t = s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, 3)
    for call in (lambda: s.search(pat, text, 3), lambda: s.search_shard(pat, buf.ptr, 0, n, 0, n, 3),
                 lambda: s.search_many([pat], [text], 3), lambda: s.search_encoded_patterns(s.encode_patterns([pat]), text, 3),
                 lambda: s.set_stream(0)):
        with pytest.raises(sassy.SassyHipError, match="in flight"):
            call()
    assert_same(s.search_finish(t).matches, want)
    assert_same(s.search(pat, text, 3), want)
    buf.free()


# ------------------------------------------------------------------ several shards / devices through the C-ABI
def _shard_results(sassy, s, buf, n, pat, k, bounds):
    halo = sassy.required_halo(len(pat), k)
    rs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        h = 0 if a == 0 else halo
        rs.append(s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, k))
    return rs


def test_merge_shards_in_c_equals_python_merge_and_oracle(sassy):
    """sassy_hip_merge_shards (the TRUE / FALSE / PASS chain in C) on the seam fixtures: a plateau that runs across
    several shard borders, plants on the borders, a poly-A text; equal to multigpu.merge_shard_results on the packed
    rows and to the oracle's search of the whole text; merges nest (incoming_state = PASS)."""
    from sassy_amd import multigpu
    rng = random.Random(91)
    cases = []
    pat = b"A" * 20
    text = b"G" * 1000 + b"A" * 90 + b"C" + b"A" * 20000 + b"G" * 3000
    cases.append((pat, text, 3, [0, 64 * 40, 64 * 100, 64 * 200, len(text)]))
    cases.append((b"A" * 24, b"A" * 30000, 2, [0, 64 * 10, 64 * 11, 64 * 300, 30000]))
    p2 = rand_seq(rng, 32)
    t2 = bytearray(rand_seq(rng, 200_000))
    b2 = [0, 64 * 500, 64 * 1000, 64 * 1001, 64 * 2500, len(t2)]
    for b in b2[1:-1]:
        for off in (-45, -31, -2, 7):
            ins = mutate(rng, p2, rng.randrange(4))
            t2[b + off:b + off + len(ins)] = ins
    cases.append((p2, bytes(t2[:200_000]), 3, b2))
    s = sassy.Searcher("dna", rc=False)
    for ci, (pat, text, k, bounds) in enumerate(cases):
        n = len(text)
        buf = sassy.DeviceBuffer(n + 256)
        buf.upload(text)
        want = oracle.search("dna", pat, text, k)
        rs = _shard_results(sassy, s, buf, n, pat, k, bounds)
        py_rows = multigpu.merge_shard_results([multigpu.pack_result(r) for r in rs])
        rs = _shard_results(sassy, s, buf, n, pat, k, bounds)  # (pack_result materialised the first set)
        merged = sassy.merge_shards(rs, incoming_state=1)
        assert merged.conditional_index == -1
        assert_same(merged.matches, want, ("c merge", ci))
        assert [int(x) for x in py_rows[:, 2]] == [m.text_end for m in want], ("python merge", ci)
        # nested: the right part merged first with an unknown left neighbour, then joined to the left part
        rs = _shard_results(sassy, s, buf, n, pat, k, bounds)
        right = sassy.merge_shards(rs[2:], incoming_state=2)
        both = sassy.merge_shards([sassy.merge_shards(rs[:2], incoming_state=1), right], incoming_state=1)
        assert_same(both.matches, want, ("nested merge", ci))
        buf.free()


def test_multi_searcher_shards_on_one_gpu(sassy):
    """sassy_hip_multi_*: the in-process multi-device searcher with several shards on the one GPU of the test box (a
    device may be named more than once): host text split with halos and uploaded by one thread per shard, searched by
    all shards at once, merged in C -- equal to the oracle, with a plateau across the shard borders, search_all,
    without_trace, a synthetic text generated shard by shard, and a search from another host thread."""
    import threading
    rng = random.Random(92)
    ms = sassy.MultiSearcher("dna", devices=[0, 0, 0])
    assert ms.shards == 3 and ms.devices() == [0, 0, 0]
    pat = rand_seq(rng, 32)
    n = 300_000 + 17
    t = bytearray(rand_seq(rng, n))
    per = -(-(-(-n // 3)) // 64) * 64
    for b in (per, 2 * per):
        for off in (-40, -20, -1, 3):
            ins = mutate(rng, pat, rng.randrange(4))
            t[b + off:b + off + len(ins)] = ins
    for _ in range(100):
        ins = mutate(rng, pat, rng.randrange(4))
        at = rng.randrange(0, n - 64)
        t[at:at + len(ins)] = ins
    text = bytes(t[:n])
    ms.set_text(text, 64, 6)
    assert_same(ms.search(pat, 3).matches, oracle.search("dna", pat, text, 3), "multi k=3")
    assert_same(ms.search(pat, 2, sassy.ALL_MINIMA).matches, oracle.search("dna", pat, text, 2, all_minima=True), "multi all")
    wo = ms.search(pat, 3, sassy.WITHOUT_TRACE).matches
    assert [(m.text_end, m.cost) for m in wo] == [(m.text_end, m.cost) for m in oracle.search("dna", pat, text, 3)]
    with pytest.raises(sassy.SassyHipError, match="halo"):
        ms.search(rand_seq(rng, 200), 20)
    # a plateau across both borders
    pa, ta = b"A" * 20, b"G" * 500 + b"A" * 150_000 + b"C" + b"A" * 50_000 + b"G" * 99
    ms.set_text(ta, 32, 3)
    assert_same(ms.search(pa, 3).matches, oracle.search("dna", pa, ta, 3), "multi plateau")
    # synthetic text, generated on the devices shard by shard (global positions), searched from another thread
    n2 = (1 << 21) + 999
    ms.generate_dna(n2, 42, 32, 3)
    p2 = bytes(oracle.generate_dna(43, 0, 32))
    ms.plant(42, p2, 3, 1 << 16)
    host = oracle.generate_dna(42, 0, n2)
    oracle.plant_window(42, n2, 0, host, p2, 3, stride=1 << 16)
    want = oracle.search("dna", p2, host.tobytes(), 3)
    got = []
    th = threading.Thread(target=lambda: got.append(ms.search(p2, 3).matches))
    th.start()
    th.join()
    assert len(want) >= 30
    assert_same(got[0], want, "multi synthetic, other thread")
    # the count of plants is sassy_hip_plant's own rule over the whole text
    stride = 1 << 16
    assert ms.plant(42, p2, 3, stride) == (0 if n2 < stride // 2 + 32 + 3 else (n2 - (stride // 2 + 32 + 3)) // stride + 1)
    # a short text over many shards: the shards' offsets are smaller than the halo a search asks for -- a halo that
    # reaches byte 0 of the text is all there is, and enough
    ms8 = sassy.MultiSearcher("dna", devices=[0] * 8)
    for n3 in (1000, 1024, 5000, 130, 7):
        t3 = bytearray(rand_seq(rng, n3))
        for _ in range(4):
            ins = mutate(rng, pat, rng.randrange(4))[:max(1, n3 - 1)]
            at = rng.randrange(0, max(1, n3 - len(ins)))
            t3[at:at + len(ins)] = ins
        t3 = bytes(t3[:n3])
        ms8.set_text(t3, 32, 3)
        assert_same(ms8.search(pat, 3).matches, oracle.search("dna", pat, t3, 3), ("tiny text over 8 shards", n3))
        assert_same(ms8.search(pat, 3, sassy.ALL_MINIMA).matches, oracle.search("dna", pat, t3, 3, all_minima=True), ("tiny all", n3))



def _multi_in_flight_case(sassy, devices, profile, rc, n, rng):
    """Searches in flight over several devices (sassy_hip_multi_search_begin / _finish): different patterns and k in
    flight at once, finished out of order, both strands with the cached reversed shards, a text that changes in between
    (the cache must go), tiny texts on a both-strand multi-searcher (ADVICE of round 4: one device takes them)."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    ms = sassy.MultiSearcher(profile, devices=devices).set_rc(rc)
    G = len(devices)
    pats = [rand_seq(rng, m) for m in (32, 23, 40, 32)]
    ks = [3, 3, 4, 2]
    t = bytearray(rand_seq(rng, n))
    per = -(-(-(-n // G)) // 64) * 64
    for g in range(1, G):
        for b in (g * per, n - g * per):
            for off in (-40, -20, -1, 3):
                p_ = rng.choice(pats)
                ins = mutate(rng, p_ if rng.random() < 0.5 else p_.translate(comp)[::-1], rng.randrange(3))
                at = max(0, min(n - len(ins), b + off))
                t[at:at + len(ins)] = ins
    for _ in range(120):
        p_ = rng.choice(pats)
        ins = mutate(rng, p_ if rng.random() < 0.5 else p_.translate(comp)[::-1], rng.randrange(4))
        at = rng.randrange(0, n - 64)
        t[at:at + len(ins)] = ins
    text = bytes(t[:n])
    ms.set_text(text, 64, 6)
    ms.set_pipe_depth(3)
    want = [oracle.search(profile, p_, text, k_, rc=rc) for p_, k_ in zip(pats, ks)]
    tickets = [ms.search_begin(pats[i], ks[i]) for i in range(3)]
    with pytest.raises(sassy.SassyHipError, match="in flight"):
        ms.search_begin(pats[3], ks[3])          # a fourth one: the depth is three
    with pytest.raises(sassy.SassyHipError, match="in flight"):
        ms.search(pats[0], 3)                    # the synchronous call is refused while tickets are open
    assert_same(ms.search_finish(tickets[1]).matches, want[1], ("in flight, second first", devices, rc))
    tickets.append(ms.search_begin(pats[3], ks[3]))
    for i in (0, 3, 2):
        assert_same(ms.search_finish(tickets[i]).matches, want[i], ("in flight", i, devices, rc))
    assert sum(len(w) for w in want) >= 40
    # a stream of them
    pend, got = [], []
    for i in range(12):
        pend.append((i % 4, ms.search_begin(pats[i % 4], ks[i % 4])))
        if len(pend) == 3:
            j, tk = pend.pop(0)
            got.append((j, ms.search_finish(tk).matches))
    while pend:
        j, tk = pend.pop(0)
        got.append((j, ms.search_finish(tk).matches))
    for j, g_ in got:
        assert_same(g_, want[j], ("stream", j))
    assert_same(ms.search(pats[0], ks[0]).matches, want[0], "synchronous again")
    # the text changes: the reversed shards are made again
    t2 = bytearray(text)
    t2[n // 2:n // 2 + 32] = pats[0].translate(comp)[::-1]
    t2[100:132] = pats[0]
    text2 = bytes(t2)
    ms.set_text(text2, 64, 6)
    tk = ms.search_begin(pats[0], 3)
    assert_same(ms.search_finish(tk).matches, oracle.search(profile, pats[0], text2, 3, rc=rc), "changed text")
    # a ticket is open: nothing may rewrite, re-lay-out or reallocate the resident shards (or their reversed copies) the
    # search in flight reads, and no synchronous search may use the parts' searchers (round 5's advice: only
    # multi_search itself was refused)
    tk = ms.search_begin(pats[0], 3)
    for call in (lambda: ms.set_text(text, 64, 6), lambda: ms.generate_dna(n, 7, 64, 6), lambda: ms.plant(1, pats[0], 3),
                 lambda: ms.set_rc(rc), lambda: ms.set_replicated(False), lambda: ms.search(pats[0], 3),
                 lambda: ms.search_encoded([pats[0][:20]], 2), lambda: ms.search_many([pats[0]], [text[:1000]], 3)):
        with pytest.raises(sassy.SassyHipError, match="in flight"):
            call()
    assert_same(ms.search_finish(tk).matches, oracle.search(profile, pats[0], text2, 3, rc=rc), "after the refused calls")
    # tiny texts: fewer blocks than parts
    for n_t in (0, 1, 31, 63, 64, 65, 130, 64 * G + 1, 64 * G * (G + 2) - 1, 64 * G * (G + 2)):
        tt = (pats[0] * 40)[:n_t] if n_t % 3 else rand_seq(rng, n_t)
        ms.set_text(tt, 64, 6)
        assert_same(ms.search(pats[0], 3).matches, oracle.search(profile, pats[0], tt, 3, rc=rc), ("tiny", n_t, G, rc))
        tk = ms.search_begin(pats[0], 3)
        assert_same(ms.search_finish(tk).matches, oracle.search(profile, pats[0], tt, 3, rc=rc), ("tiny in flight", n_t, G, rc))


def test_multi_searcher_searches_in_flight_on_one_gpu(sassy):
    rng = random.Random(95)
    _multi_in_flight_case(sassy, [0, 0, 0], "dna", False, 300_017, rng)
    _multi_in_flight_case(sassy, [0, 0, 0, 0], "iupac", True, 260_000, rng)
    _multi_in_flight_case(sassy, [0], "dna", True, 100_003, rng)


def test_two_real_devices(sassy):
    """Two real GPUs (hipGetDeviceCount() >= 2): the text sharded over them, searches one at a time and in flight, both
    strands, search_encoded with the patterns sharded, search_many with the texts sharded -- against the oracle.  Skipped,
    and reported as skipped, on a one-GPU box: nothing in this suite has ever had a second device."""
    if sassy.device_count() < 2:
        pytest.skip(f"{sassy.device_count()} HIP device(s) visible: the two-device test needs two")
    rng = random.Random(96)
    _multi_in_flight_case(sassy, [0, 1], "dna", False, 400_001, rng)
    _multi_in_flight_case(sassy, [0, 1], "iupac", True, 300_000, rng)
    ms = sassy.MultiSearcher("dna", devices=[0, 1])
    n2 = (1 << 24) + 999
    ms.generate_dna(n2, 42, 32, 3)
    p2 = bytes(oracle.generate_dna(43, 0, 32))
    ms.plant(42, p2, 3, 1 << 18)
    host = oracle.generate_dna(42, 0, n2)
    oracle.plant_window(42, n2, 0, host, p2, 3, stride=1 << 18)
    assert_same(ms.search(p2, 3).matches, oracle.search("dna", p2, host.tobytes(), 3), "two devices, synthetic text")
    assert ms.devices() == [0, 1]


def test_multi_searcher_both_strands_encoded_and_many_on_one_gpu(sassy):
    """The multi-device searcher beyond forward single-pattern searches, with device 0 named several times: both
    strands (every device searches its share of the REVERSED text as a shard of its own, text lengths that are and are
    not multiples of the block size), search_encoded with the PATTERNS sharded over devices that hold the whole text,
    search_many with the TEXTS sharded -- all against the oracle / the single-device searcher."""
    rng = random.Random(93)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for profile, n, G in (("dna", 300_017, 3), ("iupac", 64 * 3 * 700, 3), ("dna", 200_001, 5), ("dna", 5000, 4)):
        ms = sassy.MultiSearcher(profile, devices=[0] * G).set_rc(True)
        pat = rand_seq(rng, 32)
        rcp = pat.translate(comp)[::-1]
        t = bytearray(rand_seq(rng, n))
        per = -(-(-(-n // G)) // 64) * 64
        for g in range(1, G):
            for b in (g * per, n - g * per):  # the forward shards' borders and the reversed shards'
                for off in (-40, -33, -20, -1, 0, 3):
                    ins = mutate(rng, pat if rng.random() < 0.5 else rcp, rng.randrange(4))
                    at = max(0, min(n - len(ins), b + off))
                    t[at:at + len(ins)] = ins
        for _ in range(60):
            ins = mutate(rng, pat if rng.random() < 0.5 else rcp, rng.randrange(4))
            at = rng.randrange(0, n - 64)
            t[at:at + len(ins)] = ins
        t[0:32] = rcp
        t[n - 32:n] = rcp
        text = bytes(t[:n])
        ms.set_text(text, 64, 6)
        assert_same(ms.search(pat, 3).matches, oracle.search(profile, pat, text, 3, rc=True), ("multi rc", profile, n, G))
        assert_same(ms.search(pat, 2, sassy.ALL_MINIMA).matches, oracle.search(profile, pat, text, 2, rc=True, all_minima=True),
                    ("multi rc all", profile, n, G))
        wo = ms.search(pat, 3, sassy.WITHOUT_TRACE).matches
        want = oracle.search(profile, pat, text, 3, rc=True)
        assert [(m.text_end if m.strand == "+" else m.text_start, m.cost, m.strand) for m in wo] == \
               [(m.text_end if m.strand == "+" else m.text_start, m.cost, m.strand) for m in want]
    # a plateau over the reversed shards' borders
    ms = sassy.MultiSearcher("dna", devices=[0, 0, 0]).set_rc(True)
    pa, ta = b"T" * 20, b"G" * 333 + b"A" * 150_000 + b"C" + b"A" * 50_000 + b"G" * 99
    ms.set_text(ta, 32, 3)
    assert_same(ms.search(pa, 3).matches, oracle.search("dna", pa, ta, 3, rc=True), "multi rc plateau")
    # search_encoded: patterns sharded, whole text everywhere
    n = 150_000
    text = bytearray(rand_seq(rng, n))
    pats = [bytes(text[997 * i + 5:997 * i + 25]) for i in range(40)] + [rand_seq(rng, 20) for _ in range(13)]
    for i in range(0, 40, 3):  # a few of them again, mutated, elsewhere
        ins = mutate(rng, pats[i], 2)
        at = 80_000 + 500 * i
        text[at:at + len(ins)] = ins
    text = bytes(text)
    for rc in (False, True):
        me = sassy.MultiSearcher("iupac", devices=[0, 0, 0]).set_replicated(True).set_rc(rc)
        me.set_text(text, 32, 3)
        got = me.search_encoded(pats, 2).matches
        want = oracle.search_encoded("iupac", pats, text, 2, rc=rc)
        assert sorted(key(m) for m in got) == sorted(key(m) for m in want), ("multi encoded", rc, len(got), len(want))
        assert len(want) >= 40
        with pytest.raises(sassy.SassyHipError, match="whole copies"):
            me.search(pats[0], 2)
    with pytest.raises(sassy.SassyHipError, match="whole text"):
        sassy.MultiSearcher("iupac", devices=[0, 0]).set_text(text, 32, 3).search_encoded(pats, 2)
    # search_many: texts sharded
    texts = [rand_seq(rng, rng.choice([0, 1, 50, 300, 2000, 9000])) for _ in range(37)]
    mpats = [rand_seq(rng, 24) for _ in range(5)]
    texts = [bytes(bytearray(x[:10]) + mutate(rng, mpats[i % 5], i % 3) + bytearray(x[10:])) if len(x) > 100 else x for i, x in enumerate(texts)]
    for rc in (False, True):
        mm = sassy.MultiSearcher("dna", devices=[0, 0, 0, 0]).set_rc(rc)
        got = mm.search_many(mpats, texts, 2).matches
        single = sassy.Searcher("dna", rc=rc).search_many(mpats, texts, 2)

        def key_t(m):
            return (m.pattern_idx, m.text_idx) + key(m)[1:]
        assert sorted(key_t(m) for m in got) == sorted(key_t(m) for m in single), ("multi many", rc, len(got), len(single))
        assert len(single) >= 10


def test_drop_in_search_over_several_devices(sassy):
    """SASSY_HIP_DEVICES names the devices the drop-in search() cuts a host text over (read once per process: a process
    of its own; device 0 three times on the one-GPU box): same matches as the oracle, both strands."""
    import subprocess
    code = r'''
import ctypes as C, random, sys
sys.path.insert(0, ".")
import oracle, sassy_amd
rng = random.Random(5)
n = (13 << 20) + 77
text = bytearray(oracle.generate_dna(42, 0, n).tobytes())
pat = bytes(oracle.generate_dna(43, 0, 32))
comp = bytes.maketrans(b"ACGT", b"TGCA")
for i in range(300):
    ins = bytearray(pat if i % 2 else pat.translate(comp)[::-1])
    for _ in range(i % 4):
        ins[rng.randrange(len(ins))] = rng.choice(b"ACGT")
    at = rng.randrange(0, n - 64)
    text[at:at + len(ins)] = ins
text = bytes(text)
L = sassy_amd.lib()
for rc in (False, True):
    s = L.sassy_searcher(b"dna", rc, float("nan"))
    out = C.POINTER(sassy_amd.CMatch)()
    cnt = L.search(s, pat, len(pat), text, len(text), 3, C.byref(out))
    got = [(out[i].text_start, out[i].text_end, out[i].pattern_start, out[i].pattern_end, out[i].cost, out[i].strand) for i in range(cnt)]
    L.sassy_matches_free(out, cnt)
    want = [(m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, 1 if m.strand == "-" else 0)
            for m in oracle.search("dna", pat, text, 3, rc=rc)]
    assert got == want, (rc, len(got), len(want), got[:3], want[:3])
    assert len(want) >= (250 if rc else 120)
    L.sassy_searcher_free(s)
print("ok")
'''
    e = dict(os.environ)
    e["SASSY_HIP_DEVICES"] = "0,0,0"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=e, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_config5_shape_24gb_in_eight_shards_on_one_gpu(sassy):
    """BASELINE config 5's data path at its own size, on the ONE GPU of the test box: 24e9 bytes of the synthetic text
    in eight shards with halos (the in-process multi-device searcher with device 0 named eight times: generated shard
    by shard at global positions, planted, searched by eight host threads, merged in C).  Size-independent properties:
    every plant is found where it was planted, in order, across all seven seams; and around every seam, and around a
    few other places, the merged result equals the oracle on a window of the same text rebuilt on the host."""
    n = 24_000_000_000
    pat = bytes(oracle.generate_dna(43, 0, 32))
    try:
        ms = sassy.MultiSearcher("dna", devices=[0] * 8)
        ms.generate_dna(n, 42, 32, 3)
    except sassy.SassyHipError as e:
        pytest.skip(f"cannot hold 24 GB on this device: {e}")
    planted = ms.plant(42, pat, 3, 1 << 20)
    assert planted == n // (1 << 20)
    t0 = time.perf_counter()
    res = ms.search(pat, 3)
    dt = time.perf_counter() - t0
    got = res.matches
    assert planted <= len(got) <= planted + planted // 20, (planted, len(got))
    ends = [m.text_end for m in got]
    assert ends == sorted(ends)
    seen = set()
    for m in got:
        q = m.text_start >> 20
        assert abs(m.text_start - (q * (1 << 20) + (1 << 19))) <= 6, m
        assert m.cost <= 3 and 26 <= m.text_end - m.text_start <= 38
        seen.add(q)
    assert len(seen) == planted
    # windows of the same text on the host: the seven seams (shards of n / 8 bytes, rounded up to 64) and others
    per = -(-(n // 8) // 64) * 64
    wlen = 1 << 21
    starts = [b * per - wlen // 2 for b in range(1, 8)] + [0, n - wlen, 12_345_678_848, 3 * per + (1 << 19) - 4096]
    for w0 in starts:
        host = oracle.generate_dna(42, w0, wlen)
        oracle.plant_window(42, n, w0, host, pat, 3, stride=1 << 20)
        want = oracle.search("dna", pat, host.tobytes(), 3)
        lo = 0 if w0 == 0 else 64  # (a window that starts inside the text: the first m + k columns are not exact)
        sub = [m for m in got if w0 + lo <= m.text_start and m.text_end <= w0 + wlen]
        assert [(m.text_start - w0, m.text_end - w0, m.cost, m.cigar) for m in sub] == \
               [(m.text_start, m.text_end, m.cost, m.cigar) for m in want if m.text_start >= lo], w0
        assert len(sub) >= 1
    print(f"config 5 shape on one GPU: {n} B in 8 shards, {len(got)} matches, {dt * 1e3:.1f} ms for the search")


def test_searcher_stays_on_its_device_from_any_thread(sassy):
    """A searcher is bound to a device (sassy_hip_set_device, or the creating thread's current one at the first
    search); calls from other host threads run there and give the same matches."""
    import threading
    pat = bytes(oracle.generate_dna(43, 0, 32))
    text = oracle.generate_dna(42, 0, 1 << 18)
    oracle.plant_window(42, 1 << 18, 0, text, pat, 3, stride=1 << 14)
    tb = text.tobytes()
    want = oracle.search("dna", pat, tb, 3)
    s = sassy.Searcher("dna", rc=False)
    assert s.device == -1
    s.set_device(0)
    assert s.device == 0
    with pytest.raises(sassy.SassyHipError, match="no such"):
        sassy.Searcher("dna", rc=False).set_device(99)
    out = []
    ths = [threading.Thread(target=lambda: out.append(s.search(pat, tb, 3))) for _ in range(1)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert_same(out[0], want)
    assert_same(s.search(pat, tb, 3), want)
    # bound by its searches: any other device id is refused, whether or not such a device exists
    with pytest.raises(sassy.SassyHipError, match="another device"):
        s.set_device(1)
    s.set_device(0)  # (naming the device it is on stays a no-op)
    assert_same(s.search(pat, tb, 3), want)
    # ... and so is a searcher that an entry point other than a search has bound (sassy_hip_set_stream runs on it)
    s3 = sassy.Searcher("dna", rc=False)
    s3.set_stream(0)
    assert s3.device == 0
    with pytest.raises(sassy.SassyHipError, match="another device"):
        s3.set_device(1)


def test_results_that_keep_their_pinned_block(sassy):
    """A shard search whose records need no editing on the host hands its pinned block to the result (no copy); the
    blocks come from a pool, at most 16 are out at a time (further results are copied), and results stay valid
    after later searches, after their searcher is gone, and when they are freed in any order."""
    pat = bytes(oracle.generate_dna(43, 0, 32))
    n = 1 << 21
    buf = sassy.DeviceBuffer(n + 256)
    sassy.generate_dna(buf.ptr, n, 42, 0)
    sassy.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 14)
    host = buf.download(n)
    want = oracle.search("dna", pat, host, 3)
    assert len(want) >= 100
    s = sassy.Searcher("dna", rc=False)
    held = [s.search_shard(pat, buf.ptr, 0, n, 0, n, 3) for _ in range(40)]
    other = s.search_shard(pat, buf.ptr, 0, n, 0, n, 1)  # a different result in between
    del s
    first = canon(held[0])
    for i in (39, 17, 3, 20):
        assert canon(held[i]) == first, i
    assert_same(held[5].matches, want)
    assert_same(other.matches, oracle.search("dna", pat, host, 1))
    random.Random(3).shuffle(held)
    while held:
        held.pop()
    s2 = sassy.Searcher("dna", rc=False)
    assert_same(s2.search_shard(pat, buf.ptr, 0, n, 0, n, 3).matches, want)
    buf.free()
