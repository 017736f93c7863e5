"""Time-boxed differential fuzz of the HIP path against the oracle on a GPU box (beyond the seeded
cases of tests/test_gpu_parity.py): random profiles, pattern shapes, text compositions (random,
periodic, low complexity, planted near-matches with worst-case edit spacing, stray letters), both
strands, search / search_all, overhang.  Stops at the first mismatch and prints the reproducer.

    python tests/fuzz_gpu.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

os.environ.setdefault("SASSY_HIP_MULTI_MIN_TEXT", "16384")  # let the multi-pattern prefilter see small texts
import oracle  # noqa: E402  (test infrastructure: this tool is a checker, not the product)
import sassy_amd  # noqa: E402


def rand_seq(rng, n, alphabet):
    return bytes(rng.choice(alphabet) for _ in range(n))


def mutate(rng, seq, edits, spaced=0):
    out = bytearray(seq)
    for e in range(edits):
        if not out:
            break
        i = (spaced * (e + 1) - 1) % len(out) if spaced else rng.randrange(len(out))
        kind = rng.randrange(3)
        if kind == 0:
            out[i] = rng.choice([x for x in b"ACGT" if x != out[i]] or b"A")
        elif kind == 1:
            out.insert(i, rng.choice(b"ACGT"))
        else:
            del out[i]
    return bytes(out)


def key(ms):
    return [(m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar) for m in ms]


def one_case(rng, searchers):
    profile = rng.choice(["dna", "iupac", "iupac", "ascii"])
    m = rng.choice([4, 8, 12, 16, 20, 21, 24, 31, 32, 33, 40, 48, 63, 64, 65, 70, 96, 100, 128, 150, 200, 257, 400])
    kmax = max(0, min(m - 1, m // 4))
    k = rng.choice([0, 1, 2, 3, 4, 6, 8, 10, 16, 25, 40])
    k = min(k, kmax)
    n = rng.choice([500, 5_000, 20_000, 70_000, 150_000, 300_000])
    if m >= 200:
        n = min(n, 70_000)
    pal = {"dna": b"ACGT", "iupac": b"ACGTNRYSWKMBDHV", "ascii": b"ACGTXYZ acgt"}[profile]
    pat = rand_seq(rng, m, pal if rng.random() < 0.3 else pal[:4])
    if rng.random() < 0.15:  # low-complexity pattern
        unit = rand_seq(rng, rng.randrange(1, 5), b"ACGT")
        pat = (unit * (m // len(unit) + 1))[:m]
    comp = rng.random()
    if comp < 0.55:
        text = bytearray(rand_seq(rng, n, b"ACGT"))
    elif comp < 0.75:
        unit = rand_seq(rng, rng.randrange(1, 40), b"ACGT")
        text = bytearray((unit * (n // len(unit) + 1))[:n])
        for _ in range(n // 200):
            text[rng.randrange(n)] = rng.choice(b"ACGT")
    else:  # text made of pieces of the pattern
        text = bytearray()
        plain = bytes(c if c in b"ACGT" else 65 for c in pat.upper())
        while len(text) < n:
            a = rng.randrange(len(plain))
            b = rng.randrange(a, len(plain)) + 1
            text += plain[a:b]
            if rng.random() < 0.5:
                text += rand_seq(rng, rng.randrange(0, 30), b"ACGT")
        text = text[:n]
    plain = bytes(c if c in b"ACGT" else 65 for c in pat.upper())
    for _ in range(rng.randrange(0, 8)):
        ins = mutate(rng, plain, rng.randrange(0, k + 2), spaced=rng.choice([0, 0, 5, 6, 7, max(1, m // (k + 1))]))
        if len(ins) + 2 >= n:
            continue
        at = rng.choice([0, n - len(ins), rng.randrange(0, n - len(ins))])
        text[at:at + len(ins)] = ins
    if profile != "dna" and rng.random() < 0.4:
        stray = {"iupac": b"NRYnacgtuUX-*", "ascii": b"XYZ xyz"}[profile]
        for _ in range(rng.randrange(1, 40)):
            text[rng.randrange(n)] = rng.choice(stray)
    elif profile == "dna" and rng.random() < 0.3:
        for _ in range(rng.randrange(1, 200)):
            i = rng.randrange(n)
            text[i] = text[i] | 0x20
    text = bytes(text)
    rc = profile != "ascii" and rng.random() < 0.4
    allm = rng.random() < 0.25
    overhang = profile == "iupac" and rng.random() < 0.12 and n <= 20_000
    desc = dict(profile=profile, m=m, k=k, n=n, rc=rc, all_minima=allm, overhang=overhang)
    if overhang:
        alpha = rng.choice([0.0, 0.3, 0.5, 1.0])
        s = sassy_amd.Searcher(profile, rc=rc, alpha=alpha)
        got = s.search_all(pat, text, k) if allm else s.search(pat, text, k)
        want = oracle.search_overhang(profile, pat, text, k, alpha, rc=rc, all_minima=allm)
        desc["alpha"] = alpha
    else:
        s = searchers[(profile, rc)]
        got = s.search_all(pat, text, k) if allm else s.search(pat, text, k)
        want = oracle.search(profile, pat, text, k, rc=rc, all_minima=allm)
    desc["filtered"] = s.stats()["filtered"]
    desc["matches"] = len(want)
    ok = key(got) == key(want)
    return ok, desc, pat, text, got, want


def fused_case(rng, searchers):
    """Dna, forward strand, k + 1 <= 8 pieces of >= 7 rows: the bit-plane filter's fused launch (window chunks, duplicate
    reports, text stash, adopted result blocks) -- larger texts, dense plants, stretches of pattern pieces, whole
    texts and shards."""
    k = rng.choice([0, 1, 2, 3, 3, 4, 5, 6, 7])
    q = rng.choice([6, 7, 8, 9, 12])  # (6: the fused launch takes 6-row pieces when there are at most four)
    m = q * (k + 1) + rng.choice([0, 0, 1, 3, 17, 60])
    if rng.random() < 0.4:
        # the paired filter's shapes: S = ceil((k+1)/2) super-pieces of two 5- or 6-row halves, k+1 pieces shorter than 7 rows
        k = rng.choice([1, 2, 3, 3, 3, 4, 5, 6, 7])
        S = (k + 2) // 2
        q = rng.choice([5, 6])
        lo, hi = 2 * S * q, min(2 * S * (q + 1), 7 * (k + 1)) - 1
        m = rng.randrange(lo, hi + 1) if hi >= lo else lo
    n = rng.choice([200, 3_000, 50_000, 300_000, 1_000_000])
    if m > 120:
        n = min(n, 300_000)
    pat = rand_seq(rng, m, b"ACGT")
    if rng.random() < 0.1:
        unit = rand_seq(rng, rng.randrange(1, 6), b"ACGT")
        pat = (unit * (m // len(unit) + 1))[:m]
    text = bytearray(rand_seq(rng, n, b"ACGT"))
    style = rng.random()
    plants = rng.randrange(0, 10) if style < 0.6 else n // rng.choice([200, 400, 1500])
    for _ in range(plants):
        ins = mutate(rng, pat, rng.randrange(0, k + 2), spaced=rng.choice([0, 0, 5, max(1, m // (k + 1))]))
        if len(ins) + 2 >= n:
            continue
        at = rng.choice([0, n - len(ins), rng.randrange(0, n - len(ins))])
        text[at:at + len(ins)] = ins
    if style > 0.85:  # pieces of the pattern back to back: many occurrences, few matches
        at = rng.randrange(0, max(1, n - 5000))
        junk = bytearray()
        while len(junk) < min(5000, n - at):
            a = rng.randrange(m)
            junk += pat[a:a + rng.randrange(q, 2 * q + 1)]
        text[at:at + len(junk)] = junk[:len(text) - at]
    profile = "dna"
    if rng.random() < 0.45:
        # an Iupac searcher with a plain pattern takes the same launch (CHECK): runs of N of every length (inside a
        # block, over block and lane borders, longer than a window), single other letters, lower case
        profile = "iupac"
        for _ in range(rng.randrange(0, 7)):
            ln = min(n, rng.choice([1, 2, 3, 31, 40, 63, 64, 65, 128, 200, 1000, 5000, 70000]))
            at = rng.choice([0, n - ln, rng.randrange(0, n - ln + 1), rng.randrange(0, n - ln + 1) // 64 * 64])
            text[at:at + ln] = b"N" * ln
        for _ in range(rng.randrange(0, 20)):
            text[rng.randrange(n)] = rng.choice(b"RYSWKMBDHVNXn-")
        if rng.random() < 0.2:
            at = rng.randrange(n)
            text[at:at + 500] = bytes(text[at:at + 500]).lower()
        if rng.random() < 0.1:
            pat = pat[:m // 2] + b"N" + pat[m // 2 + 1:]  # (an ambiguity letter in the pattern: another path)
    if rng.random() < 0.2:
        # a stretch of one short unit and a pattern of the same unit: runs of end positions far longer than a window
        unit = rand_seq(rng, rng.randrange(1, 4), b"ACGT")
        if rng.random() < 0.7:
            pat = (unit * (m // len(unit) + 1))[:m]
        for _ in range(rng.randrange(1, 4)):
            ln = min(n, rng.choice([100, 700, 5000, 20000]))
            at = rng.randrange(0, n - ln + 1)
            st = bytearray((unit * (ln // len(unit) + 1))[:ln])
            for _ in range(rng.randrange(0, 4)):
                st[rng.randrange(ln)] = rng.choice(b"ACGT")
            text[at:at + ln] = st
    text = bytes(text[:n])
    s = searchers[(profile, False)]
    allm = rng.random() < 0.2
    desc = dict(mode="fused", profile=profile, m=m, k=k, n=n, rc=False, all_minima=allm)
    if rng.random() < 0.3 and n >= 3000 and not allm:  # as shards over a resident text
        buf = sassy_amd.DeviceBuffer(n + 256)
        buf.upload(text)
        halo = sassy_amd.required_halo(m, k)
        cuts = sorted(set([0, n] + [min(n, 64 * rng.randrange(1, max(2, n // 64))) for _ in range(rng.randrange(1, 4))]))
        rs = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            h = 0 if a == 0 else min(halo, a) // 64 * 64
            if a != 0 and h < halo:
                rs = None
                break
            rs.append(s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, k))
        if rs is None:
            got = s.search(pat, text, k)
        else:
            got = sassy_amd.merge_shards(rs, 1).matches
            desc["shards"] = len(cuts) - 1
        buf.free()
    else:
        got = s.search_all(pat, text, k) if allm else s.search(pat, text, k)
    want = oracle.search(profile, pat, text, k, all_minima=allm)
    st = s.stats()
    desc["filtered"] = st["filtered"] * 10 + st["fused"]
    desc["matches"] = len(want)
    return key(got) == key(want), desc, pat, text, got, want


def bytes_long_case(rng, searchers):
    """The cliffs of round 2: Ascii patterns with more than 64 distinct bytes (byte mode) and patterns of 1 800 .. 5 000
    rows (fewer waves per workgroup)."""
    if rng.random() < 0.5:
        m = rng.choice([70, 100, 200, 256, 300])
        pat = bytes(rng.sample(range(256), min(m, 256))) + bytes(rng.randrange(256) for _ in range(max(0, m - 256)))
        k = rng.choice([0, 2, 5, 12])
        n = rng.choice([300, 5_000, 40_000])
        text = bytearray(rng.randrange(256) for _ in range(n))
        profile = "ascii"
    else:
        m = rng.choice([1800, 2048, 2500, 3333, 4096, 5000])
        beyond_lds = rng.random() < 0.06  # (round 6: per-row carries in global memory, beyond ~9 800 rows)
        if beyond_lds:
            m = rng.choice([9_900, 10_500, 12_000])
        profile = rng.choice(["dna", "iupac"])
        pat = rand_seq(rng, m, b"ACGT")
        k = rng.choice([0, 3, 10, 25, 40])
        n = rng.choice([m + 50, 20_000, 50_000])
        if beyond_lds:
            n = rng.choice([m + 50, 2 * m])
        text = bytearray(rand_seq(rng, n, b"ACGT"))
    for _ in range(rng.randrange(0, 4)):
        ins = mutate(rng, pat, rng.randrange(0, k + 2))
        if len(ins) + 2 < n:
            at = rng.choice([0, n - len(ins), rng.randrange(0, n - len(ins))])
            text[at:at + len(ins)] = ins
    text = bytes(text[:n])
    s = searchers[(profile, False)]
    got = s.search(pat, text, k)
    want = oracle.search(profile, pat, text, k)
    desc = dict(mode="bytes_long", profile=profile, m=m, k=k, n=n, rc=False, all_minima=False, filtered=s.stats()["filtered"],
                matches=len(want))
    return key(got) == key(want), desc, pat, text, got, want


def ovenc_case(rng, searchers):
    """search_encoded_patterns of an OVERHANG searcher on one text (round 6: one pass -- the seeded search for the inside, edge
    segments for the text's two ends; or a kernel chain per pattern where that does not apply): forward searchers against
    oracle.search_overhang pattern by pattern, patterns hanging over both ends."""
    m = rng.choice([12, 16, 20, 23, 24, 32, 40, 64])
    k = min(rng.choice([0, 1, 2, 3, 4]), m // 5)
    alpha = rng.choice([0.0, 0.25, 0.5, 0.5, 1.0])
    mo = rng.choice([None, None, 0, 3, m // 2])
    npat = rng.choice([1, 3, 4, 5, 9, 12])
    pal = b"ACGT" if rng.random() < 0.7 else b"ACGTNRYW"
    pats = [rand_seq(rng, m, pal) for _ in range(npat)]
    n = rng.choice([m + k + 60, m + k + 65, 200, 3_000, 20_000, 60_000])
    t = bytearray(rand_seq(rng, n, b"ACGT"))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for p in pats:
        plain = bytes(c if c in b"ACGT" else 65 for c in p)
        if rng.random() < 0.5:
            cut = rng.randrange(1, m)
            head = mutate(rng, plain, rng.randrange(0, 2))[cut:][:n]
            t[:len(head)] = head
        if rng.random() < 0.5:
            cut = rng.randrange(1, m)
            tail = mutate(rng, plain, rng.randrange(0, 2))[:cut][-n:]
            t[n - len(tail):] = tail
        if n > 4 * m:
            ins = mutate(rng, plain, rng.randrange(0, k + 2))
            at = rng.randrange(m, n - 2 * m)
            t[at:at + len(ins)] = ins
    if rng.random() < 0.15:  # a stray letter: the one pass declines, the chains run
        t[rng.randrange(n)] = ord("N")
    t = bytes(t[:n])
    allm = rng.random() < 0.25
    s = sassy_amd.Searcher("iupac", rc=False, alpha=alpha).with_max_overhang(mo)
    got = s.search_encoded_patterns(s.encode_patterns(pats), t, k, all_minima=allm)
    kk = lambda pi, x: (pi, x.text_start, x.text_end, x.pattern_start, x.pattern_end, x.cost, x.strand, x.cigar)
    gk = sorted(kk(x.pattern_idx, x) for x in got)
    wk = sorted(kk(pi, x) for pi, p in enumerate(pats)
                for x in oracle.search_overhang("iupac", p, t, k, alpha, all_minima=allm, max_overhang=mo))
    desc = dict(mode="ovenc", m=m, k=k, alpha=alpha, max_overhang=mo, npat=npat, n=n, all_minima=allm,
                filtered=s.stats()["filtered"], matches=len(wk))
    return gk == wk, desc, b"|".join(pats), t, gk, wk


def count_case(rng, searchers):
    """The q-gram counting filter and the one-pass two-strand marks on larger texts: patterns with
    ambiguity letters, stray non-ACGT text letters (forced windows), both strands."""
    profile = rng.choice(["iupac", "iupac", "dna"])
    m = rng.choice([24, 32, 40, 64, 90, 128, 200, 330])
    k = rng.choice([1, 2, 3, 5, 8, 12, 20, 33])
    k = max(0, min(k, m // 6 if profile == "iupac" else m // 9 + 1))
    if profile == "dna" and k < 8:
        k = max(k, 8 if m >= 90 else k)  # more than 8 pieces: the counting filter takes Dna too
    k = min(k, m - 1)
    n = rng.choice([100_000, 400_000, 1_000_000])
    pat = bytearray(rand_seq(rng, m, b"ACGT"))
    if profile == "iupac":
        for _ in range(rng.randrange(0, 4)):
            pat[rng.randrange(m)] = rng.choice(b"NRYSWKM")
    pat = bytes(pat)
    plain = bytes(c if c in b"ACGT" else 65 for c in pat)
    text = bytearray(rand_seq(rng, n, b"ACGT"))
    for _ in range(rng.randrange(2, 12)):
        src = plain if rng.random() < 0.5 else oracle.reverse_complement(profile, plain)
        ins = mutate(rng, src, rng.randrange(0, k + 2), spaced=rng.choice([0, 5, 6, 7]))
        at = rng.choice([0, n - len(ins), rng.randrange(0, n - len(ins))])
        text[at:at + len(ins)] = ins
    if profile == "iupac":
        for _ in range(rng.randrange(0, 30)):
            text[rng.randrange(n)] = rng.choice(b"NRYnacgtuU-")
    text = bytes(text)
    rc = rng.random() < 0.7
    allm = rng.random() < 0.2
    s = searchers[(profile, rc)]
    got = s.search_all(pat, text, k) if allm else s.search(pat, text, k)
    want = oracle.search(profile, pat, text, k, rc=rc, all_minima=allm)
    desc = dict(mode="count", profile=profile, m=m, k=k, n=n, rc=rc, all_minima=allm, filtered=s.stats()["filtered"],
                matches=len(want))
    return key(got) == key(want), desc, pat, text, got, want


def many_case(rng, searchers):
    """search_many: several patterns x several texts (batched / per-text paths) against per-pair oracle calls."""
    profile = rng.choice(["dna", "iupac", "iupac", "ascii"])
    rc = profile != "ascii" and rng.random() < 0.5
    allm = rng.random() < 0.2
    pal = {"dna": b"ACGT", "iupac": b"ACGTNRY", "ascii": b"ACGTXYZ "}[profile]
    npat = rng.randrange(1, 5)
    pats = []
    same = rng.random() < 0.5  # patterns of one length (<= 64) can go through the pattern-tiled scan in one pass
    if same:
        npat = rng.choice([1, 2, 3, 5, 70])
    m_same = rng.choice([8, 16, 20, 24, 32, 40, 64])
    for _ in range(npat):
        m = m_same if same else rng.choice([8, 16, 20, 24, 32, 40, 64, 100])
        pats.append(rand_seq(rng, m, pal if rng.random() < 0.3 else pal[:4]))
    k = rng.choice([0, 1, 2, 3, 5])
    k = min(k, min(len(p) for p in pats) - 1)
    texts = []
    for _ in range(rng.choice([1, 2, 5, 30, 200])):
        n = rng.choice([0, 1, 10, 63, 64, 65, 150, 300, 1000, 5000])
        t = bytearray(rand_seq(rng, n, b"ACGT"))
        if n > 120 and rng.random() < 0.7:
            p = rng.choice(pats)
            ins = mutate(rng, bytes(c if c in b"ACGT" else 65 for c in p), rng.randrange(0, k + 2))
            if len(ins) < n:
                at = rng.choice([0, n - len(ins), rng.randrange(0, n - len(ins) + 1)])
                t[at:at + len(ins)] = ins
        if profile == "iupac" and rng.random() < 0.3 and n:
            t[rng.randrange(n)] = rng.choice(b"NRYn")
        texts.append(bytes(t))
    s = searchers[(profile, rc)]
    force = rng.choice([None, "0", "1"])
    seed = rng.choice([None, "0", "1"])
    # overhang (Iupac): every text gets its own overhang column and virtual columns -- several patterns of one length go
    # through one pass per strand (tiled_pertext_kernel), else a launch per pattern
    alpha = None
    if profile == "iupac" and rng.random() < 0.35:
        alpha = rng.choice([0.0, 0.25, 0.5, 0.5, 1.0])
        mo = rng.choice([None, None, 0, 3])
        if same and rng.random() < 0.7:
            pats = pats + [rand_seq(rng, m_same, b"ACGT") for _ in range(rng.choice([3, 5, 66]))]
            npat = len(pats)
        s = sassy_amd.Searcher(profile, rc=rc, alpha=alpha).with_max_overhang(mo)
    # (the switches are per searcher, read from the environment when it is made: set_option on the one that searches)
    s.set_option("many_tiled", -1 if force is None else int(force))
    s.set_option("many_seeded", -1 if seed is None else int(seed))
    try:
        got = s.search_many(pats, texts, k, all_minima=allm)
    finally:
        s.set_option("many_tiled", -1)
        s.set_option("many_seeded", -1)
    gk = [(m.pattern_idx, m.text_idx, m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar)
          for m in got]
    wk = []
    for pi, p in enumerate(pats):
        for ti, t in enumerate(texts):
            ms_ = oracle.search(profile, p, t, k, rc=rc, all_minima=allm) if alpha is None else \
                oracle.search_overhang(profile, p, t, k, alpha, rc=rc, all_minima=allm, max_overhang=mo)
            for m in ms_:
                wk.append((pi, ti, m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar))
    desc = dict(mode="many", profile=profile, k=k, rc=rc, all_minima=allm, npat=npat, ntext=len(texts), tiled=force, alpha=alpha,
                filtered=s.stats()["filtered"], matches=len(wk))
    return sorted(gk) == sorted(wk), desc, b"|".join(pats), b"|".join(texts), gk, wk


def encoded_case(rng, searchers):
    """search_encoded_patterns: the pattern-tiled one-pass scan, or (SASSY_HIP_TILED=0, long texts) one scan per
    pattern incl. the multi-pattern prefilter."""
    profile = rng.choice(["dna", "iupac"])
    rc = rng.random() < 0.5
    wide = rng.random() < 0.5  # the wider shapes only the one-pass scan sees often enough otherwise
    m = rng.choice([1, 2, 7, 12, 16, 20, 23, 24, 31, 32, 33, 40, 63, 64]) if wide else rng.choice([12, 16, 20, 23, 24, 32, 40])
    k = rng.choice([0, 1, 2, 3, 5, 8, 12]) if wide else min(rng.choice([0, 1, 2, 3]), m // 6)
    if wide and k > m + 2:
        k = m + 2
    npat = rng.choice([1, 2, 3, 8, 9, 40, 64, 65, 70, 130, 300])
    pal = b"ACGT" if profile == "dna" or rng.random() < 0.6 else b"ACGTNRYKMSW"
    pats = [rand_seq(rng, m, pal) for _ in range(npat)]
    n = rng.choice([1, 5, 70, 200, 3000, 20_000, 60_000])
    if wide and (k >= m // 2 or (k >= m // 3 and pal != b"ACGT")):
        n = min(n, 3000)  # nearly every position is a report: keep the oracle's work bounded
    t = bytearray(rand_seq(rng, n, b"ACGT"))
    for p in pats[:20]:
        ins = bytearray(mutate(rng, p, rng.randrange(0, k + 2)))
        for i, c in enumerate(ins):
            if chr(c) not in "ACGT":
                ins[i] = rng.choice(b"ACGT")
        if len(ins) < n:
            at = rng.randrange(0, n - len(ins) + 1)
            t[at:at + len(ins)] = ins
    if rng.random() < 0.2:
        for _ in range(20):
            i = rng.randrange(n); t[i] = t[i] | 0x20
    if rng.random() < 0.3:
        for _ in range(rng.choice([1, 30])):
            t[rng.randrange(n)] = rng.choice(b"NRYn-*" if profile == "iupac" else b"NX-n")
    if profile == "iupac" and n >= 200 and rng.random() < 0.3:  # runs of N (the seeded search cuts long ones out)
        for _ in range(rng.choice([1, 2, 5])):
            ln = rng.choice([1, m, m + 1, m + 2, m + 3, 2 * (m + k) + 2, 200, 2000])
            ln = min(ln, n // 2)
            at = rng.choice([0, n - ln, rng.randrange(0, n - ln + 1)])
            t[at:at + ln] = b"N" * ln
            if rng.random() < 0.2:
                t[at + ln // 2] = ord("Y")
    t = bytes(t)
    allm = wide and rng.random() < 0.3
    force = rng.choice([None, None, "0", "1", "seed", "seed"])
    s = searchers[(profile, rc)]
    s.set_option("tiled", -1)
    s.set_option("seeded", -1)
    if force == "seed":  # seed -> verify -> report wherever the shape allows it
        s.set_option("seeded", 1)
    elif force is not None:
        s.set_option("tiled", int(force))
    enc = s.encode_patterns(pats)
    # (Dna text with other letters: the scan's 2-bit equality and the traceback's letter equality can disagree;
    # the reference panics in get_trace, the oracle raises, and so must the library)
    try:
        want = oracle.search_encoded(profile, pats, t, k, rc=rc, all_minima=allm)
    except RuntimeError:
        want = None
    try:
        got = s.search_encoded_patterns(enc, t, k, all_minima=allm)
    except Exception as e:
        got = None
        if want is not None or "traceback failed" not in str(e) or not isinstance(e, sassy_amd.SassyHipError):
            print("FAILED CALL", dict(profile=profile, m=m, k=k, rc=rc, npat=npat, n=n, all_minima=allm, tiled=force))
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail.bin"), "wb") as fh:
                fh.write(b"|".join(pats) + b"\n" + t)
            raise
    s.set_option("tiled", -1)
    s.set_option("seeded", -1)
    if want is None or got is None:
        desc = dict(mode="encoded", profile=profile, m=m, k=k, rc=rc, npat=npat, n=n, all_minima=allm, tiled=force,
                    filtered=s.stats()["filtered"], matches=0)
        return want is None and got is None, desc, b"|".join(pats), t, [], []
    kk = lambda m: (m.pattern_idx, m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar)
    gk, wk = sorted(kk(m) for m in got), sorted(kk(m) for m in want)
    desc = dict(mode="encoded", profile=profile, m=m, k=k, rc=rc, npat=npat, n=n, all_minima=allm, tiled=force,
                filtered=s.stats()["filtered"], matches=len(wk))
    return gk == wk, desc, b"|".join(pats), t, gk, wk


def shard_case(rng, searchers):
    """search_shard over random 64-aligned cuts of a device text, chain resolved like multigpu does."""
    profile = rng.choice(["dna", "iupac"])
    m = rng.choice([8, 20, 32, 33, 64, 100])
    k = min(rng.choice([0, 1, 3, 5, 8]), m // 4)
    n = rng.choice([3000, 20_000, 100_000])
    pat = rand_seq(rng, m, b"ACGT")
    if rng.random() < 0.3:
        unit = rand_seq(rng, rng.randrange(1, 4), b"ACGT")
        pat = (unit * (m // len(unit) + 1))[:m]
        t = bytearray((unit * (n // len(unit) + 1))[:n])
        for _ in range(n // 300):
            t[rng.randrange(n)] = rng.choice(b"ACGT")
    else:
        t = bytearray(rand_seq(rng, n, b"ACGT"))
    cuts = sorted({64 * rng.randrange(1, n // 64) for _ in range(rng.randrange(1, 6))})
    for c in cuts:  # near-matches across the cuts
        ins = mutate(rng, pat, rng.randrange(0, k + 1))
        at = max(0, min(n - len(ins), c - rng.randrange(0, len(ins) + 1)))
        t[at:at + len(ins)] = ins
    t = bytes(t)
    halo = sassy_amd.required_halo(m, k)
    cuts = [c for c in cuts if c >= halo] or [64 * ((n // 2) // 64)]
    cuts = [c for c in cuts if c >= halo]
    bounds = [0] + cuts + [n]
    buf = sassy_amd.DeviceBuffer(n + 256)
    buf.upload(t)
    s = searchers[(profile, False)]
    allm = []
    prev_state = 1
    for a, b in zip(bounds[:-1], bounds[1:]):
        h = 0 if a == 0 else halo
        r = s.search_shard(pat, buf.ptr + a - h, h, b - a, a, n, k)
        ms = list(r.matches)
        if r.conditional_index >= 0 and prev_state != 1:
            del ms[r.conditional_index]
        allm += ms
        if r.exit_state != 2:
            prev_state = r.exit_state
    buf.free()
    want = oracle.search(profile, pat, t, k)
    desc = dict(mode="shard", profile=profile, m=m, k=k, n=n, bounds=bounds, filtered=s.stats()["filtered"], matches=len(want))
    return key(allm) == key(want), desc, pat, t, allm, want


def inflight_case(rng, searchers):
    """Searches in flight (sassy_hip_search_shard_begin / _finish): a burst of random searches over one resident
    text, two or three in flight, finished in a random order, each against the oracle."""
    profile = rng.choice(["dna", "iupac"])
    n = rng.choice([5000, 70_000, 400_000])
    comp = rng.random()
    if comp < 0.6:
        t = bytearray(rand_seq(rng, n, b"ACGT"))
    else:
        unit = rand_seq(rng, rng.randrange(1, 30), b"ACGT")
        t = bytearray((unit * (n // len(unit) + 1))[:n])
        for _ in range(n // 150):
            t[rng.randrange(n)] = rng.choice(b"ACGT")
    jobs = []
    for _ in range(rng.randrange(2, 7)):
        m = rng.choice([8, 20, 32, 33, 64, 100, 200])
        k = min(rng.choice([0, 1, 3, 5, 8, 20]), m // 4)
        pat = rand_seq(rng, m, b"ACGT")
        if comp >= 0.6 and rng.random() < 0.5:
            pat = (unit * (m // len(unit) + 1))[:m]
        for _ in range(rng.randrange(0, 4)):
            ins = mutate(rng, pat, rng.randrange(0, k + 1))
            at = rng.randrange(0, n - len(ins))
            t[at:at + len(ins)] = ins
        jobs.append((pat, k, rng.random() < 0.2))
    t = bytes(t)
    buf = sassy_amd.DeviceBuffer(n + 256)
    buf.upload(t)
    depth = rng.choice([2, 3])
    s = sassy_amd.Searcher(profile, rc=False).set_pipe_depth(depth)
    got, pending = {}, []
    for i, (pat, k, allm) in enumerate(jobs):
        pending.append((i, s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, k, sassy_amd.ALL_MINIMA if allm else 0)))
        if len(pending) == depth:
            j, tk = pending.pop(rng.randrange(len(pending)))
            got[j] = s.search_finish(tk).matches
    while pending:
        j, tk = pending.pop(rng.randrange(len(pending)))
        got[j] = s.search_finish(tk).matches
    buf.free()
    ok, nm = True, 0
    bad = None
    for i, (pat, k, allm) in enumerate(jobs):
        want = oracle.search(profile, pat, t, k, all_minima=allm)
        nm += len(want)
        if key(got[i]) != key(want):
            ok, bad = False, (pat, got[i], want)
    desc = dict(mode="inflight", profile=profile, n=n, jobs=[(len(p), k, a) for p, k, a in jobs], depth=depth,
                filtered=s.stats()["filtered"], matches=nm)
    if not ok:
        return False, desc, bad[0], t, bad[1], bad[2]
    return True, desc, b"", t, [], []


def reflanes_case(rng, searchers):
    """The reference-lane reports mode against the reference-shaped port (4 and 8 lanes), periodic and random text."""
    lanes = rng.choice([4, 8])
    m = rng.choice([8, 12, 20, 32, 40, 60, 70, 90, 130])
    k = min(rng.choice([0, 1, 2, 3, 5, 8]), m - 1)
    style = rng.randrange(3)
    if style == 0:
        sep = rng.choice([b"C", b"CG", b"CCC"])
        per = rng.choice([m - 1, m, m + 1, max(1, m // 2), 2 * m])
        text = b"A" * rng.randrange(0, 200) + (sep + b"A" * per) * rng.randrange(5, 120) + b"G" * rng.randrange(0, 100)
        pat = b"A" * m
    elif style == 1:
        unit = rand_seq(rng, rng.choice([1, 2, 3, 5]), b"ACGT")
        n = rng.choice([100, 257, 1003, 5000, 30000])
        text, pat = (unit * (n // len(unit) + 1))[:n], (unit * (m // len(unit) + 1))[:m]
    else:
        pat = rand_seq(rng, m, b"ACGT")
        text = bytearray(rand_seq(rng, rng.choice([64, 500, 3000, 20000, 100000]), b"ACGT"))
        for _ in range(3):
            if len(text) > m + 5:
                at = rng.randrange(0, len(text) - m)
                text[at:at + m] = pat
        text = bytes(text)
    s = sassy_amd.Searcher("dna", rc=False).set_reference_lanes(lanes)
    got = [(x.text_end, x.cost) for x in s.search(pat, text, k)]
    want, _ = oracle.refstyle_ends("dna", pat, text, k, lanes=lanes)
    desc = dict(mode="reflanes", lanes=lanes, m=m, k=k, n=len(text), style=style, filtered=s.stats()["filtered"], matches=len(want))
    return got == want, desc, pat, text, got, want


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    families = {"one": one_case, "fused": fused_case, "bytes_long": bytes_long_case, "count": count_case, "many": many_case,
                "encoded": encoded_case, "shard": shard_case, "inflight": inflight_case, "reflanes": reflanes_case,
                "ovenc": ovenc_case}
    ap.add_argument("--focus", default="", choices=[""] + sorted(families),
                    help="one case family only (default: the mix); 'count': larger texts through the q-gram counting filter, both strands")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    searchers = {(p, rc): sassy_amd.Searcher(p, rc=rc) for p in ("dna", "iupac", "ascii") for rc in (False, True)
                 if not (p == "ascii" and rc)}
    t0 = time.time()
    cases = 0
    kinds = {}
    total_matches = 0
    while time.time() - t0 < args.seconds:
        mode = rng.random()
        fn = (one_case if mode < 0.37 else ovenc_case if mode < 0.4 else fused_case if mode < 0.52 else bytes_long_case if mode < 0.56 else many_case
              if mode < 0.66 else encoded_case if mode < 0.76 else shard_case if mode < 0.85 else inflight_case
              if mode < 0.93 else reflanes_case)
        if args.focus:
            fn = families[args.focus]
        ok, desc, pat, text, got, want = fn(rng, searchers)
        cases += 1
        kinds[desc["filtered"]] = kinds.get(desc["filtered"], 0) + 1
        total_matches += desc["matches"]
        if not ok:
            print("MISMATCH", desc)
            print("pattern", pat)
            gk, wk = (key(got), key(want)) if desc.get("mode") in (None, "shard", "count", "inflight", "fused", "bytes_long") else (got, want)
            print("got", len(gk), "want", len(wk))
            extra = [x for x in gk if x not in set(wk)][:5]
            missing = [x for x in wk if x not in set(gk)][:5]
            print("extra", extra)
            print("missing", missing)
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail.bin"), "wb") as fh:
                fh.write(repr(desc).encode() + b"\n" + pat + b"\n" + text)
            sys.exit(1)
    print(f"fuzz ok: {cases} cases, {total_matches} matches compared, prefilter kinds used {kinds}, seed {args.seed}")


if __name__ == "__main__":
    main()
