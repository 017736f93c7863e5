"""Checks the properties of tests/golden/kats_more.json (the reference's own assertions, harvested by
tools/harvest_reference_kats.py) against an engine: engine(profile, rc, alpha, max_n_frac, all_minima, pattern, text, k)
-> list of matches with the reference's Match fields.  Used with the oracle (CPU suite) and with the HIP path (-m gpu)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_COMP = bytes.maketrans(b"ACGTRYSWKMBDHVNacgtryswkmbdhvn", b"TGCAYRSWMKVHDBNtgcayrswmkvhdbn")


def revcomp(seq: bytes) -> bytes:
    """Iupac::reverse_complement (src/profiles/iupac.rs:140-148 over the table :332-344)"""
    return seq.translate(_COMP)[::-1]


def load():
    with open(os.path.join(ROOT, "tests", "golden", "kats_more.json")) as f:
        return json.load(f)


def to_path(m):
    """Match::to_path (src/search.rs:80-103), both strands"""
    rc = m.strand in ("-", 1)
    pos = [m.pattern_start, (m.text_end - 1) if rc else m.text_start]
    sign = -1 if rc else 1
    path = [tuple(pos)]
    for cnt, op in re.findall(r"(\d+)([=XID])", m.cigar):
        for _ in range(int(cnt)):
            if op in "=X":
                pos[0] += 1
                pos[1] += sign
            elif op == "I":
                pos[0] += 1
            else:
                pos[1] += sign
            path.append(tuple(pos))
    path.pop()
    return path


def check(e, engine):
    pat, text, k = e["pattern"].encode(), e["text"].encode(), e["k"]
    if e.get("revcomp_pattern"):
        pat = revcomp(pat)
    if e.get("revcomp_text"):
        text = revcomp(text)
    n, m = len(text), len(pat)
    allm = e["mode"] == "search_all"
    run = lambda p=pat, t=text, rc=e["rc"], nf=e.get("max_n_frac"): engine(e["profile"], rc, e.get("alpha"), nf, allm, p, t, k)
    ms = run()
    prop = e["prop"]
    # what the reference asserts for every match it returns (src/search.rs:1672-1685)
    for x in ms:
        assert 0 <= x.cost <= k, (e["id"], x)
        assert x.text_start <= x.text_end <= n and x.pattern_start <= x.pattern_end <= m, (e["id"], x)
    if "expect_len" in e:
        assert len(ms) == e["expect_len"], (e["id"], ms)
    if prop == "no_panic":
        return ms
    if prop == "len":
        assert len(ms) == e["n"], (e["id"], ms)
    elif prop == "nonempty":
        assert len(ms) > 0, e["id"]
    elif prop == "exists_start_within":
        assert any(abs(x.text_start - e["at"]) <= e["tol"] for x in ms), (e["id"], ms)
    elif prop == "starts_present":
        have = {x.text_start for x in ms}
        assert all(s in have for s in e["starts"]), (e["id"], sorted(have)[:20])
    elif prop == "exists_end":
        for want in [e] + e.get("also", []):
            end, cost = want["end"], want["cost"]
            ok = [x for x in ms if x.text_end == min(end, n) and x.pattern_end == m - max(0, end - n) and
                  (x.cost <= cost if e.get("cost_cmp") == "<=" else x.cost == cost)]
            assert ok, (e["id"], end, ms)
    elif prop == "exists_text_end_cost":
        ok = [x for x in ms if x.text_end == e["end"] and x.cost == e["cost"] and
              ("pattern_end" not in e or x.pattern_end == e["pattern_end"])]
        assert ok, (e["id"], ms)
    elif prop == "same_as_rc_pattern":
        other = run(p=revcomp(pat))
        assert len(ms) == len(other), (e["id"], ms, other)
        for x in ms:
            assert any((y.text_start, y.text_end, y.cost) == (x.text_start, x.text_end, x.cost) for y in other), (e["id"], x)
    elif prop == "cigar_equal_under_rc_text":
        other = run(t=revcomp(text), rc=True)
        assert ms[0].cigar == other[0].cigar, (e["id"], ms[0], other[0])
    elif prop == "same_len_with_n_frac":
        other = run(nf=e["max_n_frac_alt"])
        assert len(ms) == len(other), (e["id"], ms, other)
    elif prop == "rc_path_prefix_complements":
        for q, r in to_path(ms[0])[:e["take"]]:
            assert pat[q:q + 1] == revcomp(text[r:r + 1]), (e["id"], q, r, ms[0])
    else:
        raise AssertionError("unknown property " + prop)
    return ms
