"""Pins the CPU oracle against every known-answer test the reference holds for the search path
(tests/golden/kats.json <- SURVEY.md App. B).  CPU only."""
import pytest

import oracle
from conftest import build_block, build_text, cigar_path, expand_positions


def _check_fields(m, exp):
    for f, v in exp.items():
        if f == "path":
            assert cigar_path(m) == [tuple(x) for x in v]
        else:
            assert getattr(m, f) == v, (f, getattr(m, f), v, m)


def _search_ids(kats):
    return [e["id"] for e in kats["search"]]


def test_search_kats(kats):
    assert len(kats["search"]) >= 11
    for e in kats["search"]:
        text = build_text(e)
        ms = oracle.search(e["profile"], e["pattern"].encode(), text, e["k"], rc=e["rc"],
                           all_minima=(e["mode"] == "search_all"))
        if "expect_len" in e:
            assert len(ms) == e["expect_len"], (e["id"], ms)
        for m, exp in zip(ms, e.get("expect", [])):
            _check_fields(m, exp)
        if "expect_first" in e:
            _check_fields(ms[0], e["expect_first"])
        if "expect_text_ends" in e:
            assert [m.text_end for m in ms] == e["expect_text_ends"], e["id"]


def test_not_rev_invariant(kats):
    e = kats["not_rev_invariant"]
    p, t = e["pattern"].encode(), e["text"].encode()
    a = oracle.search(e["profile"], p, t, e["k"])
    b = oracle.search(e["profile"], p[::-1], t[::-1], e["k"])
    assert len(a) != len(b)
    assert (len(a), len(b)) == (1, 2)  # values reproduced by the survey's scratch model (App. B #12)


def test_encoded_kats(kats):
    for e in kats["encoded"]:
        ms = oracle.search_encoded(e["profile"], [p.encode() for p in e["patterns"]],
                                   e["text"].encode(), e["k"], rc=e["rc"],
                                   all_minima=e.get("all", False))
        assert len(ms) == e["expect_len"], (e["id"], ms)
        if "expect_text_starts_in_order" in e:
            assert [m.text_start for m in ms] == e["expect_text_starts_in_order"]
        for pidx, exp in e.get("expect_by_pattern", {}).items():
            m = [x for x in ms if x.pattern_idx == int(pidx)][0]
            _check_fields(m, exp)


def test_range_kat(kats):
    for e in kats["ranges"]:
        C = oracle.last_row(e["profile"], e["pattern"].encode(), e["text"].encode())
        # char index c (0-based) <-> end position c+1
        idx = [i - 1 for i in range(1, len(C)) if C[i] <= e["k"]]
        first = [idx[0], idx[0]]
        for a in idx[1:]:
            if a == first[1] + 1:
                first[1] = a
            else:
                break
        assert first == e["expect_first_range"]


def test_profile_mask_kats(kats):
    for e in kats["profile_masks"]:
        masks = oracle.profile_masks(e["profile"], e["pattern"].encode(), build_block(e["block"]))
        for slot, exp in e["expect_positions"].items():
            got = [b for b in range(64) if (masks[int(slot)] >> b) & 1]
            assert got == expand_positions(exp), (e["id"], slot)


def test_cigar_format(kats):
    assert oracle.rle_cigar(b"===X") == "3=1X"
    assert oracle.rle_cigar(b"==X=") == "2=1X1="
    assert oracle.rle_cigar(b"") == ""
    for s in kats["cigar_strings"]["examples"]:
        import re
        ops = b"".join(op.encode() * int(c) for c, op in re.findall(r"(\d+)([=XID])", s))
        assert oracle.rle_cigar(ops) == s


def test_iupac_pattern_validity():
    # iupac.rs:156-204: letters with a code; X (empty set) is valid, Z / digits are not
    assert oracle.valid_pattern("iupac", b"ACGTNRYSWKMBDHVXacgtn")
    assert oracle.valid_pattern("iupac", b"U")
    for bad in [b"Z", b"1", b"E", b"@", b"[", b"A C"]:
        assert not oracle.valid_pattern("iupac", bad)
    assert oracle.valid_pattern("dna", b"anything goes")


def test_complements():
    assert oracle.reverse_complement("dna", b"ATCGATCA") == b"TGATCGAT"
    assert oracle.complement("dna", b"ACGTacgtN") == b"TGCAacgtN"  # dna.rs:121-133: upper case only
    assert oracle.complement("iupac", b"ACGTRYSWKMBDHVNXacgtrykm") == b"TGCAYRSWMKVHDBNXtgcayrmk"


def _end_filters(pattern_fwd: bytes):
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    return {
        # the closures of the reference's tests (src/search.rs:2552, :2590-2595)
        "text_len_gt_10_plus_m": lambda q, t, strand: len(t) > 10 + len(q),
        "suffix_is_pattern_fwd_or_complement":
            lambda q, t, strand: (t[len(t) - len(q):] if strand == "+" else t[len(t) - len(q):].translate(comp)) == pattern_fwd,
    }


def test_mode_kats(kats):
    """search_with_fn / max_n_frac known answers against the oracle's restatement of the modes."""
    for e in kats["modes"]:
        pat = e["pattern"].encode()
        text = build_text(e)
        ms = oracle.search_modes(e["profile"], pat, text, e["k"], rc=e["rc"], all_minima=e["mode"] == "search_all",
                                 end_filter=_end_filters(pat)[e["end_filter"]] if "end_filter" in e else None,
                                 max_n_frac=e.get("max_n_frac"))
        if "expect_text_start" in e:
            assert [m.text_start for m in ms] == e["expect_text_start"], (e["id"], ms)
        if "expect_text_end" in e:
            assert [m.text_end for m in ms] == e["expect_text_end"], (e["id"], ms)


def test_overhang_kats(kats):
    """Overhang known answers (lib.rs doctest, the two trace-path tests, the N-filter test)."""
    for e in kats["overhang"]:
        pat, text = e["pattern"].encode(), e["text"].encode()
        ms = oracle.search_overhang(e["profile"], pat, text, e["k"], e["alpha"], rc=e["rc"],
                                    all_minima=e["mode"] == "search_all")
        if "expect" in e:
            assert len(ms) == len(e["expect"]), (e["id"], ms)
            for m, x in zip(ms, e["expect"]):
                for f, v in x.items():
                    assert getattr(m, f) == v, (e["id"], f, m)
        if "expect_len" in e:
            assert len(ms) == e["expect_len"], (e["id"], ms)
        if "expect_first" in e:
            m, x = ms[0], e["expect_first"]
            for f in ("pattern_start", "pattern_end", "text_start", "text_end"):
                if f in x:
                    assert getattr(m, f) == x[f], (e["id"], f, m)
            # Match::to_path (src/search.rs:83-103): the cells of the alignment inside the text
            path, j, i = [], m.pattern_start, m.text_start
            import re
            for cnt, op in re.findall(r"(\d+)([=XID])", m.cigar):
                for _ in range(int(cnt)):
                    if op in "=X":
                        path.append([j, i]); j += 1; i += 1
                    elif op == "I":
                        path.append([j, i]); j += 1
                    else:
                        path.append([j, i]); i += 1
            assert path == x["path"], (e["id"], path)
