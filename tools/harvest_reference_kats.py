"""Harvests the inputs and asserted values of the reference's own tests for the search path into
tests/golden/kats_more.json (data only: byte literals, k, searcher configuration and WHAT the test asserts,
as a property the test suites re-check against the oracle and, on the GPU box, against the HIP path).

Runs in the build container only (/root/reference does not exist on the GPU box); the JSON is committed.

    python tools/harvest_reference_kats.py

Every entry: id, source (file:line of the reference test), profile, rc, alpha (null = no overhang), mode
(search | search_all), pattern, text, k, and `prop` = the reference's assertion:
  no_panic                 the test only runs the search (the reference asserts cost <= k internally, src/search.rs:1672-1685)
  len                      number of matches == n
  nonempty                 at least one match
  exists_start_within      some match has |text_start - at| <= tol
  starts_present           for every x of `starts` some match has text_start == x
  exists_end               some match ends at virtual end position `end` (text_end == min(end, n), pattern_end ==
                           m - max(0, end - n)) with cost == / <= `cost` (`cost_cmp`)
  exists_text_end_cost     some match has text_end == end and cost == cost (+ pattern_end when given)
  same_as_rc_pattern       search(pattern) and search(revcomp(pattern)) have equal length and every match of the first
                           has a twin (text_start, text_end, cost) in the second
  cigar_equal_under_rc_text  first cigar of fwd-searcher search(p, t) == first cigar of rc-searcher search(p, revcomp(t))
  first                    fields of the first match
  same_len_with_n_frac     the number of matches does not change with max_n_frac = f
"""
import json
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fn_body(path, name):
    src = open(os.path.join(REF, path)).read().split("\n")
    for i, line in enumerate(src):
        if re.search(r"\bfn %s\(\)" % re.escape(name), line):
            depth, j = 0, i
            started = False
            while j < len(src):
                depth += src[j].count("{") - src[j].count("}")
                if "{" in src[j]:
                    started = True
                if started and depth == 0:
                    break
                j += 1
            return i + 1, j + 1, "\n".join(src[i:j + 1])
    raise KeyError((path, name))


def literals(body):
    """byte literals b"..." and string literals bound by `let name = "..."` of a function body, in source order
    (comments stripped; format strings of println! / assert! are not data)"""
    out = []
    code = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    code = "\n".join(l.split("//")[0] for l in code.split("\n"))
    for m in re.finditer(r'(b"((?:[^"\\]|\\.)*)")|(let\s+(?:mut\s+)?\w+\s*(?::[^=]+)?=\s*"((?:[^"\\]|\\.)*)")', code):
        out.append(m.group(2) if m.group(1) else m.group(4))
    return out


S = "src/search.rs"
PT = "src/pattern_tiling/search.rs"
# (file, fn, id suffix, dict of fields; "pat": index of the pattern literal, "txt": index of the text literal)
SPECS = [
    (S, "overshoot_test_prefix_trace", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=10, prop="no_panic")),
    (S, "overshoot_simple_prefix", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=2, prop="exists_end", end=3, cost=2, cost_cmp="<=")),
    (S, "overshoot_simple_suffix", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=2, prop="exists_end", end=24, cost=2, cost_cmp="<=")),
    (S, "overshoot_simple_suffix_local_minima", dict(profile="iupac", alpha=0.5, mode="search", pat=0, txt=1, k=4, prop="exists_text_end_cost", end=20, pattern_end=3, cost=2, expect_len=2)),
    (S, "overshoot_test_prefix_and_suffix", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=2, prop="exists_end", end=3, cost=2, cost_cmp="==", also=[dict(end=13, cost=2)])),
    (S, "overshoot", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=10, prop="no_panic")),
    (S, "overhang_test", dict(profile="iupac", alpha=0.0, mode="search_all", pat=0, txt=1, k=100, prop="no_panic")),
    (S, "test_case1", dict(profile="dna", rc=True, mode="search", pat=0, txt=1, k=2, prop="no_panic")),
    (S, "no_extra_matches", dict(profile="dna", mode="search", pat=0, txt=1, k=6, prop="exists_start_within", at=277, tol=6)),
    (S, "print_matches", dict(profile="dna", rc=True, mode="search_all", pat=0, txt=1, k=1, prop="no_panic")),
    (S, "print_matches", dict(suffix="local", profile="dna", rc=True, mode="search", pat=0, txt=1, k=1, prop="no_panic")),
    (S, "test_fixed_matches", dict(profile="dna", mode="search_all", pat=0, k=1, prop="starts_present", starts=[50, 150, 250, 350, 450, 800],
                                   text_build=dict(fill="G", len=1000, overwrites=[50, 150, 250, 350, 450, 800]))),
    (S, "test_pattern_trace_path_0_edits_rc", dict(profile="dna", rc=True, mode="search", pat=0, txt=1, k=1, prop="rc_path_prefix_complements", take=4)),
    (S, "test_case3", dict(profile="iupac", alpha=0.4, mode="search", pat=0, txt=1, k=63, prop="no_panic")),
    (S, "test_case4", dict(profile="iupac", alpha=0.5, mode="search", pat=0, txt=1, k=3, prop="exists_text_end_cost", end=1, cost=1)),
    (S, "test_case4", dict(suffix="all", profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=3, prop="exists_text_end_cost", end=1, cost=1)),
    (S, "test_match_exact_at_end", dict(profile="iupac", alpha=0.5, mode="search", pat=0, txt=1, k=0, prop="no_panic")),
    (S, "test_match_exact_at_end", dict(suffix="all", profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=0, prop="no_panic")),
    (S, "fwd_rc_test_simple", dict(profile="iupac", rc=True, alpha=0.5, mode="search", pat=0, txt=1, k=0, prop="same_as_rc_pattern")),
    (S, "fwd_rc_test", dict(profile="iupac", rc=True, mode="search", pat=0, txt=1, k=20, prop="same_as_rc_pattern")),
    (S, "search_bug_2", dict(profile="dna", mode="search", pat=0, txt=1, k=1, prop="exists_start_within", at=436, tol=1)),
    (S, "search_bug_3", dict(profile="dna", mode="search", pat=0, txt=1, k=18, prop="exists_start_within", at=3, tol=18)),
    (S, "original_rc_bug", dict(profile="iupac", rc=True, mode="search", pat=0, txt=1, k=44, prop="no_panic")),
    (S, "original_rc_bug", dict(suffix="rcpat", profile="iupac", rc=True, mode="search", pat=0, txt=1, k=44, prop="no_panic", revcomp_pattern=True)),
    (S, "test_cigar_invariant_under_rc_text", dict(profile="dna", mode="search", pat=0, txt=1, k=1, prop="cigar_equal_under_rc_text")),
    (S, "test_cigar_rc_at_overhang_end", dict(profile="iupac", rc=True, alpha=0.5, mode="search", pat=0, txt=1, k=1, prop="nonempty")),
    (S, "test_cigar_rc_at_overhang_end", dict(suffix="rcpat", profile="iupac", rc=True, alpha=0.5, mode="search", pat=0, txt=1, k=1, prop="nonempty", revcomp_pattern=True)),
    (S, "real_data_bug", dict(profile="iupac", rc=True, alpha=0.5, mode="search", pat=0, txt=1, k=45, prop="no_panic")),
    (S, "test_simple_ascii", dict(profile="ascii", mode="search", pat=0, txt=1, k=1, prop="no_panic")),
    (S, "test_reported_start_end", dict(profile="iupac", mode="search", pat=0, k=2, prop="no_panic",
                                        text_build=dict(fill="G", len=64, splices=[[20, "revcomp:1"], [50, "lit:1"]]))),
    (S, "test_reported_start_end", dict(suffix="rc", profile="iupac", rc=True, mode="search", pat=0, k=2, prop="no_panic", revcomp_pattern=True,
                                        text_build=dict(fill="G", len=64, splices=[[20, "revcomp:1"], [50, "lit:1"]]))),
    (S, "test_searchable_slice", dict(profile="iupac", rc=True, mode="search", pat=0, txt=1, k=0, prop="nonempty")),
    (S, "diff_rc_result", dict(profile="iupac", rc=True, alpha=0.5, mode="search", pat=1, txt=0, k=12, prop="no_panic")),
    (S, "diff_rc_result", dict(suffix="rctext", profile="iupac", rc=True, alpha=0.5, mode="search", pat=1, txt=0, k=12, prop="no_panic", revcomp_text=True)),
    (S, "search_slice", dict(profile="iupac", rc=True, alpha=0.5, mode="search", pat=1, txt=0, k=1, prop="no_panic")),
    (S, "double_match_search_all", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=3, prop="no_panic")),
    (S, "n_frac_prefilter_dense_n_skipped_fwd", dict(profile="iupac", mode="search_all", pat=0, txt=1, k=2, max_n_frac=0.5, prop="len", n=0)),
    (S, "n_frac_prefilter_dense_n_skipped_rc", dict(profile="iupac", rc=True, mode="search_all", pat=0, txt=1, k=2, max_n_frac=0.5, prop="len", n=0)),
    ("src/n_filter.rs", "n_filter_fuzz_case", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=3, prop="same_len_with_n_frac", max_n_frac_alt=0.13340974)),
    ("src/n_filter.rs", "n_filter_complex_example", dict(profile="iupac", mode="search_all", pat=0, txt=1, k=1, prop="len", n=6)),
    (S, "check_iupac_comparison_used", dict(profile="iupac", mode="search_all", pat=1, txt=0, k=2, prop="nonempty")),
    # v2 / pattern tiling: single patterns through the v1 entry points (the tests print both and compare by eye)
    (PT, "test_alpha_overhang", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=2, prop="nonempty")),
    (PT, "test_prefix_overhang", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=2, prop="nonempty")),
    (PT, "test_no_matches", dict(profile="iupac", mode="search_all", pat=0, txt=1, k=1, prop="len", n=0)),
    (PT, "pattern_tiling_trace_bug", dict(profile="iupac", mode="search_all", pat=0, txt=1, k=1, prop="no_panic")),
    (PT, "pattern_tiling_trace_bug", dict(suffix="rcpat", profile="iupac", mode="search_all", pat=0, txt=1, k=1, prop="no_panic", revcomp_pattern=True)),
    (PT, "pattern_tiling_test", dict(profile="iupac", alpha=0.5, mode="search", pat=0, txt=1, k=3, prop="no_panic")),
    (PT, "test_sassy_bug", dict(profile="iupac", alpha=0.5, mode="search_all", pat=1, txt=0, k=3, prop="no_panic")),
    (PT, "mini_trace_bug", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=2, prop="no_panic")),
    (PT, "prefix_bug_using_usize", dict(profile="iupac", alpha=0.5, mode="search_all", pat=0, txt=1, k=3, prop="no_panic")),
    ("src/trace.rs", "test_traceback", dict(profile="dna", mode="search_all", pat=0, txt=1, k=17, prop="no_panic")),
    ("src/trace.rs", "test_traceback_simd", dict(profile="dna", mode="search_all", pat=0, txt=1, k=17, prop="no_panic")),
    ("src/trace.rs", "test_traceback_simd", dict(suffix="t3", profile="dna", mode="search_all", pat=0, txt=3, k=17, prop="no_panic")),
    ("src/trace.rs", "test_traceback_simd", dict(suffix="t4", profile="dna", mode="search_all", pat=0, txt=4, k=17, prop="no_panic")),
]

# encoded-pattern (v2) entries
ENC_SPECS = [
    (PT, "test_batch_size_edge_case", dict(profile="iupac", patterns=["AAAA", "CCCC", "GGGG", "TTTT"], txt=0, k=2, all=True, prop="nonempty",
                                           note="TestBackend::LANES patterns in the reference (4 with U64); the four distinct ones here")),
    (S, "test_pattern_tilling_profiles", dict(suffix="dna", profile="dna", patterns_lit=[0], txt=1, k=0, all=False, prop="len", n=0)),
]

# the Ascii profile's mask tests (src/profiles/ascii.rs:140-185): block "ElLo" + 60 x 'H', slots H l o
ASCII_MASKS = [
    dict(id="ascii_u64_search", source="src/profiles/ascii.rs:160-168", profile="ascii", pattern="Hlo",
         block=dict(fill="H", set=[[0, "E"], [1, "l"], [2, "L"], [3, "o"]]),
         expect_positions={"0": "range(4,64)", "1": [1], "2": [3]}),
]


def main():
    out = {"_comment": "Second harvest of the reference's own tests for the search path (tools/harvest_reference_kats.py): "
                       "inputs and the reference's assertions as properties.  Data only.",
           "properties": [], "encoded_properties": [], "profile_masks": ASCII_MASKS}
    seen = set()
    for path, fn, spec in SPECS:
        spec = dict(spec)
        a, b, body = fn_body(path, fn)
        lits = literals(body)
        e = {"id": fn + ("_" + spec.pop("suffix") if "suffix" in spec else ""), "source": f"{path}:{a}-{b}"}
        assert e["id"] not in seen, e["id"]
        seen.add(e["id"])
        pat = spec.pop("pat", None)
        txt = spec.pop("txt", None)
        e["pattern"] = spec.pop("pattern") if pat is None else lits[pat]
        if "text_build" in spec:
            tb = spec.pop("text_build")
            if "overwrites" in tb:  # text.splice(pos..pos + m, pattern): an overwrite
                t = bytearray(tb["fill"].encode() * tb["len"])
                for at in tb["overwrites"]:
                    t[at:at + len(e["pattern"])] = e["pattern"].encode()
                e["text"] = t.decode()
            else:
                t = bytearray(tb["fill"].encode() * tb["len"])
                comp = bytes.maketrans(b"ACGT", b"TGCA")
                for at, what in tb["splices"]:
                    kind, idx = what.split(":")
                    s = lits[int(idx)].encode()
                    if kind == "revcomp":
                        s = s.translate(comp)[::-1]
                    t[at:at] = s
                e["text"] = t.decode()
        else:
            e["text"] = lits[txt]
        e.setdefault("rc", False)
        e["alpha"] = None
        e.update(spec)
        out["properties"].append(e)
    for path, fn, spec in ENC_SPECS:
        spec = dict(spec)
        a, b, body = fn_body(path, fn)
        lits = literals(body)
        e = {"id": fn + ("_" + spec.pop("suffix") if "suffix" in spec else ""), "source": f"{path}:{a}-{b}"}
        if "patterns_lit" in spec:
            e["patterns"] = [lits[i] for i in spec.pop("patterns_lit")]
        e["text"] = lits[spec.pop("txt")]
        e["rc"] = False
        e.update(spec)
        out["encoded_properties"].append(e)
    dst = os.path.join(ROOT, "tests", "golden", "kats_more.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(dst, len(out["properties"]), "+", len(out["encoded_properties"]), "+", len(out["profile_masks"]), "entries")
    for e in out["properties"]:
        print(f'  {e["id"]:48s} m={len(e["pattern"]):4d} n={len(e["text"]):5d} k={e["k"]:3d} {e["profile"]:5s} rc={int(e["rc"])} alpha={e["alpha"]} {e["prop"]}')


if __name__ == "__main__":
    main()
