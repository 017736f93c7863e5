"""Re-run the case tests/fuzz_gpu.py saved in gpurun_out/fuzz_fail.bin (or the file given) and compare with the oracle."""
import ast, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle, sassy_amd
raw = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_fail.bin", "rb").read()
d, rest = raw.split(b"\n", 1)
desc = ast.literal_eval(d.decode())
pat, text = rest.split(b"\n", 1)
print(desc, len(pat), len(text))
key = lambda ms: [(m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar) for m in ms]
want = key(oracle.search(desc["profile"], pat, text, desc["k"], all_minima=desc["all_minima"]))
s = sassy_amd.Searcher(desc["profile"], rc=False)
for rep in range(4):
    if rep == 2:  # other searches in between: what a stale block would hold
        s.search_all(b"ACGTACG", text[:20000], 0)
        s.search(text[1000:1053], text, 2)
    got = key(s.search_all(pat, text, desc["k"]) if desc["all_minima"] else s.search(pat, text, desc["k"]))
    st = s.stats()
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    print(rep, "equal" if got == want else "DIFFERENT", len(got), len(want), "first bad rows", bad[:5], "n bad", len(bad), "last bad", bad[-3:],
          {k: st[k] for k in ("filtered", "fused", "candidates", "chunks")})
    if bad:
        i = bad[0]
        print(" got", got[i], "\n want", want[i])
