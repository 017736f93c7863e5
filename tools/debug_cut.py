"""Debug helper: replay the fuzz case that failed with the row cut-off and print got vs want."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, sassy_amd
def rand_seq(rng, n, alphabet=b"ACGT"): return bytes(rng.choice(alphabet) for _ in range(n))
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
exec("def mutate" + src.split("def mutate")[1].split("\n\n\n")[0])
profile = "dna"
rng = random.Random(42)
fwd = sassy_amd.Searcher(profile, rc=False); both = sassy_amd.Searcher(profile, rc=True)
bad = 0
for it in range(400):
    m = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 23, 31, 32, 33, 40, 63, 64, 65, 100, 130])
    k = min(rng.choice([0, 0, 1, 2, 3, 3, 5, 8]), m - 1) if m > 1 else 0
    n = rng.choice([0, 1, 2, 31, 63, 64, 65, 100, 127, 128, 129, 200, 500, 511, 512, 513, 1500, 4000])
    pal = b"ACGT"
    pat = rand_seq(rng, m, pal if rng.random() < 0.3 else b"ACGT")
    text = bytearray(rand_seq(rng, n, b"ACGT" if rng.random() < 0.7 else b"ACGTacgt"))
    for _ in range(rng.randrange(5)):
        if n > m + 8:
            at = rng.randrange(0, n - m - 6)
            ins = mutate(rng, bytes(c if c in b"ACGT" else 65 for c in pat), rng.randrange(k + 2))
            text[at:at + len(ins)] = ins
    text = bytes(text[:n])
    rc = rng.random() < 0.4
    allm = rng.random() < 0.25
    for strand_rc in ((False, True) if rc else (False,)):
        s = both if strand_rc else fwd
        got = s.search_all(pat, text, k) if allm else s.search(pat, text, k)
        want = oracle.search(profile, pat, text, k, rc=strand_rc, all_minima=allm)
        g = [(x.text_start, x.text_end, x.cost, x.strand) for x in got]
        w = [(x.text_start, x.text_end, x.cost, x.strand) for x in want]
        if g != w:
            bad += 1
            print("MISMATCH it", it, "m", m, "k", k, "n", n, "rc", strand_rc, "all", allm, "stats", s.stats()["filtered"])
            print("  missing", sorted(set(w) - set(g))[:6], "extra", sorted(set(g) - set(w))[:6])
print("bad", bad)
