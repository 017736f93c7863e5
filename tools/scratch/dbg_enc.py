import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import oracle, sassy_amd as sassy
import numpy as np
rng = random.Random(131)
def rand_seq(rng, n, alphabet=b"ACGT"): return bytes(rng.choice(alphabet) for _ in range(n))
def mutate(rng, s, edits):
    s = bytearray(s)
    for _ in range(edits):
        t, p = rng.randrange(3), rng.randrange(len(s))
        if t == 0: s[p] = rng.choice(b"ACGT")
        elif t == 1: s.insert(p, rng.choice(b"ACGT"))
        elif len(s) > 1: del s[p]
    return bytes(s)
profile, m, k, npat, n = "dna", 20, 2, 50, 120_000
pats = [rand_seq(rng, m) for _ in range(npat)]
text = bytearray(rand_seq(rng, n))
for _ in range(n // 30):
    p_ = rng.choice(pats)
    ins = mutate(rng, p_, rng.randrange(0, k + 1))
    if rng.random() < 0.5: ins = oracle.reverse_complement("iupac", ins)
    at = rng.randrange(0, n - len(ins)); text[at:at + len(ins)] = ins
tb = bytes(text)
res = []
for pin in ("1", "0"):
    os.environ["SASSY_HIP_SEEDED"] = "1"; os.environ["SASSY_HIP_ENCODED_PIN"] = pin
    s = sassy.Searcher(profile, rc=True)
    r = s.search_encoded_patterns(s.encode_patterns(pats), tb, k, as_result=True)
    res.append(r)
a, b = res[0].array, res[1].array
print(len(a), len(b))
for f in a.dtype.names:
    d = np.nonzero(a[f] != b[f])[0] if a[f].ndim == 1 else np.nonzero((a[f] != b[f]).any(axis=1))[0]
    print(f, len(d), d[:5], a[f][d[:3]], b[f][d[:3]])
