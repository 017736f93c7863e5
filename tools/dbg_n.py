import sys, os
sys.path.insert(0, ".")
import numpy as np, oracle, sassy_amd
from bench import _dna_bytes
n = 64 << 20
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_genome_like(buf.ptr, n, 42, 0, with_n=True)
pat = bytes(_dna_bytes(43, 0, 32))
s = sassy_amd.Searcher("iupac", rc=False)
r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
print("fused", s.stats()["fused"], "matches", len(r))
s2 = sassy_amd.Searcher("iupac", rc=False); s2.set_fused(False)
r2 = s2.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
a = set(zip(r.array["text_end"].tolist(), r.array["cost"].tolist()))
b = set(zip(r2.array["text_end"].tolist(), r2.array["cost"].tolist()))
miss = sorted(b - a); extra = sorted(a - b)
print("classic", len(r2), "missing", len(miss), "extra", len(extra))
st = s.stats()
bpl = st["blocks_per_chunk"]
for e, c in miss[:25]:
    print("missing end", e, "cost", c, "off in 4K region", e % 4096, "block", e // 64, "block in lane", (e // 64) % bpl, "bpl", bpl)
