import os, sys, time, json
sys.path.insert(0, ".")
import sassy_amd
from tools.bench_texts import consensus_32mer
n = 3_000_000_000
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_genome_like(buf.ptr, n, 42, 0, with_n=True)
pat = consensus_32mer(1, 2000)
s = sassy_amd.Searcher("iupac", rc=False)
ts = []
for i in range(24):
    t0 = time.perf_counter()
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
    t1 = time.perf_counter()
    ts.append(round((t1 - t0) * 1e3, 3))
    if i % 2 == 0: del r
print(ts, len(s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)))
ts = []
for i in range(12):
    t0 = time.perf_counter()
    s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
    ts.append(round((time.perf_counter() - t0) * 1e3, 3))
print("discarded", ts)
