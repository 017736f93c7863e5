"""Per-step wall times of a pipelined stream of searches right after start-up (why are the first steps slower?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd
from bench import _dna_bytes
n = 3_000_000_000
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
pat = bytes(_dna_bytes(43, 0, 32))
sassy_amd.plant(buf.ptr, n, 0, n, 42, pat, 3, 1 << 20)
s = sassy_amd.Searcher("dna", rc=False)
s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
pend = []
ts = []
t0 = time.perf_counter()
for i in range(120):
    pend.append(s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, 3))
    if len(pend) == 2:
        s.search_finish(pend.pop(0))
    ts.append(time.perf_counter())
while pend:
    s.search_finish(pend.pop(0))
d = [round((b - a) * 1e3, 3) for a, b in zip([t0] + ts[:-1], ts)]
print("step ms:", d[:60])
print("avg 60..120:", sum(d[60:]) / 60)
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))
    ts = []; t0 = time.perf_counter()
    for i in range(40):
        pend.append(s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, 3))
        if len(pend) == 2:
            s.search_finish(pend.pop(0))
        ts.append(time.perf_counter())
    while pend:
        s.search_finish(pend.pop(0))
    d = [round((b - a) * 1e3, 3) for a, b in zip([t0] + ts[:-1], ts)]
    print("after sleep:", d)
