"""Config 2 lone / in-flight search times for the switches given in the environment (one line)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sassy_amd
from bench import _dna_bytes
n = 3_000_000_000
buf = sassy_amd.DeviceBuffer(n + 4096)
pat = bytes(_dna_bytes(43, 0, 32))
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
sassy_amd.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 20)
s = sassy_amd.Searcher("dna", rc=False)
s.set_pipe_depth(3)
for _ in range(30):
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
ts = []
for _ in range(100):
    t0 = time.perf_counter()
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
    ts.append(time.perf_counter() - t0)
ts.sort()
pend = []
t0 = time.perf_counter()
for _ in range(200):
    pend.append(s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, 3))
    if len(pend) == 3:
        s.search_finish(pend.pop(0))
while pend:
    s.search_finish(pend.pop(0))
stream = (time.perf_counter() - t0) / 200
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SASSY_HIP_")}, "lone_median_ms": round(ts[50] * 1e3, 4),
                  "lone_min_ms": round(ts[0] * 1e3, 4), "stream3_ms": round(stream * 1e3, 4), "matches": len(r),
                  "chunks": s.stats()["chunks"]}))
