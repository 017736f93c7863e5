"""Do two searches on two streams overlap on the device?  T host threads, each with its own Searcher (own
stream), loop over search_shard on the same resident text; prints searches per second for T = 1, 2, 3."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sassy_amd
sys.path.insert(0, ROOT)
from bench import _dna_bytes

n = int(os.environ.get("N", 3_000_000_000)) // 64 * 64
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
pats = [bytes(_dna_bytes(43 + i, 0, 32)) for i in range(4)]
sassy_amd.plant(buf.ptr, n, 0, n, 42, pats[0], 3, 1 << 20)
searchers = [sassy_amd.Searcher("dna", rc=False) for _ in range(4)]
for s in searchers:
    s.set_timing(0)
    for _ in range(45):
        s.search_shard(pats[0], buf.ptr, 0, n, 0, n, 3)

def loop(s, pat, iters, out, i):
    t0 = time.perf_counter()
    for _ in range(iters):
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
    out[i] = (time.perf_counter() - t0, len(r))

for T in (1, 2, 3, 4):
    iters = 200
    out = [None] * T
    th = [threading.Thread(target=loop, args=(searchers[i], pats[0], iters, out, i)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"threads {T}: {T * iters / dt:.1f} searches/s = {dt / (T * iters) * 1e3:.4f} ms per search, {n * T * iters / dt / 1e12:.3f} TB/s; per-thread {[round(o[0] / iters * 1e3, 4) for o in out]} matches {out[0][1]}")
