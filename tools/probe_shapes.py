"""Lone-search time of a few (profile, m, k) shapes on a 3 GB resident text under the current environment switches."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd
from bench import _dna_bytes

n = int(float(os.environ.get("PROBE_N", "3e9"))) // 64 * 64
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
out = {"env": {k_: v for k_, v in os.environ.items() if k_.startswith("SASSY_HIP_")}}
for profile, m, k in (("dna", 32, 3), ("iupac", 200, 20), ("iupac", 32, 3), ("dna", 64, 12), ("dna", 100, 10)):
    pat = bytes(_dna_bytes(43, 0, m))
    s = sassy_amd.Searcher(profile, rc=False)
    for _ in range(8):
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
    f = 0.0
    for _ in range(10):
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        st = s.stats()
        f += (st["filter_ms"] if st["filtered"] else st["scan_ms"]) / 10
    t0 = time.perf_counter()
    for _ in range(15):
        s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
    lat = (time.perf_counter() - t0) / 15 * 1e3
    out[f"{profile} m={m} k={k}"] = {"lone_ms": round(lat, 4), "kernel_ms": round(f, 4), "filtered": st["filtered"], "matches": len(r)}
print(json.dumps(out), flush=True)
