"""Lone-search time of shapes whose pigeonhole pieces are shorter than 7 rows, with and without the bit-plane filter
(SASSY_HIP_PREFILTER unset / 1), 3 GB resident random text; one JSON line per (shape, mode)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd
from bench import _dna_bytes

n = int(float(os.environ.get("PROBE_N", "3e9"))) // 64 * 64
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
shapes = [("dna", 24, 3), ("dna", 27, 3), ("dna", 20, 2), ("dna", 18, 2), ("dna", 12, 1), ("iupac", 24, 3), ("iupac", 20, 2), ("dna", 32, 4), ("dna", 23, 3),
          ("dna", 11, 1), ("iupac", 10, 1), ("dna", 15, 2), ("dna", 17, 2)]
if os.environ.get("PROBE_SHAPES"):
    shapes = [(a, int(b), int(c)) for a, b, c in (x.split(":") for x in os.environ["PROBE_SHAPES"].split(","))]
for profile, m, k in shapes:
    pat = bytes(_dna_bytes(43, 0, m))
    for mode in ((-1,) if os.environ.get("PROBE_DEFAULT_ONLY") else (-1, 0)):
        s = sassy_amd.Searcher(profile, rc=False).set_prefilter(mode)
        for _ in range(3):
            r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        st = s.stats()
        s.set_timing(0)
        for _ in range(20):  # (a lone search settles over its first calls: clocks)
            s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        t0 = time.perf_counter()
        for _ in range(12):
            s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        lat = (time.perf_counter() - t0) / 12 * 1e3
        print(json.dumps({"shape": f"{profile} m={m} k={k}", "prefilter": mode, "lone_ms": round(lat, 4), "filtered": st["filtered"],
                          "fused": st["fused"], "piece_len": st["piece_len"], "chunks": st["chunks"], "matches": len(r)}), flush=True)
