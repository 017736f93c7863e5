"""Per-search wall time of the first searches of a new searcher (transients: allocations, fallbacks)."""
import sys, time, json, os
sys.path.insert(0, '.')
import sassy_amd
from bench import _dna_bytes
n = 3_000_000_000
buf = sassy_amd.DeviceBuffer(n + 64)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
pat = bytes(_dna_bytes(43, 0, 32))
for profile in ("dna", "iupac", "dna", "iupac"):
    s = sassy_amd.Searcher(profile, rc=False)
    if os.environ.get("PROBE_TIMING"): s.set_timing(int(os.environ["PROBE_TIMING"]))
    ts = []
    det = []
    for i in range(int(os.environ.get('PROBE_STEPS', '12'))):
        t0 = time.perf_counter()
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 3)
        ts.append(round((time.perf_counter() - t0) * 1e3, 3))
        q = s.stats()
        det.append((round(q["host_enqueue_ms"], 3), round(q["host_wait_ms"], 3), round(q["host_post_ms"], 3)))
    st = s.stats()
    print(json.dumps({"profile": profile, "ms": ts, "fused": st["fused"], "filtered": st["filtered"], "filter_ms": round(st["filter_ms"], 4), "enq_wait_post": det[1:]}), flush=True)
