"""One case of tools/bench_texts.py as lone searches only (for rocprofv3 --kernel-trace timelines):
    PROBE_CASE=<name> [PROBE_N=3e9] [PROBE_REPS=8] python tools/probe_text_case.py
Prints one JSON line (ms per lone search, matches, stats of the last search)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd
from bench import _dna_bytes
from tools.bench_texts import consensus_32mer

n = int(float(os.environ.get("PROBE_N", "3e9"))) // 64 * 64
reps = int(os.environ.get("PROBE_REPS", "8"))
rnd32 = bytes(_dna_bytes(43, 0, 32))
CASES = {
    "iid_plant_1MiB": (("dna", 1 << 20), "dna", rnd32, 3),
    "iid_plant_4KiB": (("dna", 4096), "dna", rnd32, 3),
    "repeats_random32": (("genome", False), "dna", rnd32, 3),
    "repeats_family32": (("genome", False), "dna", consensus_32mer(0, 1000), 3),
    "repeats_ACx16": (("genome", False), "dna", b"AC" * 16, 3),
    "repeats_polyA": (("genome", False), "dna", b"A" * 32, 3),
    "repeatsN_iupac_random32": (("genome", True), "iupac", rnd32, 3),
    "repeatsN_iupac_family32": (("genome", True), "iupac", consensus_32mer(1, 2000), 3),
    "repeatsN_dna_random32": (("genome", True), "dna", rnd32, 3),
    "repeats_iupac_random32": (("genome", False), "iupac", rnd32, 3),
}
name = os.environ.get("PROBE_CASE", "repeatsN_iupac_random32")
text, profile, pat, k = CASES[name]
buf = sassy_amd.DeviceBuffer(n + 4096)
if text[0] == "dna":
    sassy_amd.generate_dna(buf.ptr, n, 42, 0)
    if text[1]:
        sassy_amd.plant(buf.ptr, n, 0, n, 42, pat, k, stride=text[1])
else:
    sassy_amd.generate_genome_like(buf.ptr, n, 42, 0, with_n=text[1])
s = sassy_amd.Searcher(profile, rc=bool(int(os.environ.get("PROBE_RC", "0"))))
r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
t0 = time.perf_counter()
for _ in range(reps):
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
lone = (time.perf_counter() - t0) / reps
st = s.stats()
print(json.dumps({"case": name, "ms_lone": round(lone * 1e3, 3), "matches": len(r),
                  "stats": {x: st[x] for x in ("filtered", "fused", "piece_len", "chunks", "hit_blocks", "candidates", "host_wait_ms", "host_enqueue_ms", "host_post_ms")}}), flush=True)
