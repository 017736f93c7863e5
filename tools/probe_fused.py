"""Lone-search timing of BASELINE config 2 under the current environment switches (one JSON line).
SASSY_HIP_FUSED_PROBE=1: the fused filter also reports where its waves spend their time (100 MHz ticks)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd
from bench import _dna_bytes

n = int(float(os.environ.get("PROBE_N", "3e9"))) // 64 * 64
m, k = int(os.environ.get("PROBE_M", "32")), int(os.environ.get("PROBE_K", "3"))
pat = bytes(_dna_bytes(43, 0, m))
plain = pat
if os.environ.get("PROBE_C3"):  # BASELINE config 3: N, R, Y, W at 50 / 100 / 150 / 199 (planted with a base they contain)
    p3 = bytearray(pat); p3[50], p3[100], p3[150], p3[199] = b"NRYW"
    plain = bytes({78: 65, 82: 65, 89: 67, 87: 65}.get(c, c) for c in p3); pat = bytes(p3)
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
sassy_amd.plant(buf.ptr, n, 0, n, 42, plain, k, 1 << 20)
s = sassy_amd.Searcher(os.environ.get("PROBE_PROFILE", "dna"), rc=bool(int(os.environ.get("PROBE_RC", "0"))))
for _ in range(60):
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
f = 0.0
for _ in range(30):
    r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
    f += s.stats()["filter_ms"] / 30
st = s.stats()
s.set_timing(0)
for _ in range(5):
    s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
t0 = time.perf_counter()
for _ in range(50):
    s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
lat = (time.perf_counter() - t0) / 50 * 1e3
out = {"env": {k_: v for k_, v in os.environ.items() if k_.startswith("SASSY_HIP_")}, "lone_ms": round(lat, 4),
       "kernel_ms": round(f, 4), "frac_lone": round(n / lat / 1e6 / 8000, 4), "matches": len(r), "fused": st["fused"], "filtered": st["filtered"],
       "chunks": st["chunks"], "host_wait_ms": round(st["host_wait_ms"], 4), "host_enqueue_ms": round(st["host_enqueue_ms"], 4), "host_post_ms": round(st["host_post_ms"], 4)}
if os.environ.get("SASSY_HIP_FUSED_PROBE"):
    w = max(1, st["live_blocks"])
    out["probe"] = {"waves_with_chunks": st["live_blocks"], "stream_us_per_wave": round(st["word_rows"] / w / 100, 2),
                    "dp_us_per_wave": round(st["blocks"] / w / 100, 2), "chunks_per_wave": round(st["hit_blocks"] / w, 2)}
if os.environ.get("PROBE_COUNTERS"):  # rows the streaming DP computed per block (wave-voted cut-off)
    c0 = s.stats()
    s.enable_counters(True)
    s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
    c = s.stats()
    d = {x: c[x] - c0[x] for x in ("word_rows", "blocks", "live_blocks")}
    out["rows_per_block"] = round(d["word_rows"] / max(1, d["blocks"]), 2)
    out["live_frac"] = round(d["live_blocks"] / max(1, d["blocks"]), 5)
print(json.dumps(out), flush=True)
