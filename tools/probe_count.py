import sys, time, json
sys.path.insert(0, '.')
import numpy as np, sassy_amd
n = 3_000_000_000
buf = sassy_amd.DeviceBuffer(n + 64)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
rng = np.random.default_rng(1)
def pat(m, seed): 
    r = np.random.default_rng(seed); return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0,4,m)])
def run(profile, p, k, steps=20, warm=25):
    # the phases' kernel times from one search with events at every phase; "ms" = lone searches without events, after
    # `warm` untimed ones (a lone search settles over its first ~20 calls: clocks)
    s = sassy_amd.Searcher(profile, rc=False)
    s.set_timing(2)
    s.search_shard(p, buf.ptr, 0, n, 0, n, k)
    s.search_shard(p, buf.ptr, 0, n, 0, n, k)
    st = s.stats()
    s.set_timing(0)
    for _ in range(warm): s.search_shard(p, buf.ptr, 0, n, 0, n, k)
    t0 = time.perf_counter()
    for _ in range(steps): r = s.search_shard(p, buf.ptr, 0, n, 0, n, k)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"profile": profile, "m": len(p), "k": k, "ms": round(dt*1e3, 3), "matches": len(r),
        "path": {0: "streaming DP (scan_kernel)", 1: "slot-mask filter + chain", 2: ("paired bit-plane filter" if st["pair"] else "bit-plane filter") + (", fused launch" if st["fused"] else " + chain"),
                 3: "q-gram table filter + chain", 4: "q-gram counting filter + chain"}[int(st["filtered"])],
        "roofline_frac_lone": round(n / dt / 8e12, 4),
        **{q: (round(st[q], 3) if isinstance(st[q], float) else st[q]) for q in ("filtered", "fused", "pair", "piece_len", "filter_ms", "scan_ms", "trace_ms", "hit_blocks", "chunks")}}), flush=True)
run("dna", pat(32, 43), 3)
run("iupac", pat(32, 43), 3)
p = bytearray(pat(200, 44)); p[50], p[100], p[150], p[199] = ord("N"), ord("R"), ord("Y"), ord("W")
run("iupac", bytes(p), 20)
run("iupac", pat(20, 45), 2)
run("dna", pat(100, 46), 10)
run("iupac", pat(64, 47), 6)
run("iupac", pat(1000, 48), 100, steps=10, warm=10)
# the reference's own canonical shapes (benches/perf.rs:46-48: CRISPR guide 20 + NGG, k = 3) and the k / m ratios
# between 1/8 and 1/5, where the choice of path flips
run("iupac", pat(20, 49) + b"NGG", 3)
run("dna", pat(23, 49), 3)
run("dna", pat(32, 43), 4)
run("dna", pat(32, 43), 5)
run("dna", pat(32, 43), 6)
run("dna", pat(64, 47), 8)
run("dna", pat(64, 47), 12)
run("dna", pat(50, 50), 10)
run("iupac", pat(32, 43), 5)
# at most four pieces of six rows: the fused bit-plane launch (round 4)
run("dna", pat(20, 45), 2)
run("dna", pat(24, 51), 3)
run("iupac", pat(24, 51), 3)
run("dna", pat(27, 52), 3)
