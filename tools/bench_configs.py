"""Timings of the BASELINE.json configs that are not the bench.py line (configs 1, 3, 4), on one GPU.

bench.py measures config 2 (the configuration the metric is quoted on) under the driver's contract;
this script reports the other single-GPU configurations with the same synthetic text so
that DESIGN.md can state where they stand.  One JSON line per config.

    python tools/bench_configs.py [--configs 1,3,4,5] [--text-bytes N] [--patterns P] [--steps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import sassy_amd  # noqa: E402


def dna_bytes(seed: int, n: int) -> bytes:
    rng = np.random.default_rng(seed)
    return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)])


def timed(fn, steps):
    fn()  # warm-up (allocations, first launch)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    return (time.perf_counter() - t0) / steps, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,3,4")
    ap.add_argument("--text-bytes", type=int, default=3_000_000_000)
    ap.add_argument("--patterns", type=int, default=10_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    todo = [int(c) for c in args.configs.split(",")]
    n = args.text_bytes
    buf = sassy_amd.DeviceBuffer(n + 64)
    sassy_amd.generate_dna(buf.ptr, n, 42, 0)

    if 1 in todo:
        # config 1: 'ATCG' x 8, k=3, 1 MiB random ACGT (the reference's own CPU-runnable case)
        pat = b"ATCG" * 8
        n1 = 1 << 20
        s = sassy_amd.Searcher("dna", rc=False)
        dt, r = timed(lambda: s.search_shard(pat, buf.ptr, 0, n1, 0, n1, 3), 50)
        print(json.dumps({"config": 1, "workload": "Dna new_fwd, 'ATCG'x8, k=3, 1 MiB random ACGT (device resident)",
                          "ms_per_search": round(dt * 1e3, 4), "GB_per_s": round(n1 / dt / 1e9, 2),
                          "matches": len(r), "stats": {k: s.stats()[k] for k in ("scan_ms", "trace_ms", "filtered")}}))

    if 3 in todo:
        # config 3: |pattern| = 200 with 4 IUPAC letters, k = 20, Iupac profile
        p = bytearray(dna_bytes(44, 200))
        p[50], p[100], p[150], p[199] = ord("N"), ord("R"), ord("Y"), ord("W")
        pat = bytes(p)
        # the planted copies spell the ambiguity letters with a base they contain
        plain = bytes({ord("N"): 65, ord("R"): 65, ord("W"): 65, ord("Y"): 67}.get(c, c) for c in pat)
        planted = sassy_amd.plant(buf.ptr, n, 0, n, 42, plain, 20)
        s = sassy_amd.Searcher("iupac", rc=False)
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, 20)
        st = s.stats()  # (kernel times by HIP events: this first call only)
        s.set_timing(0)
        for _ in range(25):  # (lone searches settle over their first ~20 calls)
            s.search_shard(pat, buf.ptr, 0, n, 0, n, 20)
        dt, r = timed(lambda: s.search_shard(pat, buf.ptr, 0, n, 0, n, 20), max(args.steps, 20))
        print(json.dumps({"config": 3, "workload": f"Iupac new_fwd, |pattern|=200 (N,R,Y,W at 50/100/150/199), k=20, {n} B random ACGT + plants",
                          "ms_per_search": round(dt * 1e3, 3), "GB_per_s": round(n / dt / 1e9, 2),
                          "matches": len(r), "planted": planted,
                          "stats": {k: st[k] for k in ("scan_ms", "filter_ms", "trace_ms", "filtered", "piece_len", "hit_blocks", "chunks")}}))
        sassy_amd.generate_dna(buf.ptr, n, 42, 0)  # undo the plants

    if 4 in todo:
        # config 4: search_encoded_patterns, P random 20-mers, k = 2, Iupac searcher, fwd only
        P = args.patterns
        pats = [dna_bytes(45 + i, 20) for i in range(P)]
        s = sassy_amd.Searcher("iupac", rc=False)
        enc = s.encode_patterns(pats)
        secs = []
        for rep in range(args.steps if args.steps < 3 else 3):  # the first call also sizes the device buffers
            t0 = time.perf_counter()
            out = sassy_amd.C.c_void_p()
            sassy_amd._check(sassy_amd.lib().sassy_hip_search_encoded(s._h, enc._h, buf.ptr, n, 2, sassy_amd.TEXT_ON_DEVICE,
                                                                      sassy_amd.C.byref(out)))
            secs.append(time.perf_counter() - t0)
            r = sassy_amd.Result(out)
        dt = min(secs)
        st = s.stats()
        print(json.dumps({"config": 4, "workload": f"search_encoded_patterns, {P} random 20-mers, k=2, Iupac new_fwd, {n} B random ACGT",
                          "seconds": round(dt, 3), "seconds_each_call": [round(x, 3) for x in secs],
                          "text_GB_per_s": round(n / dt / 1e9, 3),
                          "pattern_text_GB_per_s": round(n * P / dt / 1e9, 1), "matches": len(r),
                          "stats": {k: st[k] for k in ("scan_ms", "filter_ms", "trace_ms", "scan_launches", "filtered", "chunks",
                                                       "hit_blocks", "live_blocks", "candidates")}}))

    if 5 in todo:
        # config 5's DATA PATH on ONE GPU (no scaling claim): 24e9 bytes in eight shards with halos through the
        # in-process multi-device searcher, device 0 named eight times -- eight host threads, eight resident shards,
        # eight searches on the same GPU, merged in C.  What eight GPUs would each do once, this GPU does eight times.
        buf.free()
        n5 = 24_000_000_000
        ms = sassy_amd.MultiSearcher("dna", devices=[0] * 8)
        ms.generate_dna(n5, 42, 32, 3)
        from bench import _dna_bytes
        pat = bytes(_dna_bytes(43, 0, 32))
        planted = ms.plant(42, pat, 3, 1 << 20)
        dt, r = timed(lambda: ms.search(pat, 3), max(args.steps, 10))
        print(json.dumps({"config": 5, "workload": f"Dna new_fwd, |pattern|=32, k=3, {n5} B in 8 shards ON ONE GPU (sassy_hip_multi_*, device 0 x 8)",
                          "ms_per_search": round(dt * 1e3, 3), "text_GB_per_s_one_gpu": round(n5 / dt / 1e9, 1),
                          "matches": len(r), "planted": int(planted),
                          "note": "eight shards share one GPU: the time is eight shard searches, not a scaling measurement"}))


if __name__ == "__main__":
    main()
