"""The search path on texts that are less kind to a prefilter than i.i.d. letters (SURVEY 8d's dense-plant
variant, the periodic BASELINE pattern 'ATCG'x8, a repeat-rich synthetic text, N runs).  Per case: which
filter ran, how many blocks it left for the DP (hit_blocks, chunks), the time of a lone search and of a stream of
searches (two in flight), matches / s, and a parity check of the matches inside a few 1 MiB slices against the
oracle.  One JSON line per case.

    python tools/bench_texts.py [--text-bytes N] [--steps K] [--cases a,b,..]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import oracle  # noqa: E402  (this tool is a checker / measurement script, not the product)
import sassy_amd  # noqa: E402
from bench import _dna_bytes  # noqa: E402

SL = 1 << 20


def consensus_32mer(fam: int, at: int) -> bytes:
    """32 letters of repeat family `fam`'s consensus (aux_kernels.hip: genome_like_byte), from position `at`."""
    def splitmix(x):
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return x ^ (x >> 31)
    out = bytearray()
    for cp in range(at, at + 32):
        hc = splitmix(((0xfa000000 + fam) * 0x9E3779B97F4A7C15 + (cp >> 5)) & (2**64 - 1))
        out.append(b"ACGT"[(hc >> (2 * (cp & 31))) & 3])
    return bytes(out)


def slices_parity(buf, n, profile, pat, k, arr, pool, slices):
    """matches of the whole-text search that lie inside [a + 256, a + SL - 256) against the oracle on the slice"""
    ok, compared = True, 0
    ts, te = arr["text_start"].astype(np.int64), arr["text_end"].astype(np.int64)
    for a in slices:
        a = min(max(0, a // 64 * 64), n - SL)
        sl = buf.download(SL, a)
        try:
            want = [(m.text_start + a, m.text_end + a, m.cost, m.cigar) for m in oracle.search(profile, pat, sl, k)
                    if m.text_start >= 256 and m.text_end <= SL - 256]
        except RuntimeError:
            continue  # the reference would panic in this slice (Dna traceback over a non-ACGT letter)
        sel = np.nonzero((ts >= a + 256) & (te <= a + SL - 256))[0]
        got = [(int(ts[i]), int(te[i]), int(arr["cost"][i]),
                pool[int(arr["cigar_off"][i]):int(arr["cigar_off"][i]) + int(arr["cigar_len"][i])].decode()) for i in sel]
        compared += len(want)
        if got != want:
            ok = False
    return ok, compared


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-bytes", type=int, default=3_000_000_000)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--cases", default="")
    args = ap.parse_args()
    n = args.text_bytes // 64 * 64
    buf = sassy_amd.DeviceBuffer(n + 4096)
    rnd32 = bytes(_dna_bytes(43, 0, 32))
    cases = [
        # name, text maker, profile, pattern, k
        ("iid_plant_1MiB", ("dna", 1 << 20), "dna", rnd32, 3),
        ("iid_plant_4KiB", ("dna", 4096), "dna", rnd32, 3),
        ("iid_ATCGx8", ("dna", 0), "dna", b"ATCG" * 8, 3),
        ("repeats_random32", ("genome", False), "dna", rnd32, 3),
        ("repeats_family32", ("genome", False), "dna", consensus_32mer(0, 1000), 3),
        ("repeats_ACx16", ("genome", False), "dna", b"AC" * 16, 3),
        ("repeats_polyA", ("genome", False), "dna", b"A" * 32, 3),
        ("repeatsN_iupac_random32", ("genome", True), "iupac", rnd32, 3),
        ("repeatsN_iupac_family32", ("genome", True), "iupac", consensus_32mer(1, 2000), 3),
    ]
    only = set(c for c in args.cases.split(",") if c)
    made = None
    for name, text, profile, pat, k in cases:
        if only and name not in only:
            continue
        if text != made:
            if text[0] == "dna":
                sassy_amd.generate_dna(buf.ptr, n, 42, 0)
                if text[1]:
                    sassy_amd.plant(buf.ptr, n, 0, n, 42, pat, k, stride=text[1])
            else:
                sassy_amd.generate_genome_like(buf.ptr, n, 42, 0, with_n=text[1])
            made = text
        s = sassy_amd.Searcher(profile, rc=False)
        t0 = time.perf_counter()
        r = s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        cold = time.perf_counter() - t0
        st = s.stats()
        nm = len(r)
        steps = max(3, min(args.steps, int(2.0 / max(cold, 1e-4))))
        for _ in range(3):  # (buffers and pinned blocks of this result size exist from here on)
            s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
        each = []
        for _ in range(steps):
            t0 = time.perf_counter()
            s.search_shard(pat, buf.ptr, 0, n, 0, n, k)
            each.append(time.perf_counter() - t0)
        lone = sorted(each)[len(each) // 2]  # (the median: one call in a few dozen pins a fresh host block)
        lone_mean = sum(each) / len(each)
        pend = []
        t0 = time.perf_counter()
        for _ in range(steps):
            pend.append(s.search_shard_begin(pat, buf.ptr, 0, n, 0, n, k))
            if len(pend) == 2:
                s.search_finish(pend.pop(0))
        while pend:
            s.search_finish(pend.pop(0))
        stream = (time.perf_counter() - t0) / steps
        arr, pool = r.array, r.pool
        sl = [0, n // 3, n // 2 + 12345, n - SL]
        if nm:
            sl.append(int(arr["text_start"][nm // 2]) - SL // 2)
        ok, compared = slices_parity(buf, n, profile, pat, k, arr, pool, sl)
        print(json.dumps({
            "case": name, "profile": profile, "pattern": pat.decode(), "k": k, "text_bytes": n,
            "filter_kind": st["filtered"], "piece_or_q": st["piece_len"], "hit_blocks": st["hit_blocks"], "chunks": st["chunks"],
            "hit_block_fraction": round(st["hit_blocks"] / (n / 64), 5), "matches": nm,
            "ms_lone_search": round(lone * 1e3, 3), "ms_lone_search_mean": round(lone_mean * 1e3, 3), "ms_lone_search_max": round(max(each) * 1e3, 3),
            "ms_per_search_2_in_flight": round(stream * 1e3, 3),
            "TB_per_s_lone": round(n / lone / 1e12, 3), "TB_per_s_stream": round(n / stream / 1e12, 3),
            "matches_per_s_stream": round(nm / stream, 1), "filter_ms": round(st["filter_ms"], 3),
            "tail_ms_lone": round(lone * 1e3 - st["filter_ms"], 3) if st["filtered"] else None,
            "slices_equal_oracle": ok, "matches_compared": compared}), flush=True)
    buf.free()


if __name__ == "__main__":
    main()
