"""The reference's CRISPR off-target benchmark shape (312 guides of 23 bp -- 20 bases + the NGG
PAM -- against a 3.1 Gbp genome, k = 3, 16 threads: v1 36.2 s, v2 15.7 s (AVX2) / 9.18 s (AVX-512)) on a
device-resident 3 GB random-ACGT text: search_encoded_patterns with an Iupac searcher, forward strand and both.

    python tools/bench_crispr.py [--guides 312] [--text-bytes 3000000000] [--k 3]
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd  # noqa: E402
from tools.bench_encoded import DevText  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--guides", type=int, default=312)
    ap.add_argument("--text-bytes", type=int, default=3_000_000_000)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--genome-like", action="store_true",
                    help="repeat-rich text with runs of N and scattered IUPAC letters (generate_genome_like) instead of "
                         "i.i.d. ACGT: the seeded search runs with the pattern-tiled scan around the other letters")
    ap.add_argument("--overhang", type=float, default=None,
                    help="an overhang searcher (alpha): the one pass over the text as a batch of one, beside the chain per pattern "
                         "(switch overhang_seeded = 0) on the first --chain-guides guides")
    ap.add_argument("--chain-guides", type=int, default=16)
    args = ap.parse_args()
    rng = random.Random(11)
    n = args.text_bytes
    buf = sassy_amd.DeviceBuffer(n + 4096)
    if args.genome_like:
        sassy_amd.generate_genome_like(buf.ptr, n, 42, 0, with_n=True)
    else:
        sassy_amd.generate_dna(buf.ptr, n, 42, 0)
    pats = [bytes(rng.choice(b"ACGT") for _ in range(20)) + b"NGG" for _ in range(args.guides)]
    if args.overhang is not None:
        for rc in (False, True):
            for one_pass in (True, False):
                sub = pats if one_pass else pats[:args.chain_guides]
                s = sassy_amd.Searcher("iupac", rc=rc, alpha=args.overhang)
                if not one_pass:
                    s.set_option("overhang_seeded", 0)
                enc = s.encode_patterns(sub)
                secs = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    r = s.search_encoded_patterns(enc, DevText(buf.ptr, n), args.k, as_result=True)
                    secs.append(time.perf_counter() - t0)
                print(json.dumps({"workload": f"{len(sub)} guides (20 bases + NGG), k={args.k}, Iupac searcher with overhang {args.overhang}, "
                                              f"{'both strands' if rc else 'forward strand'}, {n} B random ACGT resident in HBM",
                                  "path": "one pass (seeded search + edge segments)" if one_pass else "a kernel chain per pattern",
                                  "seconds": round(min(secs), 4), "seconds_per_guide": round(min(secs) / len(sub), 6), "matches": len(r),
                                  "filtered": s.stats()["filtered"]}), flush=True)
        return
    for rc in (False, True):
        s = sassy_amd.Searcher("iupac", rc=rc)
        enc = s.encode_patterns(pats)
        secs = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = s.search_encoded_patterns(enc, DevText(buf.ptr, n), args.k, as_result=True)
            secs.append(time.perf_counter() - t0)
        st = s.stats()
        npat = len(pats) * (2 if rc else 1)
        print(json.dumps({"workload": f"{args.guides} guides (20 bases + NGG), k={args.k}, Iupac searcher, "
                                      f"{'both strands' if rc else 'forward strand'}, {n} B "
                                      f"{'genome-like text with N runs' if args.genome_like else 'random ACGT'} resident in HBM",
                          "seconds": round(min(secs), 4), "seconds_each_call": [round(x, 4) for x in secs],
                          "pattern_text_GB_per_s": round(n * npat / min(secs) / 1e9, 1), "matches": len(r),
                          "path": st["filtered"], "table_hits": st["hit_blocks"], "verified": st["live_blocks"],
                          "zones": st["cond_resolved"],
                          "kernel_ms": round(st["scan_ms"], 2)}), flush=True)


if __name__ == "__main__":
    main()
