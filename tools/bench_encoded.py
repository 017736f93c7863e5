"""search_encoded_patterns: the pattern-tiled one-pass scan (tiled_kernel.hip) against one scan per pattern,
over (number of patterns) x (text length).  Device-resident random-ACGT text, random 20-mers, k = 2, Iupac
searcher as in BASELINE config 4.  Prints one line per shape: ms per call for both paths, and the
pattern-tiled kernel's rate in (text characters x patterns) per second.

    python tools/bench_encoded.py [--m 20] [--k 2] [--profile iupac]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sassy_amd  # noqa: E402


class DevText:
    def __init__(self, ptr, n):
        self._p, self._n, self.is_cuda = ptr, n, True

        class _DT:
            itemsize = 1
        self.dtype = _DT()

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n

    def is_contiguous(self):
        return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=20)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--profile", default="iupac")
    ap.add_argument("--max-per-pattern-s", type=float, default=20.0)
    args = ap.parse_args()
    rng = random.Random(5)
    nmax = 256 << 20
    buf = sassy_amd.DeviceBuffer(nmax + 4096)
    sassy_amd.generate_dna(buf.ptr, nmax, 7, 0)
    print(f"# m={args.m} k={args.k} profile={args.profile}")
    print("# npat  text_bytes  tiled_ms  per_pattern_ms  speedup  tiled_kernel_ms  cells/s(kernel)  matches")
    for npat in (64, 1000, 10000):
        pats = [bytes(rng.choice(b"ACGT") for _ in range(args.m)) for _ in range(npat)]
        for n in (10_000, 1 << 20, 16 << 20, 64 << 20, 256 << 20):
            res = {}
            for tiled in ("1", "0"):
                os.environ["SASSY_HIP_TILED"] = tiled
                s = sassy_amd.Searcher(args.profile, rc=False)
                enc = s.encode_patterns(pats)
                est = npat * 60e-6 + npat * n / 2e12 if tiled == "0" else 0
                if est > args.max_per_pattern_s:
                    res[tiled] = (float("nan"), 0, 0.0)
                    continue
                best = 1e30
                for rep in range(3):
                    t0 = time.perf_counter()
                    r = s.search_encoded_patterns(enc, DevText(buf.ptr, n), args.k, as_result=True)
                    best = min(best, time.perf_counter() - t0)
                st = s.stats()
                res[tiled] = (best * 1e3, len(r), st["scan_ms"] if tiled == "1" else 0.0)
            t_ms, nm, kern_ms = res["1"]
            p_ms, nm0, _ = res["0"]
            assert nm0 in (0, nm) or p_ms != p_ms, (nm, nm0)
            rate = (n * npat / (kern_ms * 1e-3)) if kern_ms else 0.0
            print(f"{npat:6d} {n:11d} {t_ms:9.2f} {p_ms:14.2f} {p_ms / t_ms:8.1f} {kern_ms:10.3f} {rate:12.3e} {nm:8d}",
                  flush=True)
    os.environ.pop("SASSY_HIP_TILED", None)


if __name__ == "__main__":
    main()
