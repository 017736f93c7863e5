"""search_encoded_patterns: seed -> verify -> report (seed_kernels.hip), the pattern-tiled one-pass scan
(tiled_kernel.hip) and one kernel chain per pattern, over (number of patterns) x (text length).  Device-resident random-ACGT text, random 20-mers, k = 2, Iupac
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
    print("# ms per call (best of 3, device-resident text); kind = what the library picks by itself "
          "(6 seeded, 5 pattern-tiled, else one chain per pattern)")
    print("# npat  text_bytes  auto_ms kind  seeded_ms  tiled_ms  per_pattern_ms  tiled_kernel_ms  char*pat/s(tiled kernel)  matches")
    modes = {"auto": {}, "seeded": {"SASSY_HIP_SEEDED": "1"}, "tiled": {"SASSY_HIP_SEEDED": "0", "SASSY_HIP_TILED": "1"},
             "chains": {"SASSY_HIP_SEEDED": "0", "SASSY_HIP_TILED": "0"}}
    for npat in (64, 1000, 10000):
        pats = [bytes(rng.choice(b"ACGT") for _ in range(args.m)) for _ in range(npat)]
        for n in (10_000, 1 << 20, 16 << 20, 64 << 20, 256 << 20):
            res = {}
            for mode, env in modes.items():
                for key in ("SASSY_HIP_SEEDED", "SASSY_HIP_TILED"):
                    os.environ.pop(key, None)
                os.environ.update(env)
                s = sassy_amd.Searcher(args.profile, rc=False)
                enc = s.encode_patterns(pats)
                est = npat * 60e-6 + npat * n / 2e12 if mode == "chains" else (n * npat / 2e12 if mode == "tiled" else 0)
                if est > args.max_per_pattern_s:
                    res[mode] = (float("nan"), -1, 0.0, -1)
                    continue
                best = 1e30
                for rep in range(3):
                    t0 = time.perf_counter()
                    r = s.search_encoded_patterns(enc, DevText(buf.ptr, n), args.k, as_result=True)
                    best = min(best, time.perf_counter() - t0)
                st = s.stats()
                res[mode] = (best * 1e3, len(r), st["scan_ms"], st["filtered"])
            counts = {v[1] for v in res.values() if v[1] >= 0}
            assert len(counts) == 1, res
            kern_ms = res["tiled"][2] if res["tiled"][3] == 5 else 0.0
            rate = (n * npat / (kern_ms * 1e-3)) if kern_ms else 0.0
            print(f"{npat:6d} {n:11d} {res['auto'][0]:8.2f} {res['auto'][3]:4d} {res['seeded'][0]:10.2f} {res['tiled'][0]:9.2f} "
                  f"{res['chains'][0]:15.2f} {kern_ms:16.3f} {rate:12.3e} {counts.pop():8d}", flush=True)
    for key in ("SASSY_HIP_SEEDED", "SASSY_HIP_TILED"):
        os.environ.pop(key, None)


if __name__ == "__main__":
    main()
