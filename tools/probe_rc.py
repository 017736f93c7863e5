"""Both-strand search of the 3 GB bench text: where the time goes (rocprofv3 this for the kernel split)."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, sassy_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
buf = sassy_amd.DeviceBuffer(n + 64)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
r = np.random.default_rng(43)
class DevText:
    """A device-resident text for the ctypes mirror (what a CUDA tensor looks like to it)."""
    is_cuda = True
    class dtype:
        itemsize = 1
    def __init__(self, ptr, n): self._p, self._n = ptr, n
    def data_ptr(self): return self._p
    def numel(self): return self._n
    def is_contiguous(self): return True
text = DevText(buf.ptr, n)
import os
M, K = int(os.environ.get("PROBE_M", "32")), int(os.environ.get("PROBE_K", "3"))
pat = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, M)])
for profile in ("dna", "iupac"):
    for rc in (False, True):
        s = sassy_amd.Searcher(profile, rc=rc)
        for _ in range(15):  # (lone searches settle over their first calls: clocks)
            s.search(pat, text, K)
        t0 = time.perf_counter()
        for _ in range(10):
            res = s.search(pat, text, K)
        dt = (time.perf_counter() - t0) / 10
        print(json.dumps({"profile": profile, "m": M, "k": K, "rc": rc, "ms": round(dt * 1e3, 3), "matches": len(res)}), flush=True)
        if rc:
            s.text_unchanged(True)
            for _ in range(10):
                s.search(pat, text, K)
            t0 = time.perf_counter()
            for _ in range(10):
                res = s.search(pat, text, K)
            dt = (time.perf_counter() - t0) / 10
            print(json.dumps({"profile": profile, "m": M, "k": K, "rc": rc, "text_unchanged": True, "ms": round(dt * 1e3, 3)}), flush=True)
