"""A text beyond 2^32 bytes (12 GB): every plant found at its place, on both strands, and a slice that
straddles the 2^32 byte border equal to the oracle's answer."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, sassy_amd, oracle
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_000_000_000
buf = sassy_amd.DeviceBuffer(n + 4096)
sassy_amd.generate_dna(buf.ptr, n, 42, 0)
pat = bytes(oracle.generate_dna(43, 0, 32))
planted = sassy_amd.plant(buf.ptr, n, 0, n, 42, pat, 3, stride=1 << 22)
class DevText:
    is_cuda = True
    class dtype: itemsize = 1
    def __init__(self, ptr, n): self._p, self._n = ptr, n
    def data_ptr(self): return self._p
    def numel(self): return self._n
    def is_contiguous(self): return True
for profile in ("dna", "iupac"):
    s = sassy_amd.Searcher(profile, rc=True)
    t0 = time.perf_counter()
    ms = s.search(pat, DevText(buf.ptr, n), 3)
    dt = time.perf_counter() - t0
    fwd = [m for m in ms if m.strand == "+"]
    slots = {m.text_start >> 22 for m in fwd}
    off = (1 << 32) - (1 << 20)
    sl = buf.download(1 << 21, off)
    want = oracle.search(profile, pat, sl, 3, rc=True)
    sub = sorted((m.text_start - off, m.text_end - off, m.cost, m.strand, m.cigar) for m in ms
                 if off + 64 <= m.text_start and m.text_end <= off + (1 << 21) - 64)
    exp = sorted((m.text_start, m.text_end, m.cost, m.strand, m.cigar) for m in want if m.text_start >= 64 and m.text_end <= (1 << 21) - 64)
    print(json.dumps({"profile": profile, "n": n, "seconds": round(dt, 4), "matches": len(ms), "planted": planted,
                      "plant_slots_found": len(slots), "max_end": max(m.text_end for m in ms),
                      "slice_equal": sub == exp, "slice_matches": len(exp), "filtered": s.stats()["filtered"]}))
    assert len(slots) >= planted - 1 and sub == exp
