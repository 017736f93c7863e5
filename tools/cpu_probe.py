"""How many host cores does this box really give us?  os.cpu_count() counts the machine's CPUs; a
container may be limited by a cgroup quota or an affinity mask.  Prints both and the scaling of the
CPU baseline's threaded scan (oracle/sassy_refstyle.c: rs_scan_mt) over the thread count."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def cgroup_cpus():
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
        except OSError:
            continue
        if path.endswith("cpu.max"):
            if txt[0] == "max":
                return None
            return float(txt[0]) / float(txt[1])
        q = float(txt[0])
        if q <= 0:
            return None
        return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    return None


info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpus": cgroup_cpus(),
        "loadavg": os.getloadavg()}
print(json.dumps(info), flush=True)
n = 1 << 30
text = oracle.generate_dna(42, 0, n)
pat = bytes(oracle.generate_dna(43, 0, 32))
for T in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if T > 2 * (os.cpu_count() or 1):
        break
    ends, i = oracle.refstyle_ends_mt("dna", pat, text, 3, T, 1.0)
    gb = n * i["passes"] / i["seconds"] / 1e9
    print(json.dumps({"threads": T, "gbps": round(gb, 2), "per_thread": round(gb / T, 3), "passes": i["passes"],
                      "busy_frac": round(i["busy_seconds"] / (i["seconds"] * T), 3)}), flush=True)
