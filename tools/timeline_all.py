"""Everything the GPU did during the LAST call of a command, from rocprofv3 --kernel-trace --memory-copy-trace CSVs: kernels
and copies in start order with their durations and the idle time in front of each (a call = the events after the last
idle gap longer than --gap-ms).   python tools/timeline_all.py <dir> [--gap-ms 20]"""
import csv, glob, os, sys

root = sys.argv[1]
gap_ms = float(sys.argv[sys.argv.index("--gap-ms") + 1]) if "--gap-ms" in sys.argv else 20.0
ev = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("sassy_hip::", "")[:60]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 8):i]) > gap_ms * 1e6:
        cut = i
call = ev[cut:]
t0 = call[0][0]
busy_end = t0
print(f"{len(call)} events, {(-t0 + max(e[1] for e in call)) / 1e6:.3f} ms from the first start to the last end")
for a, b, n in call:
    idle = (a - busy_end) / 1e3
    print(f"  t={(a - t0) / 1e6:8.3f} ms  dur {(b - a) / 1e3:9.1f} us  idle before {idle:9.1f} us  {n}")
    busy_end = max(busy_end, b)
