"""The CLI front end on a multi-GB FASTA: reader alone (GB/s) and FASTA -> TSV end to end, beside the link's ceiling.

    python tools/bench_cli.py [--bytes 3e9] [--records 24] [--width 60] [--dir /tmp]

Writes a synthetic wrapped FASTA (seeded random ACGT, `records` records, `width` bases per line) with one near-match of the
pattern planted per MiB, then times (1) sassy_amd.fastx.read_fastx_batches over it (unwrap to one buffer + offsets),
(2) `python -m sassy_amd search` in this process (stdout to a file), beside the time the PCIe link needs for as many bytes at its
measured ceiling (bench.py: h2d_inclusive).  One JSON line.
"""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import sassy_amd  # noqa: E402
from bench import _dna_bytes  # noqa: E402
from sassy_amd import cli, fastx  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=float, default=3e9)
    ap.add_argument("--records", type=int, default=24)
    ap.add_argument("--width", type=int, default=60)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("-k", type=int, default=3)
    ap.add_argument("--link-gb-per-s", type=float, default=57.5, help="the PCIe link's measured ceiling (bench.py: h2d_inclusive)")
    args = ap.parse_args()
    n = int(args.bytes) // (args.records * args.width) * (args.records * args.width)
    pat = bytes(_dna_bytes(43, 0, 32))
    # the text on the device (generator + plants), down to the host, out as wrapped FASTA
    buf = sassy_amd.DeviceBuffer(n + 4096)
    sassy_amd.generate_dna(buf.ptr, n, 42, 0)
    sassy_amd.plant(buf.ptr, n, 0, n, 42, pat, args.k, 1 << 20)
    host = np.empty(n, dtype=np.uint8)
    buf.download_into(host)
    buf.free()
    path = os.path.join(args.dir, "sassy_bench_cli.fa")
    per = n // args.records
    with open(path, "wb") as fh:
        for r in range(args.records):
            fh.write(b">chr%d synthetic\n" % (r + 1))
            rows = host[r * per:(r + 1) * per].reshape(-1, args.width)
            out = np.empty((rows.shape[0], args.width + 1), dtype=np.uint8)
            out[:, :args.width] = rows
            out[:, args.width] = 10
            out.tofile(fh)
    fsize = os.path.getsize(path)
    del host
    # (1) the reader alone
    t0 = time.perf_counter()
    nrec = nbytes = 0
    for rb in fastx.read_fastx_batches(path, cli.BATCH_BYTES):
        nrec += len(rb)
        nbytes += rb.text_bytes
    t_read = time.perf_counter() - t0
    assert nbytes == n and nrec == args.records, (nbytes, n, nrec)
    # (2) FASTA -> TSV
    tsv = os.path.join(args.dir, "sassy_bench_cli.tsv")
    argv = ["search", "-p", pat.decode(), "-k", str(args.k), "-a", "dna", "--no-rc", path]
    real = sys.stdout
    times = []
    for _ in range(2):  # (the first call also loads the kernels and sizes the device buffers)
        with open(tsv, "w") as fh:
            sys.stdout = fh
            try:
                t0 = time.perf_counter()
                cli.main(argv)
                times.append(time.perf_counter() - t0)
            finally:
                sys.stdout = real
    rows = sum(1 for _ in open(tsv)) - 1
    # (3) the link's ceiling for as many bytes: bench.py measures it on every run (h2d_inclusive.link_ceiling_GB_per_s, 1 GiB
    # from pinned memory: 57.5 GB/s on these boxes); quoted here, not measured again
    t_link = n / (args.link_gb_per_s * 1e9)
    print(json.dumps({
        "workload": f"{fsize} B FASTA ({args.records} records, {args.width} bases per line), Dna, |pattern|=32, k={args.k}, forward",
        "reader_seconds": round(t_read, 3), "reader_GB_per_s": round(fsize / t_read / 1e9, 2),
        "fasta_to_tsv_seconds": round(min(times), 3), "fasta_to_tsv_first_call_seconds": round(times[0], 3),
        "fasta_to_tsv_GB_per_s": round(fsize / min(times) / 1e9, 2), "tsv_rows": rows,
        "link_seconds_same_bytes_at_the_measured_ceiling": round(t_link, 3),
        "ratio_to_link": round(min(times) / t_link, 1),
    }))
    os.remove(path)
    os.remove(tsv)


if __name__ == "__main__":
    main()
