"""`python -m sassy_amd search ...` -- the match-table front end of the reference CLI on the GPU path.

Mirrors `sassy search` (reference: bin/grep.rs:30-157 arguments, :465-470 header, :623-660 pattern
sources, :710-757 rows; FASTA/FASTQ records as bin/input_iterator.rs:125-137 reads them): every
pattern against every record of the given FASTA / FASTQ files (plain or gzip), one TSV row per match:

    pat_id  text_id  cost  strand  start  end  match_region  cigar

Defaults follow the reference: alphabet iupac, reverse complement on, max_n_frac 0.2.  Rows come
text record by text record, patterns in input order (the reference's order depends on its thread
scheduling).  Not mirrored: grep / filter output modes, --v2, threads.
"""
from __future__ import annotations

import argparse
import sys
from typing import List, Tuple

from . import Searcher

BATCH_BYTES = 64 << 20  # input bytes per search_many call (a longer record is a batch of its own; the reader reuses its buffers)


# (the reader: sassy_amd/fastx.py -- batches of whole records as one buffer + offsets, numpy over an mmap)
from .fastx import read_fastx, read_fastx_batches  # noqa: E402,F401


def load_patterns(args) -> List[Tuple[str, bytes]]:
    if args.pattern is not None:
        return [("pattern", args.pattern.encode())]          # bin/grep.rs:625-631
    if args.pattern_file is not None:
        with open(args.pattern_file, "rb") as fh:            # one pattern per line, ids 1, 2, ...
            return [(str(i + 1), line.rstrip(b"\r\n")) for i, line in enumerate(fh)]
    if args.pattern_fasta is not None:
        return list(read_fastx(args.pattern_fasta))
    raise SystemExit("No --pattern, --pattern-file, or --pattern-fasta provided!")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m sassy_amd", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    sp = sub.add_parser("search", help="write all matches as TSV to stdout")
    g = sp.add_mutually_exclusive_group()
    g.add_argument("-p", "--pattern")
    g.add_argument("-l", "--pattern-file")
    g.add_argument("-f", "--pattern-fasta")
    sp.add_argument("-k", type=int, required=True)
    sp.add_argument("-a", "--alphabet", choices=["dna", "iupac"], default="iupac")
    sp.add_argument("--overhang", type=float, default=None,
                    help="cost per base of overhang alignment in [0, 1] (iupac only); default disabled")
    sp.add_argument("--no-rc", action="store_true")
    sp.add_argument("--max-n-frac", type=float, default=0.2)
    sp.add_argument("--sam", action="store_true")
    sp.add_argument("paths", nargs="+")
    args = ap.parse_args(argv)

    patterns = load_patterns(args)
    searcher = Searcher(args.alphabet, rc=not args.no_rc, alpha=args.overhang).with_max_n_frac(args.max_n_frac)
    out = sys.stdout
    out.write("pat_id\ttext_id\tcost\tstrand\tstart\tend\tmatch_region\tcigar\n")
    pats = [p for _, p in patterns]

    import numpy as np

    # every pattern against every record of a batch in one call (the records of a batch are one buffer + offsets:
    # fastx.RecordBatch -- nothing is copied per record, nothing is split but the stream of records into batches);
    # rows record by record, patterns in input order
    for path in args.paths:
        for batch in read_fastx_batches(path, BATCH_BYTES):
            if not len(batch):
                continue
            res = searcher.search_many(pats, batch.texts, args.k, as_result=True)
            arr = res.array
            order = np.lexsort((np.arange(len(arr)), arr["pattern_idx"], arr["text_idx"]))  # stable: keeps each pair's match order
            ms = res.lazy_matches
            base = int(batch.texts.buffer.ctypes.data)
            for i in order.tolist():
                m = ms[i]
                ti = m.text_idx
                where = (base + int(batch.texts.starts[ti]), int(batch.texts.lens[ti]))
                out.write(searcher.format_tsv(m, patterns[m.pattern_idx][0], batch.id(ti), where, sam=args.sam))
    return 0
