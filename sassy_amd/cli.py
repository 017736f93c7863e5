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
import gzip
import sys
from typing import Iterator, List, Tuple

from . import Searcher

BATCH_BYTES = 64 << 20  # text bytes per search_many call


def read_fastx(path: str) -> Iterator[Tuple[str, bytes]]:
    """(id, sequence) of every FASTA / FASTQ record; id = the header line without its marker, as
    needletail's `id()` returns it (bin/input_iterator.rs:128)."""
    if path in ("", "-"):
        fh = sys.stdin.buffer
    else:
        fh = open(path, "rb")
        if fh.read(2) == b"\x1f\x8b":
            fh.close()
            fh = gzip.open(path, "rb")
        else:
            fh.seek(0)
    with fh:
        rid, chunks = None, []
        it = iter(fh)
        for line in it:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if rid is not None:
                    yield rid, b"".join(chunks)
                rid, chunks = line[1:].decode(), []
            elif line.startswith(b"@") and rid is None:
                # FASTQ: @id / sequence / + / quality (single-line records)
                seq = next(it).rstrip(b"\r\n")
                next(it)
                next(it)
                yield line[1:].decode(), seq
            elif rid is not None:
                chunks.append(line)
        if rid is not None:
            yield rid, b"".join(chunks)


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

    def flush(batch):
        # every pattern against every record of the batch in one call (many short records -- reads --
        # share one device buffer); rows record by record, patterns in input order
        if not batch:
            return
        ms = searcher.search_many(pats, [seq for _, seq in batch], args.k)
        ms.sort(key=lambda m: (m.text_idx, m.pattern_idx))  # stable: keeps each pair's match order
        for m in ms:
            text_id, seq = batch[m.text_idx]
            out.write(searcher.format_tsv(m, patterns[m.pattern_idx][0], text_id, seq, sam=args.sam))

    batch, batch_bytes = [], 0
    for path in args.paths:
        for rec in read_fastx(path):
            batch.append(rec)
            batch_bytes += len(rec[1])
            if batch_bytes >= BATCH_BYTES:  # the reference batches ~1 MB of records per task (bin/input_iterator.rs:6)
                flush(batch)
                batch, batch_bytes = [], 0
    flush(batch)
    return 0
