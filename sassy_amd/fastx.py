"""Chunked FASTA / FASTQ input for the CLI front end: whole batches of records as ONE buffer + offsets (`TextBatch`), parsed
with numpy over an mmap of the file (or a zlib stream for .gz) instead of a Python loop over lines.

What the reference does per record with needletail (bin/input_iterator.rs:105-209: records are read one by one and handed
out in batches of about 1 MB), here per batch of `batch_bytes`:

  * records whose sequence is ONE line -- FASTQ, unwrapped FASTA, reads -- are not copied at all: the batch's buffer is
    the file's own bytes (the mmap), text i = the bytes between two newlines, and `search_many` gets those addresses;
  * wrapped FASTA (a genome: 60 or 80 bases per line) is unwrapped record by record with one strided copy
    (rows of `width + 1` bytes, the newline column dropped) after one strided compare has checked that the lines are
    regular; a record with ragged lines falls back to a mask over its bytes;
  * nothing is ever split but the stream of records into batches: a record is one text, however long.

Ids stay in the file's bytes until somebody asks for one (`RecordBatch.id(i)`): a read set has a million of them, and
only the records with a match are printed.
"""
from __future__ import annotations

import mmap
import os
import zlib
from typing import Iterator, List, Optional, Tuple

import numpy as np

from . import TextBatch

NL, CR, GT, AT, PLUS = 10, 13, 62, 64, 43


class RecordBatch:
    """Records of one batch: `texts` (a TextBatch over `buffer`), ids on demand."""

    def __init__(self, texts: TextBatch, raw: np.ndarray, id_starts: np.ndarray, id_lens: np.ndarray):
        self.texts = texts
        self._raw, self._id_starts, self._id_lens = raw, id_starts, id_lens

    def __len__(self) -> int:
        return len(self.texts)

    def id(self, i: int) -> str:
        a, n = int(self._id_starts[i]), int(self._id_lens[i])
        return self._raw[a:a + n].tobytes().decode()

    def sequence(self, i: int) -> bytes:
        a, n = int(self.texts.starts[i]), int(self.texts.lens[i])
        return self.texts.buffer[a:a + n].tobytes()

    @property
    def text_bytes(self) -> int:
        return int(self.texts.lens.sum())


def _strip_cr(raw: np.ndarray, starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Line lengths without a trailing carriage return."""
    has = (lens > 0) & (raw[np.minimum(starts + lens - 1, raw.size - 1).astype(np.int64)] == CR)
    return lens - has.astype(lens.dtype)


def _line_table(raw: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(start, length) of every line of `raw` (the last one may lack its newline)."""
    nl = np.flatnonzero(raw == NL)
    starts = np.empty(nl.size + 1, dtype=np.int64)
    starts[0] = 0
    starts[1:] = nl + 1
    ends = np.empty(nl.size + 1, dtype=np.int64)
    ends[:-1] = nl
    ends[-1] = raw.size
    if starts[-1] >= raw.size:  # the file ends with a newline: no line behind it
        starts, ends = starts[:-1], ends[:-1]
    return starts, ends - starts


def _parse_single_line_records(raw: np.ndarray, lines_per_record: int) -> Optional[RecordBatch]:
    """FASTQ (4 lines per record) or unwrapped FASTA (2): the sequences are lines of `raw` -- no copy.  None if the chunk
    does not have that shape (multi-line records: the general parsers take it)."""
    ls, ll = _line_table(raw)
    if ls.size == 0 or ls.size % lines_per_record:
        return None
    marker = AT if lines_per_record == 4 else GT
    if not (raw[ls[0::lines_per_record]] == marker).all():
        return None
    if lines_per_record == 4 and not (raw[ls[2::4]] == PLUS).all():
        return None
    if lines_per_record == 2 and (raw[ls[1::2]] == GT).any():
        return None  # (an empty record followed by a header: not this shape)
    ll = _strip_cr(raw, ls, ll)
    seq_s, seq_l = ls[1::lines_per_record], ll[1::lines_per_record]
    return RecordBatch(TextBatch(raw, seq_s.astype(np.uint64), seq_l.astype(np.uint64)), raw, ls[0::lines_per_record] + 1,
                       ll[0::lines_per_record] - 1)


def _big(n: int):
    """n bytes of anonymous memory, huge pages asked for: (mmap object, numpy view).  The first touch of ordinary pages is
    the slowest thing a reader can do (0.25 GB/s in this container, 1.8 with huge pages; a touched page: 13)."""
    m = mmap.mmap(-1, max(n, 1))
    try:
        m.madvise(mmap.MADV_HUGEPAGE)
    except (AttributeError, OSError, ValueError):
        pass
    return m, np.frombuffer(m, dtype=np.uint8)


class _Work:
    """Buffers that are reused from batch to batch.  The file's bytes of batch i are read into raw[i % 2], the unwrapped
    text of a wrapped-FASTA batch lives in out[i % 2]: a batch is valid until the batch after the next one is produced."""

    def __init__(self):
        self.flag = np.empty(0, dtype=bool)
        self.out = [np.empty(0, dtype=np.uint8), np.empty(0, dtype=np.uint8)]
        self.raw = [_big(0), _big(0)]
        self.turn = 0

    def flags(self, n: int) -> np.ndarray:
        if self.flag.size < n:
            self.flag = _big(n + n // 8)[1].view(bool)
        return self.flag[:n]

    def output(self, n: int) -> np.ndarray:
        if self.out[self.turn].size < n:
            self.out[self.turn] = _big(n + n // 8)[1]
        return self.out[self.turn]

    def raw_buffer(self, n: int, keep: int = 0):
        """The current turn's raw buffer with room for n bytes (its first `keep` bytes survive a growth)."""
        m, a = self.raw[self.turn]
        if a.size < n:
            m2, a2 = _big(n + n // 4)
            a2[:keep] = a[:keep]
            self.raw[self.turn] = (m2, a2)
        return self.raw[self.turn]


def _count(seg: np.ndarray, byte: int, work: _Work) -> int:
    f = work.flags(seg.size)
    np.equal(seg, byte, out=f)
    return int(np.count_nonzero(f))


def _unwrap_record(raw: np.ndarray, a: int, b: int, out: np.ndarray, at: int, work: _Work) -> int:
    """The sequence lines raw[a:b] of one FASTA record, without their line ends, into out[at:]; returns the length."""
    if b <= a:
        return 0
    seg = raw[a:b]
    hits = np.flatnonzero(seg[:1 << 16] == NL)
    if not hits.size and seg.size > (1 << 16):
        hits = np.flatnonzero(seg == NL)
    first = int(hits[0]) if hits.size else seg.size
    if first >= seg.size:  # one line without a newline
        n = seg.size - (1 if seg.size and seg[-1] == CR else 0)
        out[at:at + n] = seg[:n]
        return n
    crlf = first > 0 and seg[first - 1] == CR
    width = first - (1 if crlf else 0)      # bases per full line
    row = first + 1                         # bytes per full line
    full = seg.size // row
    if width > 0 and full > 0:
        body = seg[:full * row].reshape(full, row)
        tail = seg[full * row:]
        tail_nl = int((tail == NL).sum())
        # regular lines: every row ends with its newline, and the record holds no other one (nor a stray CR) -- the newline
        # column is a strided compare, the count one pass into a reused flag array
        # (a CR that is not part of a line end stays in the text, as it does for a line-by-line reader)
        if tail_nl <= 1 and (body[:, row - 1] == NL).all() and _count(seg, NL, work) == full + tail_nl and \
                (not crlf or (body[:, width] == CR).all()):
            keep = tail[(tail != NL) & (tail != CR)]
            if keep.size <= width:
                out[at:at + full * width].reshape(full, width)[:, :] = body[:, :width]
                n = full * width
                out[at + n:at + n + keep.size] = keep
                return n + int(keep.size)
    return _unwrap_masked(seg, out, at)


def _unwrap_masked(seg: np.ndarray, out: np.ndarray, at: int) -> int:
    keep = seg[(seg != NL) & (seg != CR)]
    out[at:at + keep.size] = keep
    return int(keep.size)


def _parse_fasta_chunk(raw: np.ndarray, work: _Work, buf=None) -> RecordBatch:
    """Any FASTA chunk that starts with a header: wrapped records are unwrapped into a buffer of their own."""
    # record starts: '>' at a line start (memmem over the buffer: headers are rare)
    if buf is not None and raw.size:  # (raw = the first raw.size bytes of buf)
        found = [0] if raw[0] == GT else []
        p = buf.find(b"\n>", 0, raw.size)
        while p >= 0:
            found.append(p + 1)
            p = buf.find(b"\n>", p + 1, raw.size)
        gt = np.asarray(found, dtype=np.int64)
    else:
        f = work.flags(raw.size)
        np.equal(raw, GT, out=f)
        gt = np.flatnonzero(f)
        if gt.size:
            gt = gt[(gt == 0) | (raw[np.maximum(gt - 1, 0)] == NL)]
    n_rec = gt.size
    hdr_end = np.empty(n_rec, dtype=np.int64)
    for i in range(n_rec):  # (the header's end: the first newline behind it -- a short search per record)
        a = int(gt[i])
        seg = raw[a:a + 65536]
        hit = np.flatnonzero(seg == NL)
        if hit.size:
            hdr_end[i] = a + int(hit[0])
        else:
            hit = np.flatnonzero(raw[a:] == NL)
            hdr_end[i] = a + (int(hit[0]) if hit.size else raw.size - a)
    rec_end = np.empty(n_rec, dtype=np.int64)
    rec_end[:-1] = gt[1:]
    if n_rec:
        rec_end[-1] = raw.size
    out = work.output(raw.size)
    starts = np.zeros(n_rec, dtype=np.uint64)
    lens = np.zeros(n_rec, dtype=np.uint64)
    at = 0
    for i in range(n_rec):
        n = _unwrap_record(raw, int(hdr_end[i]) + 1, int(rec_end[i]), out, at, work)
        starts[i], lens[i] = at, n
        at += n
    id_lens = hdr_end - gt - 1
    if n_rec:
        id_lens = id_lens - (raw[np.maximum(hdr_end - 1, 0)] == CR)
    return RecordBatch(TextBatch(out[:at], starts, lens), raw, gt + 1, id_lens)


def _parse_chunk(raw: np.ndarray, fastq: bool, work: _Work, buf=None) -> RecordBatch:
    if fastq:
        rb = _parse_single_line_records(raw, 4)
        if rb is None:
            raise ValueError("multi-line FASTQ records are not supported")
        return rb
    # unwrapped FASTA (every record header + one line): no copy; else the general FASTA parser
    if raw.size and raw.size < (1 << 31):
        probe = raw[:1 << 16]
        nl = np.flatnonzero(probe == NL)
        if nl.size >= 2 and nl[1] + 1 < probe.size and probe[nl[0] + 1] != GT and probe[nl[1] + 1] == GT:
            rb = _parse_single_line_records(raw, 2)
            if rb is not None:
                return rb
    return _parse_fasta_chunk(raw, work, buf)


def _cut(buf, lo: int, hi: int, fastq: bool, at_eof: bool) -> int:
    """Where to end the chunk that starts at `lo` and may reach `hi`: the last record boundary in front of `hi`."""
    if at_eof:
        return hi
    if not fastq:
        p = buf.rfind(b"\n>", lo, hi)
        return p + 1 if p > lo else -1
    # FASTQ: records are four lines; a line that starts with '@' may be a quality line: count lines from the chunk's start
    view = np.frombuffer(buf, dtype=np.uint8, count=hi - lo, offset=lo)
    nl = np.flatnonzero(view == NL)
    full = nl.size // 4 * 4
    return lo + int(nl[full - 1]) + 1 if full else -1


def _sources(path: str):
    """A function fill(view) -> bytes written (0 at the end of the input) for a plain file, a gzip file or standard input."""
    if path in ("", "-"):
        import sys
        fh = sys.stdin.buffer
        head = fh.peek(2)[:2] if hasattr(fh, "peek") else b""
    else:
        fh = open(path, "rb", buffering=0)
        head = fh.read(2)
        fh.seek(0)
    if head != b"\x1f\x8b":
        return (lambda view: fh.readinto(view) or 0), fh
    state = {"d": zlib.decompressobj(wbits=31), "left": b"", "eof": False}

    def fill(view) -> int:
        out = 0
        while out < len(view):
            if state["left"]:
                take = state["left"][:len(view) - out]
                view[out:out + len(take)] = take
                state["left"] = state["left"][len(take):]
                out += len(take)
                continue
            if state["eof"]:
                break
            piece = fh.read(4 << 20)
            if not piece:
                state["left"] = state["d"].flush()
                state["eof"] = True
                continue
            data = state["d"].decompress(piece)
            while state["d"].eof and state["d"].unused_data:  # several gzip members in one file
                rest = state["d"].unused_data
                state["d"] = zlib.decompressobj(wbits=31)
                data += state["d"].decompress(rest)
            state["left"] = data
        return out

    return fill, fh


def read_fastx_batches(path: str, batch_bytes: int = 256 << 20) -> Iterator[RecordBatch]:
    """Batches of whole records, about `batch_bytes` of input each (a longer record is a batch of its own).  The bytes of a
    batch live in buffers that are reused: a batch is valid until the batch after the next one is produced."""
    work = _Work()
    fill, fh = _sources(path)
    try:
        carry = 0          # bytes of the next batch that are already in the current raw buffer (behind the last cut)
        fastq = None
        eof = False
        while True:
            want = batch_bytes
            m, raw = work.raw_buffer(carry + want, carry)
            have = carry
            while True:
                while have < carry + want and not eof:
                    got = fill(memoryview(m)[have:carry + want])
                    if got == 0:
                        eof = True
                    have += got
                if have == 0:
                    return
                if fastq is None:
                    first = bytes(raw[:1])
                    if first not in (b">", b"@"):
                        raise ValueError(f"{path}: neither FASTA nor FASTQ (first byte {first!r})")
                    fastq = first == b"@"
                end = _cut(m, 0, have, fastq, eof)
                if end > 0:
                    break
                want *= 2  # a record longer than the batch: read on
                m, raw = work.raw_buffer(carry + want, have)
            batch = _parse_chunk(raw[:end], fastq, work, m)
            # what lies behind the cut opens the next batch: into the other raw buffer
            rest = have - end
            work.turn ^= 1
            m2, raw2 = work.raw_buffer(rest + batch_bytes, 0)
            raw2[:rest] = raw[end:have]
            carry = rest
            yield batch
            if eof and carry == 0:
                return
    finally:
        if path not in ("", "-"):
            fh.close()


def read_fastx(path: str) -> Iterator[Tuple[str, bytes]]:
    """(id, sequence) of every record (the record-by-record view of read_fastx_batches)."""
    for rb in read_fastx_batches(path):
        for i in range(len(rb)):
            yield rb.id(i), rb.sequence(i)
