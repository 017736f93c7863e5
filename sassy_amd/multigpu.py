"""Multi-GPU driver of the search path: one process per GPU, text sharded by position, one
collective exchange of the (tiny) match lists at the end.

The reference has no distributed code (it is a single-process CPU library; its thread-level
parallelism over independent records is the model: bin/grep.rs:476-503).  What shards here is
what SURVEY 8(e) names: the text is cut into G contiguous shards; shard g owns the end positions
whose 64-byte block lies inside it and scans from `halo` bytes to its left; no data-path
collective is needed, only a gather of match records to rank 0 (torch.distributed backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Payload is kilobytes: this
is latency-bound, never link-bound.

Cross-shard exactness: a <=k plateau that runs across a shard border is resolved exactly like
across lane chunks inside one GPU -- every shard reports its exit state (decreasing TRUE / FALSE /
PASS) and marks the one report that depends on its left neighbour; rank 0 walks the chain.

Records travel as numpy / torch arrays (one row per match), never as Python objects.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

STATE_FALSE, STATE_TRUE, STATE_PASS = 0, 1, 2
CIGAR_BYTES = 40
COLS = 7 + CIGAR_BYTES // 8  # int64 columns per packed match


def shard_bounds(total_len: int, world: int) -> List[Tuple[int, int]]:
    """[(start, end)) per rank: equal shares rounded up to whole 64-byte blocks."""
    per = -(-total_len // world)
    per = -(-per // 64) * 64
    out = []
    for r in range(world):
        a = min(r * per, total_len)
        b = min((r + 1) * per, total_len)
        out.append((a, b))
    return out


@dataclass
class ShardResult:
    """rows: int64 array [n, COLS] = pattern_idx, text_start, text_end, pattern_start,
    pattern_end, cost, strand, then CIGAR_BYTES bytes of NUL-padded cigar text."""
    rows: np.ndarray
    exit_state: int          # STATE_*
    conditional_index: int   # row of the report that depends on the left shard, or -1

    def __len__(self):
        return int(self.rows.shape[0])


def pack_result(result, out: Optional[np.ndarray] = None) -> ShardResult:
    """sassy_amd.Result -> ShardResult: the C-ABI's row packer (sassy_hip_pack_rows), straight into
    `out` (e.g. a pinned staging buffer) when given."""
    a = result.array
    n = len(a)
    rows = out[:n] if out is not None and out.shape[0] >= n else np.empty((n, COLS), dtype=np.int64)
    if n:
        from . import lib, _check
        _check(lib().sassy_hip_pack_rows(a.ctypes.data, n, result.pool, len(result.pool), rows.ctypes.data, CIGAR_BYTES))
    return ShardResult(rows, result.exit_state, result.conditional_index)


def pack_result_numpy(result) -> ShardResult:
    """The same in numpy (kept as the cross-check of the C packer in the tests)."""
    a = result.array
    n = len(a)
    rows = np.zeros((n, COLS), dtype=np.int64)
    if n:
        # text_start, text_end, pattern_start, pattern_end are bytes 16..47 of the 64-byte record
        raw = a.view(np.uint8).reshape(n, 64)
        rows[:, 0] = a["pattern_idx"].view(np.int64)
        rows[:, 1:5] = raw[:, 16:48].view(np.int64).reshape(n, 4)
        rows[:, 5] = a["cost"]
        rows[:, 6] = a["strand"]
        clen = a["cigar_len"].astype(np.int64)
        if int(clen.max(initial=0)) > CIGAR_BYTES:
            raise ValueError("cigar longer than the fixed gather field")
        pool = np.frombuffer(result.pool, dtype=np.uint8) if result.pool else np.zeros(1, np.uint8)
        off = a["cigar_off"].astype(np.int64)
        stride = int(off[1] - off[0]) if n > 1 else 0
        if n > 1 and stride >= CIGAR_BYTES and len(pool) >= int(off[-1]) + CIGAR_BYTES and \
                bool((off == off[0] + stride * np.arange(n, dtype=np.int64)).all()):
            # the device wrote the cigar strings into equally spaced, NUL-padded slots: one strided view
            cig = np.lib.stride_tricks.as_strided(pool[int(off[0]):], shape=(n, CIGAR_BYTES), strides=(stride, 1))
            keep = np.arange(CIGAR_BYTES, dtype=np.int64)[None, :] < clen[:, None]
            cig = np.where(keep, cig, 0).astype(np.uint8)
        else:
            idx = off[:, None] + np.arange(CIGAR_BYTES, dtype=np.int64)[None, :]
            keep = np.arange(CIGAR_BYTES, dtype=np.int64)[None, :] < clen[:, None]
            cig = np.where(keep, pool[np.minimum(idx, len(pool) - 1)], 0).astype(np.uint8)
        rows[:, 7:] = np.ascontiguousarray(cig).view(np.int64)
    return ShardResult(rows, result.exit_state, result.conditional_index)


def rows_from_matches(matches) -> np.ndarray:
    """Match objects -> packed rows (tests / small inputs)."""
    rows = np.zeros((len(matches), COLS), dtype=np.int64)
    for i, m in enumerate(matches):
        cig = m.cigar.encode()
        if len(cig) > CIGAR_BYTES:
            raise ValueError("cigar longer than the fixed gather field")
        r = rows[i]
        r[0], r[1], r[2], r[3], r[4] = m.pattern_idx, _s64(m.text_start), _s64(m.text_end), \
            _s64(m.pattern_start), _s64(m.pattern_end)
        r[5], r[6] = m.cost, 1 if m.strand == "-" else 0
        r[7:].view(np.uint8)[: len(cig)] = np.frombuffer(cig, dtype=np.uint8)
    return rows


def matches_from_rows(rows: np.ndarray, Match) -> list:
    out = []
    for row in rows:
        cig = bytes(row[7:].view(np.uint8)).rstrip(b"\0").decode()
        out.append(Match(int(row[0]), _u64(row[1]), _u64(row[2]), _u64(row[3]), _u64(row[4]),
                         int(row[5]), "-" if row[6] else "+", cig))
    return out


def _s64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _u64(v) -> int:
    v = int(v)
    return v + (1 << 64) if v < 0 else v


def merge_shard_results(shards: Sequence[ShardResult]) -> np.ndarray:
    """Concatenate shard rows in text order, dropping conditional reports whose plateau was
    entered by an increase (decreasing = FALSE arriving from the left)."""
    parts = []
    incoming = STATE_TRUE  # column 0 of the text: decreasing = true (src/search.rs:1055)
    for sh in shards:
        rows = sh.rows
        if sh.conditional_index >= 0 and incoming != STATE_TRUE:
            rows = np.delete(rows, sh.conditional_index, axis=0)
        parts.append(rows)
        if sh.exit_state != STATE_PASS:
            incoming = sh.exit_state
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, COLS), dtype=np.int64)


def gather_shard_results(local, torch, dist, device) -> Optional[List[ShardResult]]:
    """The one exchange of the path: all ranks' match lists to rank 0.
    Two collectives: all_gather of the 3-word headers (count, exit state, conditional index),
    then gather of the records padded to the largest count.  `local` is a ShardResult or a
    sassy_amd.Result; a Result is packed while the header exchange is in flight."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = len(local)
    head = torch.tensor([n_local, local.exit_state, local.conditional_index], dtype=torch.int64, device=device)
    heads = [torch.empty_like(head) for _ in range(world)]
    work = dist.all_gather(heads, head, async_op=True)
    if not isinstance(local, ShardResult):
        local = pack_result(local)
    work.wait()
    heads = torch.stack(heads).cpu().tolist()
    counts = [h[0] for h in heads]
    maxc = max(counts)
    if maxc == 0:
        if rank != 0:
            return None
        return [ShardResult(np.zeros((0, COLS), np.int64), h[1], h[2]) for h in heads]
    padded = np.zeros((maxc, COLS), dtype=np.int64)
    padded[: len(local)] = local.rows
    mine = torch.from_numpy(padded).to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gather_list=bufs, dst=0)
    if rank != 0:
        return None
    return [ShardResult(bufs[r][: counts[r]].cpu().numpy(), heads[r][1], heads[r][2]) for r in range(world)]


class MatchGather:
    """The same exchange set up once for a stream of searches (bench.py): every rank's header and rows
    travel in ONE collective of fixed size -- `capacity_rows` rows per rank, chosen by the caller from
    what the workload can report -- through pinned staging buffers on both sides; no per-call
    allocation, no second round for the sizes.  A rank that has more rows than the capacity makes
    rank 0 raise (use gather_shard_results for unbounded lists)."""

    def __init__(self, torch, dist, device, capacity_rows: int):
        self.torch, self.dist, self.device = torch, dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.cap = int(capacity_rows)
        self.words = 3 + self.cap * COLS
        pin = device.type == "cuda"
        self.stage = torch.empty(self.words, dtype=torch.int64, pin_memory=pin)
        self.stage_np = self.stage.numpy()
        self.rows_np = self.stage_np[3:].reshape(self.cap, COLS)
        self.dev = torch.empty(self.words, dtype=torch.int64, device=device)
        self.copied = torch.cuda.Event() if pin else None
        if self.rank == 0:
            self.all_dev = torch.empty((self.world, self.words), dtype=torch.int64, device=device)
            self.all_host = torch.empty((self.world, self.words), dtype=torch.int64, pin_memory=pin)
            self.all_np = self.all_host.numpy()

    def gather(self, local) -> Optional[List[ShardResult]]:
        """local: a ShardResult or a sassy_amd.Result (packed straight into the staging buffer)."""
        if self.copied is not None:
            self.copied.synchronize()  # the previous call's upload has left the staging buffer
        n = len(local)
        if n <= self.cap:
            if isinstance(local, ShardResult):
                self.rows_np[:n] = local.rows
            else:
                pack_result(local, out=self.rows_np)
        self.stage_np[0], self.stage_np[1], self.stage_np[2] = n, local.exit_state, local.conditional_index
        self.dev.copy_(self.stage, non_blocking=True)
        if self.copied is not None:
            self.copied.record()
        if self.rank != 0:
            self.dist.gather(self.dev, dst=0)
            return None
        self.dist.gather(self.dev, gather_list=list(self.all_dev.unbind(0)), dst=0)
        self.all_host.copy_(self.all_dev)  # one device -> host copy (synchronous)
        out = []
        for r in range(self.world):
            cnt = int(self.all_np[r, 0])
            if cnt > self.cap:
                raise OverflowError(f"rank {r} reports {cnt} matches, gather capacity is {self.cap}")
            rows = self.all_np[r, 3:3 + cnt * COLS].reshape(cnt, COLS)
            out.append(ShardResult(rows, int(self.all_np[r, 1]), int(self.all_np[r, 2])))
        return out


class GatherWorker:
    """Runs MatchGather.gather + merge_shard_results on a worker thread, in submission order (every
    rank issues its collectives in the same order), so that the exchange of search i overlaps
    search i+1.  flush() returns when everything submitted has been gathered."""

    def __init__(self, gatherer: MatchGather):
        import queue
        import threading
        self.g = gatherer
        self.q = queue.Queue()
        self.last = None
        self.error = None
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        if self.g.device.type == "cuda":
            self.g.torch.cuda.set_device(self.g.device)
        while True:
            item = self.q.get()
            try:
                if item is None:
                    return
                if self.error is None:
                    shards = self.g.gather(item)
                    if shards is not None:
                        self.last = merge_shard_results(shards).copy()
            except BaseException as e:  # surfaced by flush()
                self.error = e
            finally:
                self.q.task_done()

    def submit(self, local):
        self.q.put(local)

    def flush(self):
        self.q.join()
        if self.error is not None:
            raise self.error
        return self.last

    def close(self):
        self.q.put(None)
        self.t.join()
