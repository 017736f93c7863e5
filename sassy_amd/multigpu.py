"""Multi-GPU driver of the search path: one process per GPU, text sharded by position, one
collective exchange of the (tiny) match lists at the end.

The reference has no distributed code (it is a single-process CPU library; its thread-level
parallelism over independent records is the model: bin/grep.rs:476-503).  What shards here is
what SURVEY 8(e) names: the text is cut into G contiguous shards; shard g owns the end positions
whose 64-byte block lies inside it and scans from `halo` bytes to its left; no data-path
collective is needed, only an exchange of match records (torch.distributed backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  Payload is kilobytes: this is latency-bound,
never link-bound.

Cross-shard exactness: a <=k plateau that runs across a shard border is resolved exactly like
across lane chunks inside one GPU -- every shard reports its exit state (decreasing TRUE / FALSE /
PASS) and marks the one report that depends on its left neighbour; rank 0 walks the chain.

Records travel as numpy / torch arrays (one row per match), never as Python objects.  A row is
7 + cigar_bytes / 8 int64 words; the width of the cigar field follows from the search
(cigar_bytes_for(m, k): a cigar of a match of an m-row pattern with at most k edits never has more
than 2 (m + k + 1) + 2 characters -- the device's own string slot), it is not a constant.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

STATE_FALSE, STATE_TRUE, STATE_PASS = 0, 1, 2
FIXED_COLS = 7   # pattern_idx, text_start, text_end, pattern_start, pattern_end, cost, strand
HEAD_WORDS = 4   # rows, exit state, conditional index, error flag


def cigar_bytes_for(pattern_len: int, k: int) -> int:
    """Width of the cigar field that holds every cigar a search with this (m, k) can produce: the
    device's string slot 2 (m + k + 1) + 2 (host.hip: T.str_stride), in whole 8-byte words."""
    return (2 * (pattern_len + k + 1) + 2 + 7) // 8 * 8


def cols_for(cigar_bytes: int) -> int:
    if cigar_bytes % 8:
        raise ValueError("cigar_bytes must be a multiple of 8")
    return FIXED_COLS + cigar_bytes // 8


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
    """rows: int64 array [n, 7 + cigar_bytes / 8] = pattern_idx, text_start, text_end,
    pattern_start, pattern_end, cost, strand, then cigar_bytes bytes of NUL-padded cigar text."""
    rows: np.ndarray
    exit_state: int          # STATE_*
    conditional_index: int   # row of the report that depends on the left shard, or -1

    def __len__(self):
        return int(self.rows.shape[0])

    @property
    def cigar_bytes(self) -> int:
        return 8 * (int(self.rows.shape[1]) - FIXED_COLS)


def _needed_cigar_bytes(result) -> int:
    a = result.array
    longest = int(a["cigar_len"].max(initial=0)) if len(a) else 0
    return max(8, (longest + 7) // 8 * 8)


def pack_result(result, out: Optional[np.ndarray] = None, cigar_bytes: Optional[int] = None) -> ShardResult:
    """sassy_amd.Result -> ShardResult: the C-ABI's row packer (sassy_hip_pack_rows), straight into
    `out` (e.g. a pinned staging buffer, shape [cap, 7 + cigar_bytes / 8]) when given.  Without an
    explicit width the field is as wide as the longest cigar of this result."""
    a = result.array
    n = len(a)
    if out is not None:
        cigar_bytes = 8 * (int(out.shape[1]) - FIXED_COLS)
    elif cigar_bytes is None:
        cigar_bytes = _needed_cigar_bytes(result)
    cols = cols_for(cigar_bytes)
    rows = out[:n] if out is not None and out.shape[0] >= n else np.empty((n, cols), dtype=np.int64)
    if n:
        from . import lib, _check
        _check(lib().sassy_hip_pack_rows(a.ctypes.data, n, result.pool, len(result.pool), rows.ctypes.data, cigar_bytes))
    return ShardResult(rows, result.exit_state, result.conditional_index)


def pack_result_numpy(result, cigar_bytes: Optional[int] = None) -> ShardResult:
    """The same in numpy (kept as the cross-check of the C packer in the tests)."""
    a = result.array
    n = len(a)
    if cigar_bytes is None:
        cigar_bytes = _needed_cigar_bytes(result)
    rows = np.zeros((n, cols_for(cigar_bytes)), dtype=np.int64)
    if n:
        # text_start, text_end, pattern_start, pattern_end are bytes 16..47 of the 64-byte record
        raw = a.view(np.uint8).reshape(n, 64)
        rows[:, 0] = a["pattern_idx"].view(np.int64)
        rows[:, 1:5] = raw[:, 16:48].view(np.int64).reshape(n, 4)
        rows[:, 5] = a["cost"]
        rows[:, 6] = a["strand"]
        clen = a["cigar_len"].astype(np.int64)
        if int(clen.max(initial=0)) > cigar_bytes:
            raise ValueError("cigar longer than the gather field")
        pool = np.frombuffer(result.pool, dtype=np.uint8) if result.pool else np.zeros(1, np.uint8)
        off = a["cigar_off"].astype(np.int64)
        idx = off[:, None] + np.arange(cigar_bytes, dtype=np.int64)[None, :]
        keep = np.arange(cigar_bytes, dtype=np.int64)[None, :] < clen[:, None]
        cig = np.where(keep, pool[np.minimum(idx, len(pool) - 1)], 0).astype(np.uint8)
        rows[:, FIXED_COLS:] = np.ascontiguousarray(cig).view(np.int64)
    return ShardResult(rows, result.exit_state, result.conditional_index)


def rows_from_matches(matches, cigar_bytes: Optional[int] = None) -> np.ndarray:
    """Match objects -> packed rows (tests / small inputs)."""
    if cigar_bytes is None:
        longest = max((len(m.cigar) for m in matches), default=0)
        cigar_bytes = max(8, (longest + 7) // 8 * 8)
    rows = np.zeros((len(matches), cols_for(cigar_bytes)), dtype=np.int64)
    for i, m in enumerate(matches):
        cig = m.cigar.encode()
        if len(cig) > cigar_bytes:
            raise ValueError("cigar longer than the gather field")
        r = rows[i]
        r[0], r[1], r[2], r[3], r[4] = m.pattern_idx, _s64(m.text_start), _s64(m.text_end), \
            _s64(m.pattern_start), _s64(m.pattern_end)
        r[5], r[6] = m.cost, 1 if m.strand == "-" else 0
        r[FIXED_COLS:].view(np.uint8)[: len(cig)] = np.frombuffer(cig, dtype=np.uint8)
    return rows


def matches_from_rows(rows: np.ndarray, Match) -> list:
    out = []
    for row in rows:
        cig = bytes(row[FIXED_COLS:].view(np.uint8)).rstrip(b"\0").decode()
        out.append(Match(int(row[0]), _u64(row[1]), _u64(row[2]), _u64(row[3]), _u64(row[4]),
                         int(row[5]), "-" if row[6] else "+", cig))
    return out


def _s64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _u64(v) -> int:
    v = int(v)
    return v + (1 << 64) if v < 0 else v


def widen_rows(rows: np.ndarray, cols: int) -> np.ndarray:
    """Rows with a narrower cigar field padded (with NULs) to `cols` columns."""
    if rows.shape[1] == cols:
        return rows
    if rows.shape[1] > cols:
        raise ValueError("cannot narrow packed rows")
    out = np.zeros((rows.shape[0], cols), dtype=np.int64)
    out[:, : rows.shape[1]] = rows
    return out


def merge_shard_results(shards: Sequence[ShardResult]) -> np.ndarray:
    """Concatenate shard rows in text order, dropping conditional reports whose plateau was
    entered by an increase (decreasing = FALSE arriving from the left)."""
    parts = []
    incoming = STATE_TRUE  # column 0 of the text: decreasing = true (src/search.rs:1055)
    cols = max((int(sh.rows.shape[1]) for sh in shards), default=cols_for(8))
    for sh in shards:
        rows = widen_rows(sh.rows, cols)
        if sh.conditional_index >= 0 and incoming != STATE_TRUE:
            rows = np.delete(rows, sh.conditional_index, axis=0)
        parts.append(rows)
        if sh.exit_state != STATE_PASS:
            incoming = sh.exit_state
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, cols), dtype=np.int64)


def gather_shard_results(local, torch, dist, device) -> Optional[List[ShardResult]]:
    """The one exchange of the path for lists of unknown size: all ranks' match lists to rank 0.
    Two collectives: all_gather of the 4-word headers (count, exit state, conditional index, cigar
    field width), then gather of the records padded to the largest count and the widest field.
    `local` is a ShardResult or a sassy_amd.Result; a Result is packed while the header exchange is
    in flight."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = len(local)
    cb_local = local.cigar_bytes if isinstance(local, ShardResult) else _needed_cigar_bytes(local)
    head = torch.tensor([n_local, local.exit_state, local.conditional_index, cb_local], dtype=torch.int64, device=device)
    heads = [torch.empty_like(head) for _ in range(world)]
    work = dist.all_gather(heads, head, async_op=True)
    if not isinstance(local, ShardResult):
        local = pack_result(local, cigar_bytes=cb_local)
    work.wait()
    heads = torch.stack(heads).cpu().tolist()
    counts = [h[0] for h in heads]
    maxc = max(counts)
    cols = cols_for(max(h[3] for h in heads))
    if maxc == 0:
        if rank != 0:
            return None
        return [ShardResult(np.zeros((0, cols), np.int64), h[1], h[2]) for h in heads]
    padded = np.zeros((maxc, cols), dtype=np.int64)
    padded[: len(local), : local.rows.shape[1]] = local.rows
    mine = torch.from_numpy(padded).to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gather_list=bufs, dst=0)
    if rank != 0:
        return None
    return [ShardResult(bufs[r][: counts[r]].cpu().numpy(), heads[r][1], heads[r][2]) for r in range(world)]


class GatherError(RuntimeError):
    """Raised on EVERY rank when some rank flagged an error in its header: the ranks fail together
    instead of one leaving the others inside a collective."""


class MatchGather:
    """The same exchange set up once for a stream of searches (bench.py): the four-word headers of all ranks go to
    all ranks (one tiny all_gather: every rank must see every count, exit state and error flag to grow and to fail
    together), the match rows only to rank 0 (one gather of as many rows as the longest list holds) -- no rank but the
    merging one receives anybody's rows.  Pinned staging buffers, no per-call allocation.  The capacity is not
    derived from the workload: when some rank has more rows than fit, every rank sees that in the headers and all of
    them grow to the next power of two that holds the largest list before the rows move.

    cigar_bytes: width of the cigar field, cigar_bytes_for(m, k) of the searches to come."""

    def __init__(self, torch, dist, device, capacity_rows: int = 1024, cigar_bytes: int = 40):
        self.torch, self.dist, self.device = torch, dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.cols = cols_for(cigar_bytes)
        self.cigar_bytes = cigar_bytes
        self.pin = device.type == "cuda"
        self.copied = torch.cuda.Event() if self.pin else None
        self.regrown = 0
        self.head_stage = torch.empty(HEAD_WORDS, dtype=torch.int64, pin_memory=self.pin)
        self.head_np = self.head_stage.numpy()
        self.head_dev = torch.empty(HEAD_WORDS, dtype=torch.int64, device=device)
        self.heads_dev = torch.empty((self.world, HEAD_WORDS), dtype=torch.int64, device=device)
        self.heads_host = torch.empty((self.world, HEAD_WORDS), dtype=torch.int64, pin_memory=self.pin)
        self.heads_np = self.heads_host.numpy()
        self._alloc(max(1, int(capacity_rows)))

    def _alloc(self, cap: int):
        torch = self.torch
        self.cap = cap
        self.stage = torch.empty(cap * self.cols, dtype=torch.int64, pin_memory=self.pin)
        self.rows_np = self.stage.numpy().reshape(cap, self.cols)
        self.dev = torch.empty(cap * self.cols, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            self.all_dev = torch.empty((self.world, cap * self.cols), dtype=torch.int64, device=self.device)
            self.all_host = torch.empty((self.world, cap * self.cols), dtype=torch.int64, pin_memory=self.pin)
            self.all_np = self.all_host.numpy()

    def _all_gather_heads(self):
        d = self.dist
        if hasattr(d, "all_gather_into_tensor"):
            try:
                d.all_gather_into_tensor(self.heads_dev.view(-1), self.head_dev)
                return
            except (RuntimeError, NotImplementedError):  # a backend without the flat form
                pass
        d.all_gather(list(self.heads_dev.unbind(0)), self.head_dev)

    def gather(self, local, error: bool = False) -> Optional[List[ShardResult]]:
        """local: a ShardResult or a sassy_amd.Result (packed straight into the staging buffer), or
        None with error=True (this rank failed before the exchange: all ranks raise GatherError)."""
        if self.copied is not None:
            self.copied.synchronize()  # the previous call's upload has left the staging buffers
        n, state, cond = 0, STATE_PASS, -1
        overflow_rows = None  # this rank's rows when they do not fit the staging buffer yet
        if local is not None and not error:
            # Everything that can fail on this rank alone -- reading the result, packing its rows (a cigar wider than
            # cigar_bytes, a result already freed) -- happens HERE, in front of the header exchange: the error flag
            # travels in the header and every rank raises GatherError together.  A rank that raised between the two
            # collectives would leave the others inside dist.gather for good.
            try:
                n, state, cond = len(local), local.exit_state, local.conditional_index
                if n:
                    dst = self.rows_np if n <= self.cap else np.empty((n, self.cols), dtype=np.int64)
                    if isinstance(local, ShardResult):
                        dst[:n] = widen_rows(local.rows, self.cols)
                    else:
                        pack_result(local, out=dst)
                    if n > self.cap:
                        overflow_rows = dst
            except Exception:  # keep the collectives going: the header tells everybody
                error, n = True, 0
        self.head_np[0], self.head_np[1], self.head_np[2], self.head_np[3] = n, state, cond, 1 if error else 0
        self.head_dev.copy_(self.head_stage, non_blocking=True)
        self._all_gather_heads()
        self.heads_host.copy_(self.heads_dev)  # synchronous device -> host copy
        if int(self.heads_np[:, 3].max()) != 0:
            bad = [r for r in range(self.world) if self.heads_np[r, 3]]
            raise GatherError(f"rank(s) {bad} reported an error before the match exchange")
        need = int(self.heads_np[:, 0].max())
        if need > self.cap:  # every rank sees the same headers: all grow alike
            kept = None if overflow_rows is not None or n == 0 else self.rows_np[:n].copy()
            cap = self.cap
            while cap < need:
                cap *= 2
            self._alloc(cap)
            self.regrown += 1
            if overflow_rows is not None:
                self.rows_np[:n] = overflow_rows  # (plain copies of int64 rows: nothing left that can raise)
            elif kept is not None:
                self.rows_np[:n] = kept
        if need > 0:
            words = need * self.cols
            self.dev[:words].copy_(self.stage[:words], non_blocking=True)
            if self.copied is not None:
                self.copied.record()
            # the rows travel to rank 0 only
            bufs = [self.all_dev[r, :words] for r in range(self.world)] if self.rank == 0 else None
            self.dist.gather(self.dev[:words], gather_list=bufs, dst=0)
        if self.rank != 0:
            return None
        out = []
        if need > 0:
            self.all_host[:, : need * self.cols].copy_(self.all_dev[:, : need * self.cols])
        for r in range(self.world):
            cnt = int(self.heads_np[r, 0])
            rows = self.all_np[r, : cnt * self.cols].reshape(cnt, self.cols)
            out.append(ShardResult(rows, int(self.heads_np[r, 1]), int(self.heads_np[r, 2])))
        return out


class GatherWorker:
    """Runs MatchGather.gather + merge_shard_results on a worker thread, in submission order (every
    rank issues its collectives in the same order), so that the exchange of search i overlaps
    search i+1.  flush() returns when everything submitted has been gathered.  After a GatherError
    -- which every rank gets for the same submission -- the rest of the queue is dropped on all ranks
    alike, so no rank is left inside a collective."""

    def __init__(self, gatherer: MatchGather):
        import queue
        import threading
        self.g = gatherer
        self.q = queue.Queue()
        self.last = None
        self.error = None
        self.busy_s = 0.0   # seconds the worker spent in gather + merge (the exchange's share of a step, per rank)
        self.gathers = 0
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
                    import time
                    local, failed = item
                    t0 = time.perf_counter()
                    shards = self.g.gather(local, error=failed)
                    if shards is not None:
                        self.last = merge_shard_results(shards).copy()
                    self.busy_s += time.perf_counter() - t0
                    self.gathers += 1
            except BaseException as e:  # surfaced by flush()
                self.error = e
            finally:
                self.q.task_done()

    def submit(self, local):
        self.q.put((local, False))

    def submit_error(self):
        """This rank's search failed: take part in the exchange with the error flag set."""
        self.q.put((None, True))

    def flush(self):
        self.q.join()
        if self.error is not None:
            raise self.error
        return self.last

    def close(self):
        self.q.put(None)
        self.t.join()



def init_or_fallback(dist, backend: str, device_id, coll_device, torch, timeout_s: float = 180.0, preflight_cmd=None):
    """Brings up the process group of a one-rank-per-GPU run so that a broken collective path cannot sink the run.

    1. every rank joins a gloo group (CPU, TCP on the launcher's MASTER_ADDR: failures here are ordinary exceptions);
    2. backend "nccl" (= RCCL): rank 0 runs `preflight_cmd` (default: tools/preflight_multigpu.py with as many ranks --
       fresh processes doing an all_reduce, the header all_gather, the row gather and one real search on the same GPUs)
       and tells the others over gloo whether it passed: an RCCL hang or abort kills the preflight's processes, not this one;
    3. the gloo group is replaced by the real one and one all_reduce + barrier runs on it.
    Returns ("ranks", None) with the real group initialised, or ("inproc", reason): the caller's rank 0 then runs the same
    workload through the in-process multi-device searcher (bench.py --mode inproc), the other ranks return.
    SASSY_BENCH_PREFLIGHT=0 skips step 2.  SASSY_BENCH_INJECT_INIT_FAILURE = "preflight" | "rank<r>" | "real": tests."""
    import datetime
    import os
    import subprocess
    import sys
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    inject = os.environ.get("SASSY_BENCH_INJECT_INIT_FAILURE", "")
    to = datetime.timedelta(seconds=timeout_s)

    def down():
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass

    try:
        if inject == f"rank{rank}":
            raise RuntimeError("injected failure on this rank before the rendezvous (SASSY_BENCH_INJECT_INIT_FAILURE)")
        dist.init_process_group(backend="gloo", timeout=to)
        verdict = [None]
        if rank == 0:
            why = None
            if inject == "preflight":
                why = "injected preflight failure (SASSY_BENCH_INJECT_INIT_FAILURE)"
            elif backend == "nccl" and os.environ.get("SASSY_BENCH_PREFLIGHT", "1") != "0":
                root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
                cmd = preflight_cmd or [sys.executable, os.path.join(root, "tools", "preflight_multigpu.py"), "--gpus", str(world),
                                        "--mbytes", "8", "--timeout", str(int(timeout_s))]
                env = {k_: v for k_, v in os.environ.items()
                       if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                     "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
                try:
                    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s + 60)
                    if r.returncode != 0 or '"ok": true' not in r.stdout:
                        why = f"RCCL preflight failed (exit {r.returncode}): {(r.stderr or r.stdout)[-300:].strip()}"
                except Exception as e:  # noqa: BLE001 -- timeout, missing file: all of it means "do not trust the collective path"
                    why = f"RCCL preflight did not finish: {type(e).__name__}: {str(e)[:200]}"
            verdict[0] = why
        dist.broadcast_object_list(verdict, src=0)
        dist.barrier()
        down()
        if verdict[0] is not None:
            return "inproc", verdict[0]
        if inject == "real":
            raise RuntimeError("injected failure of the real process group (SASSY_BENCH_INJECT_INIT_FAILURE)")
        kw = dict(backend=backend, timeout=to)
        if device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(**kw)
        t = torch.ones(1, dtype=torch.float64, device=coll_device)
        dist.all_reduce(t)
        if int(t.item()) != world:
            raise RuntimeError(f"first all_reduce over {world} ranks summed to {t.item()}")
        dist.barrier()
        return "ranks", None
    except Exception as e:  # noqa: BLE001
        down()
        return "inproc", f"{type(e).__name__}: {str(e)[:300]}"
