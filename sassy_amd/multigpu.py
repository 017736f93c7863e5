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
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

STATE_FALSE, STATE_TRUE, STATE_PASS = 0, 1, 2
_CIGAR_BYTES = 40
_COLS = 7 + _CIGAR_BYTES // 8  # int64 columns per packed match


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
    matches: list            # sassy_amd.Match (global coordinates)
    exit_state: int          # STATE_*
    conditional_index: int   # index into matches of the report that depends on the left shard, or -1


def merge_shard_results(shards: Sequence[ShardResult]) -> list:
    """Concatenate shard results in text order, dropping conditional reports whose plateau was
    entered by an increase (decreasing = FALSE arriving from the left)."""
    out = []
    incoming = STATE_TRUE  # column 0 of the text: decreasing = true (src/search.rs:1055)
    for sh in shards:
        ms = list(sh.matches)
        if sh.conditional_index >= 0 and incoming != STATE_TRUE:
            del ms[sh.conditional_index]
        out.extend(ms)
        if sh.exit_state != STATE_PASS:
            incoming = sh.exit_state
    return out


def pack_matches(matches, torch, device):
    """[count, _COLS] int64 tensor: idx, start, end, pstart, pend, cost, strand, cigar bytes."""
    import numpy as np
    arr = np.zeros((len(matches), _COLS), dtype=np.int64)
    for i, m in enumerate(matches):
        cig = m.cigar.encode()
        if len(cig) > _CIGAR_BYTES:
            raise ValueError("cigar longer than the fixed gather field")
        row = arr[i]
        row[0], row[1], row[2], row[3], row[4] = m.pattern_idx, _s64(m.text_start), _s64(m.text_end), \
            _s64(m.pattern_start), _s64(m.pattern_end)
        row[5], row[6] = m.cost, 1 if m.strand == "-" else 0
        row[7:].view(np.uint8)[: len(cig)] = np.frombuffer(cig, dtype=np.uint8)
    return torch.from_numpy(arr).to(device)


def unpack_matches(t, Match):
    import numpy as np
    arr = t.cpu().numpy()
    out = []
    for row in arr:
        cig = bytes(row[7:].view(np.uint8)).rstrip(b"\0").decode()
        out.append(Match(int(row[0]), _u64(row[1]), _u64(row[2]), _u64(row[3]), _u64(row[4]),
                         int(row[5]), "-" if row[6] else "+", cig))
    return out


def _s64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _u64(v) -> int:
    v = int(v)
    return v + (1 << 64) if v < 0 else v


def gather_shard_results(local: ShardResult, torch, dist, device, Match) -> Optional[List[ShardResult]]:
    """The one exchange of the path: all ranks' match lists to rank 0.
    Two collectives: all_gather of the 3-word headers (count, exit state, conditional index),
    then gather of the records padded to the largest count."""
    world, rank = dist.get_world_size(), dist.get_rank()
    head = torch.tensor([len(local.matches), local.exit_state, local.conditional_index],
                        dtype=torch.int64, device=device)
    heads = [torch.empty_like(head) for _ in range(world)]
    dist.all_gather(heads, head)
    counts = [int(h[0]) for h in heads]
    maxc = max(counts)
    if maxc == 0:
        if rank != 0:
            return None
        return [ShardResult([], int(h[1]), int(h[2])) for h in heads]
    mine = pack_matches(local.matches, torch, device)
    if mine.shape[0] < maxc:
        pad = torch.zeros((maxc - mine.shape[0], _COLS), dtype=torch.int64, device=device)
        mine = torch.cat([mine, pad], dim=0)
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gather_list=bufs, dst=0)
    if rank != 0:
        return None
    return [ShardResult(unpack_matches(bufs[r][: counts[r]], Match), int(heads[r][1]), int(heads[r][2]))
            for r in range(world)]
