"""CPU-only tests of the N > 1 path: shard bounds, the cross-shard plateau chain, and the gather
of match lists with torch.distributed (gloo, world_size 2) -- the same code that runs over RCCL."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sassy_amd import multigpu  # noqa: E402
from sassy_amd import Match  # noqa: E402


def M(ts, te, cost=0, strand="+", cigar="4="):
    return Match(0, ts, te, 0, te - ts, cost, strand, cigar)


def test_shard_bounds():
    for total, world in [(3_000_000_000 * 8, 8), (1000, 3), (64, 2), (65, 2), (0, 2), (24_000_000_001, 8)]:
        b = multigpu.shard_bounds(total, world)
        assert len(b) == world
        assert b[0][0] == 0 and b[-1][1] == total
        for (a0, b0), (a1, b1) in zip(b[:-1], b[1:]):
            assert b0 == a1 and a0 % 64 == 0 and (b0 % 64 == 0 or b0 == total)


def test_merge_chain():
    S = multigpu.ShardResult
    T, F, P = multigpu.STATE_TRUE, multigpu.STATE_FALSE, multigpu.STATE_PASS
    # no conditionals: plain concatenation
    assert multigpu.merge_shard_results([S([M(1, 5)], T, -1), S([M(70, 74)], T, -1)]) == [M(1, 5), M(70, 74)]
    # conditional report survives when the chain delivers TRUE through PASS shards
    sh = [S([M(1, 5)], T, -1), S([], P, -1), S([M(200, 204), M(300, 304)], F, 0), S([M(400, 404)], T, 0)]
    assert multigpu.merge_shard_results(sh) == [M(1, 5), M(200, 204), M(300, 304)]
    # ... and is dropped when the plateau was entered by an increase
    sh = [S([M(1, 5)], F, -1), S([], P, -1), S([M(200, 204), M(300, 304)], T, 0)]
    assert multigpu.merge_shard_results(sh) == [M(1, 5), M(300, 304)]
    # first shard: column 0 counts as decreasing
    assert multigpu.merge_shard_results([S([M(1, 5)], P, 0)]) == [M(1, 5)]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        big = (1 << 64) - 1
        if rank == 0:
            local = multigpu.ShardResult([M(3, 7, 1, "+", "3=1X"), M(90, 95, 2, "-", "2=1I2=1D")], multigpu.STATE_FALSE, -1)
        else:
            local = multigpu.ShardResult([M(3_000_000_100, 3_000_000_132, 3, "+", "10=1X10=1X10="),
                                          Match(0, big, 3_000_000_500, big, 32, 0, "+", ""),
                                          M(3_000_000_900, 3_000_000_932)], multigpu.STATE_TRUE, 0)
        got = multigpu.gather_shard_results(local, torch, dist, dev, Match)
        if rank == 0:
            assert len(got) == 2
            assert got[0].matches == local.matches and got[0].exit_state == multigpu.STATE_FALSE
            assert got[1].conditional_index == 0 and len(got[1].matches) == 3
            assert got[1].matches[1].text_start == big and got[1].matches[0].cigar == "10=1X10=1X10="
            merged = multigpu.merge_shard_results(got)
            assert [m.text_start for m in merged] == [3, 90, big, 3_000_000_900]
        else:
            assert got is None
        # empty everywhere
        got = multigpu.gather_shard_results(multigpu.ShardResult([], multigpu.STATE_PASS, -1), torch, dist, dev, Match)
        if rank == 0:
            assert [len(g.matches) for g in got] == [0, 0]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_gather_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
