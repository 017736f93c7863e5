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
    def S(ms, state, cond):
        return multigpu.ShardResult(multigpu.rows_from_matches(ms), state, cond)

    def merged(shards):
        return multigpu.matches_from_rows(multigpu.merge_shard_results(shards), Match)

    T, F, P = multigpu.STATE_TRUE, multigpu.STATE_FALSE, multigpu.STATE_PASS
    # no conditionals: plain concatenation
    assert merged([S([M(1, 5)], T, -1), S([M(70, 74)], T, -1)]) == [M(1, 5), M(70, 74)]
    # conditional report survives when the chain delivers TRUE through PASS shards
    sh = [S([M(1, 5)], T, -1), S([], P, -1), S([M(200, 204), M(300, 304)], F, 0), S([M(400, 404)], T, 0)]
    assert merged(sh) == [M(1, 5), M(200, 204), M(300, 304)]
    # ... and is dropped when the plateau was entered by an increase
    sh = [S([M(1, 5)], F, -1), S([], P, -1), S([M(200, 204), M(300, 304)], T, 0)]
    assert merged(sh) == [M(1, 5), M(300, 304)]
    # first shard: column 0 counts as decreasing
    assert merged([S([M(1, 5)], P, 0)]) == [M(1, 5)]


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
            mine = [M(3, 7, 1, "+", "3=1X"), M(90, 95, 2, "-", "2=1I2=1D")]
            local = multigpu.ShardResult(multigpu.rows_from_matches(mine), multigpu.STATE_FALSE, -1)
        else:
            mine = [M(3_000_000_100, 3_000_000_132, 3, "+", "10=1X10=1X10="),
                    Match(0, big, 3_000_000_500, big, 32, 0, "+", ""),
                    M(3_000_000_900, 3_000_000_932)]
            local = multigpu.ShardResult(multigpu.rows_from_matches(mine), multigpu.STATE_TRUE, 0)
        got = multigpu.gather_shard_results(local, torch, dist, dev)
        if rank == 0:
            assert len(got) == 2
            ms0 = multigpu.matches_from_rows(got[0].rows, Match)
            ms1 = multigpu.matches_from_rows(got[1].rows, Match)
            assert ms0 == mine and got[0].exit_state == multigpu.STATE_FALSE
            assert got[1].conditional_index == 0 and len(ms1) == 3
            assert ms1[1].text_start == big and ms1[0].cigar == "10=1X10=1X10="
            merged = multigpu.matches_from_rows(multigpu.merge_shard_results(got), Match)
            assert [m.text_start for m in merged] == [3, 90, big, 3_000_000_900]
        else:
            assert got is None
        # empty everywhere
        empty = multigpu.ShardResult(multigpu.rows_from_matches([]), multigpu.STATE_PASS, -1)
        got = multigpu.gather_shard_results(empty, torch, dist, dev)
        if rank == 0:
            assert [len(g) for g in got] == [0, 0]
        # the one-collective variant bench.py uses, directly and on its worker thread; capacity 2 is
        # too small for rank 1's three rows: every rank grows and repeats the exchange once
        cb = multigpu.cigar_bytes_for(32, 3)
        mg = multigpu.MatchGather(torch, dist, dev, capacity_rows=2, cigar_bytes=cb)
        for it in range(3):
            got = mg.gather(local)
            assert mg.cap == 4 and mg.regrown == 1
            if rank == 0:
                assert [len(g) for g in got] == [2, 3] and got[1].conditional_index == 0
                merged = multigpu.matches_from_rows(multigpu.merge_shard_results(got), Match)
                assert [m.text_start for m in merged] == [3, 90, big, 3_000_000_900]
                assert merged[0].cigar == "3=1X" and merged[3].cigar == "4="
            else:
                assert got is None
        w = multigpu.GatherWorker(mg)
        for i in range(5):
            w.submit(local if i % 2 == 0 else empty)
        last = w.flush()
        if rank == 0:
            assert last is not None and [int(r[1]) for r in last] == [3, 90, -1, 3_000_000_900]
        # config-3 shapes: cigars of a 200-row pattern with k = 20 (up to 442 characters) travel too
        long_cigar = "".join(f"{3 + (i % 5)}={1}X" for i in range(40))[:400]
        cb3 = multigpu.cigar_bytes_for(200, 20)
        assert cb3 >= len(long_cigar)
        mg3 = multigpu.MatchGather(torch, dist, dev, capacity_rows=4, cigar_bytes=cb3)
        mine3 = [M(1000 * rank + 5, 1000 * rank + 210, 7, "+", long_cigar)]
        got = mg3.gather(multigpu.ShardResult(multigpu.rows_from_matches(mine3), multigpu.STATE_TRUE, -1))
        if rank == 0:
            merged = multigpu.matches_from_rows(multigpu.merge_shard_results(got), Match)
            assert [m.cigar for m in merged] == [long_cigar, long_cigar] and merged[1].text_start == 1005
        # a rank that fails before the exchange takes every rank down with it, together: rank 1 flags
        # an error for the 2nd submission, both ranks raise GatherError from flush(), nobody hangs
        w2 = multigpu.GatherWorker(mg)
        w2.submit(local)
        if rank == 1:
            w2.submit_error()
        else:
            w2.submit(local)
        w2.submit(local)  # dropped on both ranks
        try:
            w2.flush()
            raise AssertionError("no GatherError")
        except multigpu.GatherError as e:
            assert "[1]" in str(e)
        w2.close()
        w.close()
        # a rank whose rows cannot be packed (here: a cigar wider than the gather's field; rank 0 only) must not leave
        # the others inside the row gather: packing happens in front of the header exchange, the flag travels in the
        # header, both ranks raise together -- with rows that fit the staging buffer and with rows that overflow it
        for cap in (8, 1):
            mg4 = multigpu.MatchGather(torch, dist, dev, capacity_rows=cap, cigar_bytes=8)
            wide = [M(10, 60, 5, "+", "10=1X10=1X10=1X10="), M(100, 104), M(200, 204)]
            narrow = [M(20, 24), M(30, 34)]
            rows = multigpu.rows_from_matches(wide if rank == 0 else narrow)
            try:
                mg4.gather(multigpu.ShardResult(rows, multigpu.STATE_TRUE, -1))
                raise AssertionError("no GatherError")
            except multigpu.GatherError as e:
                assert "[0]" in str(e)
            # the gatherer is still usable: the next exchange is in step on both ranks
            got = mg4.gather(multigpu.ShardResult(multigpu.rows_from_matches(narrow), multigpu.STATE_TRUE, -1))
            if rank == 0:
                assert [len(g) for g in got] == [2, 2]
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


def test_multi_layout_covers_every_share_of_the_reversed_text():
    """sassy_hip_multi_layout (no device needed): for every text length -- tiny texts above all: fewer blocks than parts,
    a trailing part without bytes -- and 1 .. 8 parts, each part's resident bytes [offset - halo, offset + len + behind)
    cover the forward bytes of its share of the REVERSED text with that share's halo (the Rc strand of a both-strand
    multi-searcher, multi_rc_prepare); texts shorter than 64 n (n + 2) bytes are one part's."""
    import sassy_amd
    lens = list(range(0, 700)) + [1023, 1024, 1025, 4095, 4096, 5119, 5120, 5121, 65536 + 7, 10**6 + 3, 3 * 10**9]
    for parts in range(1, 9):
        for n in lens:
            for (m, k) in ((32, 3), (200, 20)):
                e, rows = sassy_amd.multi_layout(n, parts, m, k)
                assert e >= 1, (n, parts, m, k, rows)
                assert e == (1 if parts == 1 or n < 64 * parts * (parts + 2) else parts)
                halo_need = sassy_amd.required_halo(m, k)
                assert sum(r[1] for r in rows) == n and all(r[1] == 0 for r in rows[e:])
                at = 0
                for i, (off, ln, halo, behind, fa, fb, hrev) in enumerate(rows):
                    assert off == at or ln == 0
                    at += ln
                    assert halo % 64 == 0 and halo <= off and off + ln + behind <= n
                    if i and ln:
                        assert halo >= min(halo_need, off) // 64 * 64
                    if i < e and fb > fa:
                        assert off - halo <= fa // 16 * 16 and fb + hrev <= off + ln + behind, (n, parts, i, rows[i])
                # the reversed shares tile the text as well
                assert sum(r[5] - r[4] for r in rows[:e]) == n


def _fallback_worker(rank, world, port, inject, q):
    import torch
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    if inject:
        os.environ["SASSY_BENCH_INJECT_INIT_FAILURE"] = inject
    try:
        verdict, why = multigpu.init_or_fallback(dist, "gloo", None, torch.device("cpu"), torch, timeout_s=8.0)
        extra = None
        if verdict == "ranks":  # the group it left behind works
            t = torch.full((4,), float(rank + 1))
            dist.all_reduce(t)
            extra = float(t[0])
            dist.destroy_process_group()
        q.put((rank, verdict, bool(why), extra, dist.is_initialized()))
    except Exception as e:  # pragma: no cover
        q.put((rank, "raised", repr(e), None, None))


@pytest.mark.parametrize("inject", ["", "preflight", "real", "rank1"])
def test_rank_path_falls_back_to_inproc_gloo_world2(inject):
    """bench.py's rank path must not be able to fail on first contact (multigpu.init_or_fallback, gloo world 2): a sound
    collective path gives ("ranks", None) and a working group; a failed preflight, a failing real group, a rank that dies
    before the rendezvous all give ("inproc", reason) on every surviving rank -- rank 0 then runs --mode inproc -- with
    no process group left behind and nobody hanging."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fallback_worker, args=(r, 2, port, inject, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    if inject == "":
        assert res == [(0, "ranks", False, 3.0, False), (1, "ranks", False, 3.0, False)], res
    else:
        assert [r[:3] for r in res] == [(0, "inproc", True), (1, "inproc", True)], res
        assert all(r[4] is False for r in res), res
