"""Worker of tests/test_gpu_parity.py::test_two_ranks_share_one_gpu_seam_plateau -- run under
torch.distributed.run with WORLD_SIZE ranks that share the visible GPU(s) (gloo for the match
exchange, as bench.py --allow-shared-gpu does).  Every rank builds the same text, uploads ITS shard
(+ halo) only, searches it with sassy_hip_search_shard and takes part in the one-collective match
exchange (multigpu.MatchGather on its worker thread); rank 0 merges with the seam protocol and
compares with the oracle on the whole text.  Prints one JSON line on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cases():
    import random
    rng = random.Random(7)
    out = []
    # (1) a <=k plateau that runs across the rank seam: pattern A^20 on a long A run with one C in it; the
    # run starts in rank 0's shard and ends in rank 1's (any world: the run covers the middle of the text)
    pat = b"A" * 20
    text = b"G" * 1500 + b"A" * 90 + b"C" + b"A" * 30000 + b"G" * 2500
    out.append(("dna", pat, text, 3))
    # (2) periodic text / periodic pattern (BASELINE config 1's pattern) with plants near the seam
    pat = b"ATCG" * 8
    t = bytearray(rng.choice(b"ACGT") for _ in range(40000))
    mid = len(t) // 2 // 64 * 64
    for off in (-40, -8, 0, 17, 64, 700):
        t[mid + off:mid + off + 32] = pat
    t[mid - 300:mid - 300 + 31] = pat[:10] + pat[11:]
    out.append(("dna", pat, bytes(t), 3))
    # (3) config-3 shape: 200-row Iupac pattern, k = 20 -- cigars far beyond 40 characters must travel
    pat = bytearray(rng.choice(b"ACGT") for _ in range(200))
    pat[50], pat[100], pat[150], pat[199] = ord("N"), ord("R"), ord("Y"), ord("W")
    t = bytearray(rng.choice(b"ACGT") for _ in range(60000))
    base = bytes({ord("N"): 65, ord("R"): 71, ord("Y"): 67, ord("W"): 84}.get(c, c) for c in pat)
    for pos in (5000, 29990 // 64 * 64 - 100, 30100, 52000):
        ins = bytearray(base)
        for e in range(12):
            ins[(e * 17 + 3) % len(ins)] = rng.choice(b"ACGT")
        del ins[77]
        t[pos:pos + len(ins)] = ins
    out.append(("iupac", bytes(pat), bytes(t), 20))
    return out


def main():
    import torch
    import torch.distributed as dist

    import oracle
    import sassy_amd
    from sassy_amd import multigpu

    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1, "no HIP device"
    torch.cuda.set_device(rank % n_dev)
    dist.init_process_group(backend="gloo")
    report = []
    ok = True
    try:
        for profile, pat, text, k in cases():
            n = len(text)
            bounds = multigpu.shard_bounds(n, world)
            a, b = bounds[rank]
            halo = 0 if rank == 0 else sassy_amd.required_halo(len(pat), k)
            assert a >= halo
            buf = sassy_amd.DeviceBuffer(halo + (b - a) + 256)
            buf.upload(text[a - halo:b])
            s = sassy_amd.Searcher(profile, rc=False)
            mg = multigpu.MatchGather(torch, dist, torch.device("cpu"), capacity_rows=2,
                                      cigar_bytes=multigpu.cigar_bytes_for(len(pat), k))
            w = multigpu.GatherWorker(mg)
            for _ in range(2):  # twice: the second exchange runs at the grown capacity
                w.submit(s.search_shard(pat, buf.ptr, halo, b - a, a, n, k))
            rows = w.flush()
            w.close()
            if rank == 0:
                got = multigpu.matches_from_rows(rows, sassy_amd.Match)
                want = oracle.search(profile, pat, text, k)
                key = lambda m: (m.text_start, m.text_end, m.cost, m.strand, m.cigar)
                same = [key(m) for m in got] == [key(m) for m in want]
                ok = ok and same and len(want) > 0
                report.append({"profile": profile, "m": len(pat), "k": k, "n": n, "matches": len(want), "same": same,
                               "longest_cigar": max((len(m.cigar) for m in want), default=0), "regrown": mg.regrown})
            buf.free()
    finally:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"ok": ok, "world": world, "cases": report}), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
