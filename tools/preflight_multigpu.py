#!/usr/bin/env python
"""tools/preflight_multigpu.py [--gpus 2] [--mbytes 64] -- half a minute that tells whether the RCCL branch of bench.py
can run on this node at all, before a scaling run is spent on it: N ranks (one per GPU, torch.distributed.run on
127.0.0.1), backend "nccl" (= RCCL on ROCm): an all_reduce, the header all_gather and the row gather bench.py's
MatchGather uses with `mbytes` of int64 rows per rank, then ONE real search per rank on a 64 MiB shard of the synthetic
text with its matches gathered to rank 0 and merged.  Rank 0 prints one JSON line {"ok": true, ...}; any failure is a
non-zero exit code with the exception on stderr.  (bench.py --mode inproc is the scaling run that needs none of this.)"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, timeout=args.timeout)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--mbytes", type=int, default=64)
    ap.add_argument("--timeout", type=int, default=300)
    ap.add_argument("--backend", default="nccl", help="'gloo' checks the script itself on a box without N GPUs (ranks share device 0)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch(args))
    import torch
    import torch.distributed as dist
    import sassy_amd
    from sassy_amd import multigpu
    from bench import _dna_bytes
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks but {n_dev} HIP device(s): one rank per GPU")
    dev = torch.device("cuda", local % max(1, n_dev))
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
        cdev = dev
    else:
        dist.init_process_group("gloo")
        cdev = torch.device("cpu")
    t_init = time.perf_counter() - t0
    x = torch.full((1024,), float(rank + 1), device=cdev)
    dist.all_reduce(x)
    assert float(x[0]) == world * (world + 1) / 2, float(x[0])
    rows = args.mbytes * (1 << 20) // 8
    mine = torch.arange(rows, dtype=torch.int64, device=cdev) + rank
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    if cdev.type == "cuda":
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    dist.gather(mine, gather_list=bufs, dst=0)
    if cdev.type == "cuda":
        torch.cuda.synchronize()
    t_gather = time.perf_counter() - t1
    if rank == 0:
        for r in range(world):
            assert int(bufs[r][5]) == 5 + r
    # one real search per rank, gathered and merged like bench.py's steps
    n_per = 64 << 20
    m, k = 32, 3
    pat = bytes(_dna_bytes(43, 0, m))
    halo = 0 if rank == 0 else sassy_amd.required_halo(m, k)
    a = rank * n_per
    buf = torch.empty(halo + n_per + 4096, dtype=torch.uint8, device=dev)
    sassy_amd.generate_dna(buf.data_ptr(), halo + n_per, 42, a - halo)
    sassy_amd.plant(buf.data_ptr(), halo + n_per, a - halo, n_per * world, 42, pat, k, 1 << 20)
    torch.cuda.synchronize()
    s = sassy_amd.Searcher("dna", rc=False)
    r = s.search_shard(pat, buf.data_ptr(), halo, n_per, a, n_per * world, k)
    mg = multigpu.MatchGather(torch, dist, cdev, capacity_rows=64, cigar_bytes=multigpu.cigar_bytes_for(m, k))
    shards = mg.gather(r)
    merged = None
    if rank == 0:
        merged = multigpu.merge_shard_results(shards)
        assert len(merged) >= 60 * world - 2, len(merged)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"ok": True, "backend": dist.get_backend(), "world": world, "init_s": round(t_init, 2),
                          "gather_mbytes_per_rank": args.mbytes, "gather_s": round(t_gather, 4),
                          "gather_GB_per_s_into_rank0": round(args.mbytes * (world - 1) / 1024 / max(t_gather, 1e-9), 2),
                          "matches_merged": int(len(merged)), "seconds": round(time.perf_counter() - t0, 2)}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
