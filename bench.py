#!/usr/bin/env python
"""bench.py -- the headline measurement of the search path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the one the metric is quoted on; configs[4] for N > 1):
Searcher::<Dna>::new_fwd().search(pattern, text, k) with |pattern| = 32 (seeded random 32-mer),
k = 3, on 3 GB of synthetic random-ACGT text PER GPU that is already resident in HBM when the
timed region starts (generated on the device; one planted near-match per MiB so that
"matches/sec" means something).  A "step" is one full search of the resident shard: scan kernel,
report resolution, traceback, Match records (with cigars) on the host -- and, for N > 1, the
RCCL gather of all ranks' match lists to rank 0 and the cross-shard merge.  Weak scaling: the
per-GPU text is fixed, total text = N x 3 GB.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel (scan_kernel):
algorithmic bytes per launch = text bytes scanned (every text byte is read from HBM exactly once,
SURVEY 8d) divided by the kernel's duration measured with HIP events on the searcher's stream
inside the timed steps.  `cpu_baseline` (N = 1 only) times the reference-shaped CPU port
(oracle/sassy_refstyle.c, kind "port": the reference is Rust and cannot be built here) on the
host cores of the same box over the same text.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PATTERN_SEEDS = (43, 46, 47, 48)  # the rotation of the searches in flight (one more than the default three in flight)
METRIC = "GB text/sec (+ matches/sec) |P|=32 k=3 DNA, 1/2/4/8 MI355X vs CPU"
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(host_text, pat, k, profile, gpu_ends, min_seconds):
    """Reference-shaped CPU port on all host cores over the same text (see module docstring): the
    threads live inside liboracle.so (oracle/sassy_refstyle.c: rs_scan_mt -- persistent pthreads, one
    shard each, the clock runs between two barriers so thread creation is not timed, passes repeat
    until min_seconds have gone by)."""
    import oracle

    n = host_text.size
    cores = os.cpu_count() or 1
    usable, why = usable_cpus()
    oracle.lib()  # build / load outside the timed region
    ends, info = oracle.refstyle_ends_mt(profile, pat, host_text, k, usable, min_seconds)
    T = info["shards"]
    gbps = n * info["passes"] / info["seconds"] / 1e9
    # single thread, one search call, on a 2^28-byte slice (comparable to the reference's
    # published 1.2-2.1 GB/s per thread)
    sl = host_text[: min(n, 1 << 28)]
    t1 = time.perf_counter()
    oracle.refstyle_ends(profile, pat, sl, k)
    st = time.perf_counter() - t1
    single = sl.size / st / 1e9
    return {
        "value": round(gbps, 3),
        "unit": "GB/s",
        "cores": T,
        "kind": "port",
        "sample": f"{info['passes']} passes ({info['seconds']:.2f} s wall) over the full {n} byte text of this run, "
                  f"{T} persistent pthreads (= the CPUs this container may use: {why}; the host has {cores} hardware threads) "
                  f"inside liboracle.so, one shard each with m+k+1 bytes of overlap (whole "
                  f"blocks), clock between two barriers (oracle/sassy_refstyle.c rs_scan_mt, gcc -O3 -mavx2 -mbmi2, "
                  f"4x u64 lanes)",
        "single_thread_gbps": round(single, 3),
        "per_thread_gbps": round(gbps / T, 4),
        "parallel_efficiency": round(gbps / T / single, 3) if single > 0 else None,
        "cpu_seconds": round(info["busy_seconds"], 2),
        "host_cpus": cores,
        "usable_cpus": usable,
        "usable_cpus_source": why,
        "ends_equal_gpu": ends == gpu_ends,
    }


def usable_cpus():
    """The CPUs this process may really use: the machine's count, cut down by the affinity mask and by the
    container's cgroup CPU quota (a 256-thread host that grants 16 CPUs runs 256 busy threads at 1/16 speed
    each: the baseline would time the scheduler, not the scan)."""
    n = os.cpu_count() or 1
    why = "os.cpu_count()"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, why = aff, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p_
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, why = max(1, int(quota + 0.5)), f"cgroup cpu quota ({quota:g} CPUs)"
    return n, why


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here (one
    process per GPU under torch.distributed.run, rendezvous on 127.0.0.1) and hand their exit code
    back.  Under a launcher (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a quarter of a second of searches -- the first few dozen calls after start-up run ~2 % slower
    # than the steady state (clocks, first touches)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--text-bytes", type=int, default=3_000_000_000, help="text bytes per GPU")
    ap.add_argument("--pattern-len", type=int, default=32)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--profile", default="dna")
    ap.add_argument("--plant-stride", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune-searches", type=int, default=0,
                    help="extra untimed searches before the warm-up (only of use with SASSY_HIP_TUNE=1: the opt-in "
                         "geometry tuner settles within 36 searches); the default run has none")
    ap.add_argument("--cpu-seconds", type=float, default=2.0,
                    help="wall seconds the CPU baseline's threads keep scanning (at least one pass)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="searches in flight on the device (1 = every search alone: begin, wait, next; 3 measured best: "
                         "profiles/r02_in_flight.txt)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of BASELINE configs 3 and 4 on the resident text (N = 1, after the timed steps)")
    ap.add_argument("--mode", choices=["ranks", "inproc"], default="ranks",
                    help="N > 1: 'ranks' (default) = one process per GPU, torch.distributed over RCCL, match rows gathered to "
                         "rank 0; 'inproc' = ONE process driving all N devices through the C-ABI's multi-device searcher "
                         "(sassy_hip_multi_*: a host thread and a resident shard per device, merge in C) -- no "
                         "torch.distributed, no RCCL: a scaling run that does not depend on the collective path")
    ap.add_argument("--settle", type=int, default=100,
                    help="untimed searches in front of the timed region on top of --warmup (clocks, first touches, the "
                         "pipeline of searches in flight): the first dozens of steps after start-up run a few percent slow")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="debugging the N > 1 path on a box with fewer GPUs than ranks: ranks share devices and "
                         "the match exchange goes over gloo; never a valid scaling measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.mode == "inproc":
        return main_inproc(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch

    import sassy_amd
    from sassy_amd import multigpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the number of ranks must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False)")
    n_dev = torch.cuda.device_count()
    # one rank per GPU over RCCL; --allow-shared-gpu (debugging the N > 1 path on a box with fewer GPUs
    # than ranks): the ranks share devices and the match exchange goes over gloo with host tensors
    shared_gpu = world > n_dev
    if shared_gpu and not args.allow_shared_gpu:
        raise SystemExit(f"--gpus {args.gpus} but only {n_dev} HIP device(s) visible (one rank per GPU)")
    torch.cuda.set_device(local_rank % n_dev)
    device = torch.device("cuda", local_rank % n_dev)
    coll_device = torch.device("cpu") if shared_gpu else device
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The collective path must not be able to sink a scaling run on first contact: the process group is brought up and
        # one all_reduce + barrier is run under a timeout (multigpu.init_or_fallback); if any of it fails on any rank, rank 0
        # runs the SAME workload through the in-process multi-device searcher (--mode inproc: a host thread per GPU, no
        # torch.distributed) and says so in config.parallelism; the other ranks leave quietly.
        verdict, why = multigpu.init_or_fallback(dist, "gloo" if shared_gpu else "nccl", None if shared_gpu else device,
                                                 coll_device, torch, timeout_s=float(os.environ.get("SASSY_BENCH_INIT_TIMEOUT", "180")))
        if verdict != "ranks":
            if rank != 0:
                return
            args.fallback_reason = why
            args.allow_shared_gpu = args.allow_shared_gpu or shared_gpu
            return main_inproc(args)

    # ---------------------------------------------------------------- workload
    n_per = args.text_bytes // 64 * 64
    total = n_per * world
    m, k = args.pattern_len, args.k
    seed_text = 42
    # The patterns: seeded random ACGT m-mers (a tiny host-side use of the same counter-based generator,
    # computed with numpy to keep the oracle out of the product path).  The stream of searches ROTATES through
    # PATTERN_SEEDS -- one pattern more than searches are in flight, so that no lane ever meets the pattern it searched
    # last (its row table and pattern are uploaded for every search, as for a stream of unrelated queries: the
    # reference's workers take whatever task comes next, bin/grep.rs:516-537) --, each planted once per
    # `plant_stride` bytes at its own phase of the stride.  pats[0] (seed 43) is the pattern of the lone-search and
    # roofline measurements, of the CPU baseline and of the earlier rounds' lines.
    pats = []
    for sd in PATTERN_SEEDS:
        pt = bytes(_dna_bytes(sd, 0, m))
        if args.profile == "iupac" and m >= 200:
            p = bytearray(pt)
            p[50], p[100], p[150], p[199] = ord("N"), ord("R"), ord("Y"), ord("W")
            pt = bytes(p)
        pats.append(pt)
    pat = pats[0]
    halo = 0 if rank == 0 else sassy_amd.required_halo(m, k)
    a = rank * n_per
    buf = torch.empty(halo + n_per + 4096, dtype=torch.uint8, device=device)
    sassy_amd.generate_dna(buf.data_ptr(), halo + n_per, seed_text, a - halo)
    planted = 0
    for j, pt in enumerate(pats):
        cnt = sassy_amd.plant(buf.data_ptr(), halo + n_per, a - halo, total, seed_text + j,
                              bytes({ord("Y"): 67}.get(c, c if c in b"ACGT" else 65) for c in pt), k, args.plant_stride,
                              phase=(args.plant_stride // (len(pats) + 1)) * j // 64 * 64)  # (clear of other_configs' plants a quarter stride on)
        planted = cnt if j == 0 else planted
    torch.cuda.synchronize()
    searcher = sassy_amd.Searcher(args.profile, rc=False)
    searcher.set_pipe_depth(max(1, min(4, args.in_flight)))

    # N > 1: the match lists go to rank 0 in one fixed-size collective per search (header + rows; the
    # capacity starts at 1024 rows and every rank doubles it alike when a header shows that some list did
    # not fit -- nothing here knows how many matches the text holds), issued by a worker thread so that the
    # exchange of search i overlaps search i+1; every timed step's exchange is complete before the closing
    # barrier (sync() drains the worker)
    gather_worker = None
    if world > 1:
        gather_worker = multigpu.GatherWorker(multigpu.MatchGather(
            torch, dist, coll_device, capacity_rows=1024, cigar_bytes=multigpu.cigar_bytes_for(m, k)))

    # A stream of searches over the resident text keeps `--in-flight` (default 3) of them in flight on the
    # device (include/sassy_hip.h: sassy_hip_search_shard_begin / sassy_hip_search_finish): step i queues
    # search i and then waits for search i-1, whose chunk DP / traceback tail ran underneath search i's
    # prefilter.  Every step is one complete search with its Match records on the host; all K searches of
    # the timed region are begun AND finished inside it (drain() before the closing barrier).
    pending = []
    step_no = [0]

    def finish_oldest():
        r = searcher.search_finish(pending.pop(0))
        st_ = searcher.stats()
        if world > 1:
            gather_worker.submit(r)
            return gather_worker.last, st_
        return r, st_  # the records are already on the host in their final form (r.array + r.pool)

    last = [None, None]

    def step():
        # one full search of the resident shard; Match records arrive on the host as one packed
        # array (include/sassy_hip.h: sassy_hip_Match + cigar pool), for N > 1 gathered to rank 0
        try:
            pt = pats[step_no[0] % len(pats)]
            step_no[0] += 1
            if args.in_flight <= 1:
                r = searcher.search_shard(pt, buf.data_ptr(), halo, n_per, a, total, k)
                st_ = searcher.stats()
                if world > 1:
                    gather_worker.submit(r)
                    r = gather_worker.last
                last[0], last[1] = r, st_
                return r, st_
            pending.append(searcher.search_shard_begin(pt, buf.data_ptr(), halo, n_per, a, total, k))
            if len(pending) >= args.in_flight:
                last[0], last[1] = finish_oldest()
            return last[0], last[1]
        except Exception:
            if world > 1:
                gather_worker.submit_error()  # the other ranks must not wait for this one's rows forever
            raise

    def drain():
        while pending:
            last[0], last[1] = finish_oldest()

    def sync():
        drain()
        if gather_worker is not None:
            gather_worker.flush()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the very first search of this process on this text: buffers are allocated, the pattern tables uploaded,
    # kernels loaded -- reported on its own, never part of the timed region
    t_cold = time.perf_counter()
    searcher.search_shard(pat, buf.data_ptr(), halo, n_per, a, total, k)
    cold_ms = (time.perf_counter() - t_cold) * 1e3
    for _ in range(args.tune_searches):
        step()
    sync()
    # One search at a time (nothing else in flight): its latency, and the dominant kernel's HIP-event duration for
    # the roofline object (with searches in flight two launches of that kernel overlap and share the HBM).  Runs
    # BEFORE the warm-up steps: 50 searches, reported as single_search_latency_ms, not part of the timed region.
    lat_t0 = time.perf_counter()
    n_lat = 50
    alone_scan_ms = alone_filter_ms = 0.0
    for _ in range(n_lat):
        searcher.search_shard(pat, buf.data_ptr(), halo, n_per, a, total, k)
        st1 = searcher.stats()
        alone_scan_ms += st1["scan_ms"] / n_lat
        alone_filter_ms += st1["filter_ms"] / n_lat
    torch.cuda.synchronize()
    latency_events_ms = (time.perf_counter() - lat_t0) / n_lat * 1e3
    # the same lone search without the two HIP events around the dominant kernel (each costs a few microseconds of
    # stream idle time): what a caller of sassy_hip_search_shard sees when nobody is timing kernels
    searcher.set_timing(0)
    for _ in range(5):
        searcher.search_shard(pat, buf.data_ptr(), halo, n_per, a, total, k)
    lat_t0 = time.perf_counter()
    for _ in range(n_lat):
        searcher.search_shard(pat, buf.data_ptr(), halo, n_per, a, total, k)
    latency_ms = (time.perf_counter() - lat_t0) / n_lat * 1e3
    fused_launch = bool(searcher.stats().get("fused", 0))
    searcher.set_timing(1)

    for _ in range(args.warmup + max(0, args.settle)):
        step()
    sync()
    gather_busy0 = gather_worker.busy_s if gather_worker is not None else 0.0
    t0 = time.perf_counter()
    scan_ms, trace_ms, filter_ms, call_ms, matches, st = 0.0, 0.0, 0.0, 0.0, None, None
    host = [0.0, 0.0, 0.0]
    for _ in range(args.steps):
        matches, st = step()
        if st is None:  # the first steps of a pipeline only queue work
            continue
        host[0] += st["host_enqueue_ms"]
        host[1] += st["host_wait_ms"]
        host[2] += st["host_post_ms"]
        scan_ms += st["scan_ms"]
        trace_ms += st["trace_ms"]
        filter_ms += st["filter_ms"]
        call_ms += st["total_ms"]
    # (this rank's own clock: its searches begun and finished, its exchanges done -- before the closing barrier)
    drain()
    if gather_worker is not None:
        gather_worker.flush()
    own_elapsed = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    gather_busy = (gather_worker.busy_s - gather_busy0) if gather_worker is not None else 0.0
    matches, st = last[0], last[1]
    if gather_worker is not None:
        matches = gather_worker.last  # rank 0: the merged rows of the last search; None elsewhere
    el = torch.tensor([elapsed, scan_ms / max(1, args.steps), filter_ms / max(1, args.steps)],
                      dtype=torch.float64, device=coll_device)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed, scan_avg_ms, filter_avg_ms = float(el[0]), float(el[1]), float(el[2])
    # per rank: ms per step by the rank's own clock and the exchange worker's busy time per step (N > 1: what a first
    # SCALE line needs beside BENCH N = 1 -- is a rank slow, or is it the gather?)
    per_rank = torch.tensor([own_elapsed / args.steps * 1e3, gather_busy / args.steps * 1e3], dtype=torch.float64, device=coll_device)
    if dist is not None:
        allr = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
    else:
        allr = [per_rank]
    per_rank_ms = [round(float(x[0]), 4) for x in allr]
    per_rank_gather_ms = [round(float(x[1]), 4) for x in allr]
    # every pattern of the rotation once more, alone: its matches on this rank's shard
    pattern_matches = [len(searcher.search_shard(pt, buf.data_ptr(), halo, n_per, a, total, k)) for pt in pats]
    matches0 = searcher.search_shard(pat, buf.data_ptr(), halo, n_per, a, total, k)
    filtered = bool(st["filtered"])
    # the dominant kernel: the prefilter scan when the pattern splits into selective pieces
    # (every text byte is read once by it), else the streaming DP kernel.  Its HIP events are the
    # only ones recorded inside the timed region (timing level 1).
    # With searches in flight two launches of the dominant kernel overlap and share the HBM: its duration in the
    # timed region (dom_inflight_ms) says how long a launch was resident, not how fast the kernel streams.  The
    # roofline object therefore uses the HIP-event duration of the same kernel in the one-search-at-a-time loop
    # of this same run (n_lat launches right behind the timed region); both are reported.
    dom_inflight_ms = filter_avg_ms if filtered else scan_avg_ms
    dom_ms = (alone_filter_ms if filtered else alone_scan_ms) if args.in_flight > 1 else dom_inflight_ms
    dom_name = {0: "scan_kernel", 1: "filter_kernel", 2: "filter_dna_kernel", 3: "filter_table_kernel", 4: "filter_count_kernel"}[int(st["filtered"])]
    # phase breakdown from a few extra, untimed steps with every phase timed (more events = more
    # stream idle time, so these are not part of the measurement above)
    searcher.set_timing(2)
    phase = {"scan_path_ms": 0.0, "trace_ms": 0.0}
    n_extra = min(5, max(1, args.steps))
    for _ in range(n_extra):
        searcher.search_shard(pat, buf.data_ptr(), halo, n_per, a, total, k)
        st2 = searcher.stats()
        phase["scan_path_ms"] += st2["scan_ms"] / n_extra
        phase["trace_ms"] += st2["trace_ms"] / n_extra
    searcher.set_timing(1)

    if rank != 0:
        if gather_worker is not None:
            gather_worker.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total * args.steps / elapsed / 1e9
    achieved = n_per / (dom_ms / 1e3) / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            # measured by tools/prof.sh (rocprofv3 PMC, separate passes) for this same workload;
            # only valid for the text size it was taken at
            t = json.load(open(tpath))
            if t.get("text_bytes_per_gpu") == n_per:
                traffic = t.get("kernels", {}).get(dom_name, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this workload, separate passes (tools/prof.sh); not measured in this run"
        except Exception:
            traffic = None
    out = {
        "metric": METRIC,
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "per_rank_ms_per_step": per_rank_ms,
        "per_rank_gather_ms_per_step": per_rank_gather_ms,
        "gather_share_of_step": round(max(per_rank_gather_ms) / ms_per_step, 4) if ms_per_step > 0 else 0.0,
        "single_search_latency_ms": round(latency_ms, 4),
        "single_search_latency_with_kernel_events_ms": round(latency_events_ms, 4),
        "single_search_roofline_frac": round(n_per / (latency_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
        "fused_filter_launch": fused_launch,
        "cold_first_search_ms": round(cold_ms, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config {'2' if world == 1 else '5'}: Searcher::<{args.profile.capitalize()}>::new_fwd()"
                         f".search, |pattern|={m}, k={k}, {n_per} B random-ACGT text per GPU resident in HBM; the stream "
                         f"of searches rotates through {len(pats)} DIFFERENT seeded random patterns (seeds {list(PATTERN_SEEDS)}: "
                         f"never the pattern a lane searched last), each planted once per {args.plant_stride} B; step = scan + "
                         f"traceback + Match records on host, {max(1, args.in_flight)} search(es) in flight" + (" + RCCL gather to rank 0 (one collective per search, overlapped with the next search)" if world > 1 else "")),
            "text_bytes_per_gpu": n_per,
            "total_text_bytes": total,
            "pattern_len": m,
            "k": k,
            "profile": args.profile,
            "searches_in_flight": max(1, args.in_flight),
            "parallelism": f"text sharded x{world}, one process per GPU"
                           + ("" if world == 1 else f", {dist.get_backend()} ({'RCCL' if dist.get_backend() == 'nccl' else 'debug: shared GPU'}) world {dist.get_world_size()}"),
            "setup": (f"one cold search (cold_first_search_ms), {args.tune_searches} tuning searches, 50 one-at-a-time searches "
                      f"(single_search_latency_ms and the roofline object's kernel time), then the W warm-up steps + {max(0, args.settle)} "
                      f"more untimed steps (--settle) and the K timed steps"),
        },
        "matches": len(matches),
        "matches_per_pattern_rank0": dict(zip([str(sd) for sd in PATTERN_SEEDS], pattern_matches)),
        "matches_per_s": round(sum(pattern_matches) / len(pattern_matches) * (world if world > 1 else 1) * args.steps / elapsed, 1),
        "planted_rank0": planted,
        "dominant_kernel_ms": round(dom_ms, 4),
        "dominant_kernel_ms_in_flight": round(dom_inflight_ms, 4),
        "phases_untimed_ms": {"scan_path": round(phase["scan_path_ms"], 4), "rank_and_trace": round(phase["trace_ms"], 4)},
        "prefilter": {"enabled": filtered, "piece_len": st["piece_len"], "hit_blocks": st["hit_blocks"],
                      "chunks": st["chunks"]},
        "host_ms_per_step": {"enqueue": round(host[0] / args.steps, 4), "wait": round(host[1] / args.steps, 4),
                             "post": round(host[2] / args.steps, 4)},
        "c_abi_call_ms_per_step": round(call_ms / args.steps, 4),
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": dom_name,
            "algorithmic_bytes_per_launch": n_per,
            "launch_ms": round(dom_ms, 4),
            "measured": (f"HIP events carried by the kernel's dispatch (hipExtLaunchKernelGGL start / stop events on the searcher's "
                         f"stream: the launch's own begin and end), {n_lat} launches of the one-search-at-a-time loop of this run"
                         if args.in_flight > 1 else "HIP events carried by the kernel's dispatch on the searcher's stream, the timed steps"),
        },
    }
    # the whole search against the same roofline: text bytes of one GPU / time per step (the kernel figure above
    # leaves out the chunk DP, traceback, launches and the host's share)
    out["roofline_search"] = {"bound": "hbm", "achieved": round(n_per / (ms_per_step / 1e3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                              "unit": "GB/s", "frac": round(n_per / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
                              "what": f"text bytes per GPU / ms_per_step, {max(1, args.in_flight)} search(es) in flight",
                              "frac_single_search": round(n_per / (latency_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4)}
    if world == 1 and not args.no_cpu_baseline:
        host = buf[:n_per].cpu().numpy()
        gpu_ends = [(int(e), int(c)) for e, c in zip(matches0.array["text_end"], matches0.array["cost"])]
        out["cpu_baseline"] = cpu_baseline(host, pat, k, args.profile, gpu_ends, args.cpu_seconds)
        out["h2d_inclusive"] = h2d_inclusive(sassy_amd, args.profile, pat, host, k, len(matches0))
        if not args.no_other_configs:
            out["other_configs"] = other_configs(sassy_amd, buf[:n_per])
            out["other_configs"]["2_dense"] = dense_config(sassy_amd, buf[:n_per], pat, k)
    print(json.dumps(out), flush=True)
    if gather_worker is not None:
        gather_worker.close()
    if dist is not None:
        dist.destroy_process_group()


def main_inproc(args):
    """--mode inproc: the same workload and the same JSON line, driven through sassy_hip_multi_* from this one process
    (a host thread, a bound searcher and a resident shard per device; shard results merged in C).  No torch, no RCCL."""
    import sassy_amd
    n_dev = sassy_amd.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device")
    world = args.gpus
    if world > n_dev and not args.allow_shared_gpu:
        raise SystemExit(f"--gpus {world} but only {n_dev} HIP device(s) visible (one shard per GPU)")
    devices = [g % n_dev for g in range(world)]
    n_per = args.text_bytes // 64 * 64
    total = n_per * world
    m, k = args.pattern_len, args.k
    pat = bytes(_dna_bytes(43, 0, m))
    ms = sassy_amd.MultiSearcher(args.profile, devices=devices)
    ms.generate_dna(total, 42, m, k)
    planted = ms.plant(42, pat, k, args.plant_stride)
    t_cold = time.perf_counter()
    r = ms.search(pat, k)
    cold_ms = (time.perf_counter() - t_cold) * 1e3
    # one search at a time: the dominant kernel's HIP-event duration for the roofline object (with searches in flight two
    # launches share the HBM) and the latency of a lone multi-device search
    n_lat = 30
    kern = [0.0] * world
    lat_t0 = time.perf_counter()
    for _ in range(n_lat):
        r = ms.search(pat, k)
        for g in range(world):
            kern[g] += ms.shard_stats(g)["filter_ms"] or ms.shard_stats(g)["scan_ms"]
    lone_ms = (time.perf_counter() - lat_t0) / n_lat * 1e3
    st = ms.shard_stats(0)
    dom_ms = max(kern) / n_lat
    # the stream of searches: `--in-flight` of them begun and not yet finished on every device
    # (sassy_hip_multi_search_begin / _finish); every timed search is begun AND finished inside the timed region
    depth = max(1, min(4, args.in_flight))
    ms.set_pipe_depth(depth)
    pending = []

    def step():
        if depth <= 1:
            return ms.search(pat, k)
        pending.append(ms.search_begin(pat, k))
        return ms.search_finish(pending.pop(0)) if len(pending) >= depth else None

    def drain(last):
        while pending:
            last = ms.search_finish(pending.pop(0))
        return last

    for _ in range(args.warmup + max(0, args.settle)):
        r = step() or r
    r = drain(r)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step() or r
    r = drain(r)
    elapsed = time.perf_counter() - t0
    dom_name = {0: "scan_kernel", 1: "filter_kernel", 2: "filter_dna_kernel", 3: "filter_table_kernel", 4: "filter_count_kernel"}[int(st["filtered"])]
    achieved = n_per / (dom_ms / 1e3) / 1e9 if dom_ms > 0 else 0.0
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": METRIC,
        "value": round(total * args.steps / elapsed / 1e9, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "cold_first_search_ms": round(cold_ms, 3),
        "config": {
            "workload": (f"BASELINE config {'2' if world == 1 else '5'}: Searcher::<{args.profile.capitalize()}>::new_fwd().search, "
                         f"|pattern|={m} (seeded random), k={k}, {n_per} B random-ACGT text per GPU resident in HBM, one planted "
                         f"near-match per {args.plant_stride} B; step = one sassy_hip_multi_search_begin / _finish pair: every device searches "
                         f"its shard at once (scan + traceback + Match records on the host), the shard results are chained and merged in C"),
            "text_bytes_per_gpu": n_per, "total_text_bytes": total, "pattern_len": m, "k": k, "profile": args.profile,
            "searches_in_flight": depth,
            "parallelism": f"text sharded x{world}, ONE process, a host thread per device (sassy_hip_multi_*), no torch.distributed / RCCL; devices {devices}"
                           + (f"; FALLBACK from the rank path: {args.fallback_reason}" if getattr(args, "fallback_reason", None) else ""),
            "setup": f"one cold search, {n_lat} one-at-a-time searches (single_search_latency_ms, the roofline object's kernel time), "
                     f"{args.warmup} warm-up + {max(0, args.settle)} settling searches, then the K timed searches ({depth} in flight)",
        },
        "single_search_latency_ms": round(lone_ms, 4),
        "matches": len(r),
        "matches_per_s": round(len(r) * args.steps / elapsed, 1),
        "planted": planted,
        "dominant_kernel_ms": round(dom_ms, 4),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": dom_name,
                     "algorithmic_bytes_per_launch": n_per, "launch_ms": round(dom_ms, 4),
                     "measured": f"HIP events carried by the kernel's dispatch on each shard searcher's stream, {n_lat} one-at-a-time multi-device searches of this run; the slowest device's average"},
        "roofline_search": {"bound": "hbm", "achieved": round(n_per / (ms_per_step / 1e3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(n_per / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "what": f"text bytes per GPU / ms_per_step, {depth} multi-device search(es) in flight"},
        "cpu_baseline": None,
        "cpu_baseline_note": "reported by the default mode at N = 1 (python bench.py)",
    }
    print(json.dumps(out), flush=True)


def dense_config(sassy_amd, text, pat, k):
    """SURVEY 8(d)'s dense variant of config 2 on the resident text (N = 1, after everything else: it plants a near-match
    of the bench pattern every 4 096 bytes): the output path -- matches per second with the records on the host."""
    n = text.numel()
    planted = sassy_amd.plant(text.data_ptr(), n, 0, n, 42, pat, k, 4096)
    s = sassy_amd.Searcher("dna", rc=False)
    s.set_timing(0)
    for _ in range(4):
        r = s.search_shard(pat, text.data_ptr(), 0, n, 0, n, k)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        r = s.search_shard(pat, text.data_ptr(), 0, n, 0, n, k)
    dt = (time.perf_counter() - t0) / reps
    st = s.stats()
    return {"workload": f"config 2 with a planted near-match every 4 096 B ({planted} plants), {n} B, lone searches",
            "ms_per_search": round(dt * 1e3, 3), "matches": len(r), "matches_per_s": round(len(r) / dt, 1),
            "text_GB_per_s": round(n / dt / 1e9, 1), "fused": int(st["fused"]), "path": int(st["filtered"])}


# lone searches of a fresh searcher settle over their first ~20 calls (0.79 -> 0.65 ms for config 3 on an MI355X that has just
# run the planting kernels: clocks, not the library's state); the 20 timed calls follow that many untimed ones
LONE_WARMUP = int(os.environ.get("BENCH_LONE_WARMUP", "25"))


def other_configs(sassy_amd, text):
    """BASELINE configs 3 and 4 on the text that is resident anyway (N = 1, after the timed steps; reported next
    to the bench line, never part of `value`): config 3 = one 200-row Iupac pattern, k = 20, with one planted
    near-match of it per MiB (planted here, behind the timed region); config 4 = search_encoded_patterns with
    10 000 pre-encoded 20-mers, k = 2, Iupac searcher.  Parity of both at this size: tests/."""
    n = text.numel()
    res = {}
    p = bytearray(_dna_bytes(44, 0, 200))
    p[50], p[100], p[150], p[199] = ord("N"), ord("R"), ord("Y"), ord("W")
    # near-matches of THIS pattern (its ambiguity letters spelled with a base they contain), one per MiB, a quarter
    # MiB behind the bench pattern's: the search then has a tail to run -- chunk DP and 200 x 43-band tracebacks
    plain = bytes({ord("N"): 65, ord("R"): 65, ord("Y"): 67, ord("W"): 65}.get(c, c) for c in p)
    shift = 1 << 18
    planted3 = sassy_amd.plant(text.data_ptr() + shift, n - shift, 0, n - shift, 44, plain, 20, 1 << 20)
    s3 = sassy_amd.Searcher("iupac", rc=False)
    r = s3.search_shard(bytes(p), text.data_ptr(), 0, n, 0, n, 20)
    path3 = s3.stats()["filtered"]
    s3.set_timing(0)  # (no kernel events around the filter: a lone search as a caller runs it)
    for _ in range(LONE_WARMUP):
        r = s3.search_shard(bytes(p), text.data_ptr(), 0, n, 0, n, 20)
    t0 = time.perf_counter()
    for _ in range(20):
        r = s3.search_shard(bytes(p), text.data_ptr(), 0, n, 0, n, 20)
    dt = (time.perf_counter() - t0) / 20
    res["3"] = {"workload": f"Iupac new_fwd, |pattern|=200 (N, R, Y, W at 50/100/150/199), k=20, {n} B",
                "ms_per_search": round(dt * 1e3, 3), "text_GB_per_s": round(n / dt / 1e9, 1), "matches": len(r),
                "planted": int(planted3), "roofline_frac": round(n / dt / 1e9 / HBM_PEAK_GBPS, 4), "path": path3}
    # ... and as a stream of searches, three in flight (what a pool of workers sees: bin/grep.rs:516-537), alternating between
    # this pattern and a second one of the same shape with plants of its own -- a lane never meets the pattern it searched last
    pb = bytearray(_dna_bytes(54, 0, 200))
    pb[50], pb[100], pb[150], pb[199] = ord("N"), ord("R"), ord("Y"), ord("W")
    plain_b = bytes({ord("N"): 65, ord("R"): 65, ord("Y"): 67, ord("W"): 65}.get(c, c) for c in pb)
    shift_b = 3 << 18
    planted3b = sassy_amd.plant(text.data_ptr() + shift_b, n - shift_b, 0, n - shift_b, 54, plain_b, 20, 1 << 20)
    s3.set_pipe_depth(3)
    pats3 = (bytes(p), bytes(pb))
    pend, got = [], {}
    def flight_step(i):
        pend.append((i & 1, s3.search_shard_begin(pats3[i & 1], text.data_ptr(), 0, n, 0, n, 20)))
        if len(pend) >= 3:
            which, t = pend.pop(0)
            got[which] = len(s3.search_finish(t))
    for i in range(40):
        flight_step(i)
    t0 = time.perf_counter()
    for i in range(40, 140):
        flight_step(i)
    while pend:
        which, t = pend.pop(0)
        got[which] = len(s3.search_finish(t))
    dtf = (time.perf_counter() - t0) / 100
    res["3"]["in_flight_3"] = {"what": "100 searches, three in flight, two patterns of this shape alternating (each with its own plants)",
                               "ms_per_search": round(dtf * 1e3, 3), "roofline_frac": round(n / dtf / 1e9 / HBM_PEAK_GBPS, 4),
                               "matches_per_pattern": [got.get(0), got.get(1)], "planted_second_pattern": int(planted3b)}
    # the reference's own benchmark shape (benches/perf.rs:46-48: |pattern| = 23, k = 3 -- every CRISPR guide 20 + PAM) and
    # m = 32 with k = 4: shapes whose k+1 pigeonhole pieces are 5 / 6 rows -- the paired filter's fused launch
    for name, profile, m_, k_ in (("m23k3", "dna", 23, 3), ("m23k3_iupac", "iupac", 23, 3), ("m32k4", "dna", 32, 4)):
        ps = bytes(_dna_bytes(49, 0, m_))
        ss = sassy_amd.Searcher(profile, rc=False)
        r = ss.search_shard(ps, text.data_ptr(), 0, n, 0, n, k_)
        st = ss.stats()
        ss.set_timing(0)
        for _ in range(LONE_WARMUP):
            r = ss.search_shard(ps, text.data_ptr(), 0, n, 0, n, k_)
        t0 = time.perf_counter()
        for _ in range(20):
            r = ss.search_shard(ps, text.data_ptr(), 0, n, 0, n, k_)
        dt = (time.perf_counter() - t0) / 20
        res[name] = {"workload": f"{profile.capitalize()} new_fwd, |pattern|={m_}, k={k_}, {n} B, lone searches",
                     "ms_per_search": round(dt * 1e3, 3), "text_GB_per_s": round(n / dt / 1e9, 1), "matches": len(r),
                     "roofline_frac": round(n / dt / 1e9 / HBM_PEAK_GBPS, 4), "path": int(st["filtered"]), "fused": int(st["fused"]),
                     "pair": int(st.get("pair", 0)), "piece_len": int(st["piece_len"])}
    flat = _dna_bytes(45, 0, 20 * 10_000).tobytes()
    pats = [flat[20 * i:20 * i + 20] for i in range(10_000)]
    s4 = sassy_amd.Searcher("iupac", rc=False)
    enc = s4.encode_patterns(pats)
    secs = []
    for _ in range(2):  # the first call also sizes the device buffers
        t0 = time.perf_counter()
        r = s4.search_encoded_patterns(enc, text, 2, as_result=True)
        secs.append(time.perf_counter() - t0)
    res["4"] = {"workload": f"Iupac new_fwd, search_encoded_patterns, 10 000 seeded random 20-mers, k=2, {n} B",
                "seconds": round(min(secs), 4), "seconds_first_call": round(secs[0], 4),
                "pattern_text_TB_per_s": round(n * 10_000 / min(secs) / 1e12, 1), "matches": len(r),
                "path": s4.stats()["filtered"]}
    # a CRISPR guide set (the reference's off-target workflow): 312 guides of 20 bases with their PAM "NGG", k = 3, both
    # strands, through search_encoded_patterns -- the seeded search with ambiguity letters in its seeds
    import numpy as np
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    guides = [bytes(acgt[rng.integers(0, 4, 20)]) + b"NGG" for _ in range(312)]
    sg = sassy_amd.Searcher("iupac", rc=True)
    encg = sg.encode_patterns(guides)
    secs = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = sg.search_encoded_patterns(encg, text, 3, as_result=True)
        secs.append(time.perf_counter() - t0)
    res["crispr_guides"] = {"workload": f"Iupac searcher, both strands, search_encoded_patterns, 312 guides (20 + NGG), k=3, {n} B",
                            "seconds": round(min(secs), 4), "seconds_first_call": round(secs[0], 4),
                            "pattern_text_TB_per_s": round(n * 624 / min(secs) / 1e12, 1), "matches": len(r),
                            "path": sg.stats()["filtered"]}
    # the reference's default for DNA is both strands (c/example.c:14, python default): the benchmark shape with rc = true,
    # the searcher told that the text did not change (the reversed copy is kept)
    ps = bytes(_dna_bytes(49, 0, 23))
    sb = sassy_amd.Searcher("dna", rc=True)
    sb.search(ps, text, 3)  # (makes the reversed copy)
    sb.text_unchanged(True)
    sb.set_timing(0)
    for _ in range(LONE_WARMUP):
        r = sb.search(ps, text, 3)
    t0 = time.perf_counter()
    for _ in range(20):
        r = sb.search(ps, text, 3)
    dt = (time.perf_counter() - t0) / 20
    res["m23k3_both_strands"] = {"workload": f"Dna, both strands, |pattern|=23, k=3, {n} B, lone searches, text unchanged",
                                 "ms_per_search": round(dt * 1e3, 3), "matches": len(r)}
    return res


def h2d_inclusive(sassy_amd, profile, pat, host_text, k, want_matches):
    """The drop-in entry point's rate: `search(searcher, pattern, text, ...)` of include/sassy.h takes a HOST
    text pointer (reference: src/c.rs:89-122), so each call moves the text over PCIe before it can be
    scanned.  Timed here on the same text through the same C symbol, after one untimed call (buffers)."""
    import ctypes as C
    L = sassy_amd.lib()
    s = L.sassy_searcher(profile.encode(), False, float("nan"))
    out = C.POINTER(sassy_amd.CMatch)()
    addr = host_text.ctypes.data
    n = host_text.size
    times = []
    cnt = 0
    for i in range(3):
        t0 = time.perf_counter()
        cnt = L.search(s, pat, len(pat), C.cast(addr, C.c_char_p), n, k, C.byref(out))
        dt = time.perf_counter() - t0
        L.sassy_matches_free(out, cnt)
        if i:
            times.append(dt)
    L.sassy_searcher_free(s)
    best = min(times)
    # the link's own ceiling on this box, measured the same minute: one hipMemcpy of 1 GiB from PINNED host memory
    link = None
    try:
        import torch
        nb = min(n, 1 << 30)
        src = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
        dst = torch.empty(nb, dtype=torch.uint8, device="cuda")
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        link = round(3 * nb / (time.perf_counter() - t0) / 1e9, 2)
        del src, dst
    except Exception:  # noqa: BLE001 -- the ceiling is context, never a reason to lose the bench line
        link = None
    return {"value": round(n / best / 1e9, 2), "unit": "GB/s", "ms_per_search": round(best * 1e3, 2),
            "what": "drop-in search() of include/sassy.h on a host text (pageable numpy memory): upload over PCIe + "
                    "scan + matches, best of 2 calls after one untimed call; never the headline value",
            "link_ceiling_GB_per_s": link,
            "link_ceiling_what": "hipMemcpy host -> device of 1 GiB from pinned memory on this box (one PCIe link), three copies back to back",
            "matches": int(cnt), "matches_equal_resident": int(cnt) == int(want_matches)}


def _dna_bytes(seed, first, n):
    """The synthetic-text function of SURVEY 8(d) in numpy (pattern generation only)."""
    import numpy as np
    idx = np.arange(first, first + n, dtype=np.uint64)
    blk = idx >> np.uint64(5)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + blk
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    code = (x >> (np.uint64(2) * (idx & np.uint64(31)))) & np.uint64(3)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[code.astype(np.int64)]


if __name__ == "__main__":
    main()
