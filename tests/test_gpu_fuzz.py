"""A bounded, seeded slice of the differential fuzzer (tests/fuzz_gpu.py) inside the driver's `pytest -m gpu` run: every
case family -- single searches on every profile and shape, the fused / paired launches, patterns beyond 64 distinct bytes
and long patterns, the q-gram counting filter with both strands, search_many (with overhang), search_encoded (seeded,
tiled, per-pattern; with overhang on one text), shards with seams, searches in flight, the reference-lane mode -- runs for a fixed time from a fixed
seed in a process of its own, every result compared with the oracle.  The line the fuzzer prints (cases, matches
compared, prefilter kinds that ran) goes to the test's output, so that the driver's record carries it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = ["one", "fused", "bytes_long", "count", "many", "encoded", "shard", "inflight", "reflanes", "ovenc"]
SECONDS = 15  # per family: 10 families, ~3 minutes in all (the driver gives the whole GPU suite 20 minutes)


@pytest.mark.gpu
@pytest.mark.parametrize("family", FAMILIES)
def test_fuzz_slice_against_the_oracle(family, capsys):
    env = {k: v for k, v in os.environ.items() if not k.startswith("SASSY_HIP_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "--seconds", str(SECONDS), "--seed", "6",
                        "--focus", family], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, (family, tail)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("fuzz ok:")]
    assert line, (family, tail)
    cases = int(line[-1].split()[2])
    assert cases >= 3, (family, line[-1])
    with capsys.disabled():  # (shown with -q as well: the driver's log carries the cases and matches compared)
        print(f"\n[fuzz slice] {family}: {line[-1]}", flush=True)
