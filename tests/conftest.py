import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "kats.json")) as f:
        return json.load(f)


def build_text(entry):
    """Materialise the 'text' of a KAT entry (literal, or fill + splices as the reference test
    builds it with Vec::splice)."""
    if "text" in entry:
        return entry["text"].encode()
    tb = entry["text_build"]
    t = bytearray(tb["fill"].encode() * tb["len"])
    for at, s in tb["splices"]:
        t[at:at] = s.encode()
    return bytes(t)


def build_block(spec):
    b = bytearray(spec["fill"].encode() * 64)
    for at, ch in spec["set"]:
        b[at] = ord(ch)
    return bytes(b)


def expand_positions(v):
    if isinstance(v, list):
        return v
    out = []
    for part in v.split("+"):
        a, b = part.strip()[len("range("):-1].split(",")
        out.extend(range(int(a), int(b)))
    return out


def cigar_path(m):
    """Match::to_path (src/search.rs:83-103) for a forward match: list of (pattern, text) pairs."""
    import re
    j, i = m.pattern_start, m.text_start
    path = [(j, i)]
    for cnt, op in re.findall(r"(\d+)([=XID])", m.cigar):
        for _ in range(int(cnt)):
            if op in "=X":
                j, i = j + 1, i + 1
            elif op == "I":
                j += 1
            else:
                i += 1
            path.append((j, i))
    path.pop()
    return path
