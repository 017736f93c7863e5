"""The HIP path (through the C-ABI) on the second harvest of the reference's own tests (tests/golden/kats_more.json):
every vector must satisfy what the reference asserts AND give the oracle's matches, field for field."""
import pytest

import oracle
import kat_props
from test_reference_props import oracle_engine

pytestmark = pytest.mark.gpu
K = kat_props.load()


@pytest.fixture(scope="module")
def sassy():
    import sassy_amd
    assert sassy_amd.device_count() > 0, "no HIP device: the GPU tests must not silently skip"
    return sassy_amd


def key(m):
    return (m.pattern_idx, m.text_start, m.text_end, m.pattern_start, m.pattern_end, m.cost, m.strand, m.cigar)


@pytest.mark.parametrize("e", K["properties"], ids=[e["id"] for e in K["properties"]])
def test_reference_property_through_hip(sassy, e):
    calls = []

    def hip_engine(profile, rc, alpha, max_n_frac, all_minima, pattern, text, k):
        s = sassy.Searcher(profile, rc=rc, alpha=alpha)
        if max_n_frac is not None:
            s.with_max_n_frac(max_n_frac)
        got = s.search_all(pattern, text, k) if all_minima else s.search(pattern, text, k)
        want = oracle_engine(profile, rc, alpha, max_n_frac, all_minima, pattern, text, k)
        assert [key(m) for m in got] == [key(m) for m in want], (e["id"], got[:4], want[:4], len(got), len(want))
        calls.append(len(got))
        return got

    kat_props.check(e, hip_engine)
    assert calls


@pytest.mark.parametrize("e", K["encoded_properties"], ids=[e["id"] for e in K["encoded_properties"]])
def test_reference_encoded_property_through_hip(sassy, e):
    pats = [p.encode() for p in e["patterns"]]
    s = sassy.Searcher(e["profile"], rc=e["rc"])
    got = s.search_encoded_patterns(s.encode_patterns(pats), e["text"].encode(), e["k"], all_minima=e.get("all", False))
    want = oracle.search_encoded(e["profile"], pats, e["text"].encode(), e["k"], rc=e["rc"], all_minima=e.get("all", False))
    assert sorted(key(m) for m in got) == sorted(key(m) for m in want), (e["id"], got[:4], want[:4])
    if e["prop"] == "nonempty":
        assert got
    else:
        assert len(got) == e["n"]
