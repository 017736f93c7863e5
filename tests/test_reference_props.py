"""The oracle against the second harvest of the reference's own tests (tests/golden/kats_more.json): long patterns
(m = 126, k = 44 .. 63), overhang with both strands, Rc twins, Ascii, the n_frac filters.  CPU only."""
import pytest

import oracle
import kat_props

K = kat_props.load()


def oracle_engine(profile, rc, alpha, max_n_frac, all_minima, pattern, text, k):
    if max_n_frac is not None:
        return oracle.search_modes(profile, pattern, text, k, rc=rc, all_minima=all_minima, max_n_frac=max_n_frac, alpha=alpha)
    if alpha is not None:
        return oracle.search_overhang(profile, pattern, text, k, alpha, rc=rc, all_minima=all_minima)
    return oracle.search(profile, pattern, text, k, rc=rc, all_minima=all_minima)


def test_harvest_size():
    assert len(K["properties"]) + len(K["encoded_properties"]) + len(K["profile_masks"]) >= 55


@pytest.mark.parametrize("e", K["properties"], ids=[e["id"] for e in K["properties"]])
def test_reference_property(e):
    kat_props.check(e, oracle_engine)


@pytest.mark.parametrize("e", K["encoded_properties"], ids=[e["id"] for e in K["encoded_properties"]])
def test_reference_encoded_property(e):
    ms = oracle.search_encoded(e["profile"], [p.encode() for p in e["patterns"]], e["text"].encode(), e["k"], rc=e["rc"],
                               all_minima=e.get("all", False))
    if e["prop"] == "nonempty":
        assert ms
    else:
        assert len(ms) == e["n"], ms


def test_ascii_profile_masks():
    from conftest import build_block, expand_positions
    for e in K["profile_masks"]:
        masks = oracle.profile_masks(e["profile"], e["pattern"].encode(), build_block(e["block"]))
        for slot, exp in e["expect_positions"].items():
            got = [b for b in range(64) if (masks[int(slot)] >> b) & 1]
            assert got == expand_positions(exp), (e["id"], slot, got)
