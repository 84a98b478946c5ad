"""GPU (-m gpu): typo correction on the device (kiwi_amd/csrc/typo.cpp on the host, typo_lattice_kernel.hip, viterbi_kernel_typo.hip) against
the CPU oracle, whose typo path is pinned to the real reference by tests/test_typo_oracle.py.

First run on an MI355X in round 2 (profiles/r02_a_*): all green at first contact; the experimental gate is gone."""
import os
import random

import pytest

from corpora import EDGE_TEXTS, dictionary_mix, force_lanes, synthetic
from test_hipemu import _analyze_typo, _norm, _typo_lattices, _typo_pair

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip.so")


@pytest.mark.parametrize("continual,threshold,top_n,lanes,lengthening", [(float("inf"), 2.5, 1, "16", float("inf")), (1.0, 2.5, 1, "16", float("inf")), (1.0, 1.2, 3, "16", float("inf")),
                                                                         (1.0, 2.5, 2, "64", float("inf")), (1.0, 2.5, 1, "16", 0.25),
                                                                         (1.0, 2.5, 1, "pos", 0.25), (float("inf"), 2.5, 1, "pos", float("inf"))])
def test_typo_analyses_bit_exact_vs_oracle(small_model, monkeypatch, continual, threshold, top_n, lanes, lengthening):
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_model
    force_lanes(monkeypatch, lanes)
    prod, orc_t = _typo_pair(LIB, continual, lengthening)
    dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
    rnd = random.Random(11)
    texts = [misspell(t, rnd, True, continual == 1.0, lengthening < 1e9) for t in synthetic(sm, 400, 591, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 200, 592)] + EDGE_TEXTS
    got = _analyze_typo(dev, prod, texts, threshold, top_n)
    for t, y in zip(texts, got):
        assert _norm(orc.analyze_typo(orc_t, t, threshold, 0, top_n=top_n)) == _norm(y), t
    for t in texts[:100]:
        if t.strip():
            assert _typo_lattices(dev, prod, t, threshold) == orc.split_typo(orc_t, t, threshold), t
    dev.close(); prod.close()


def test_typo_analyses_through_the_capacity_ladder(small_model, monkeypatch):
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_model
    monkeypatch.setenv("KAMD_TEST_TINY_ARENAS", "1")
    prod, orc_t = _typo_pair(LIB, 1.0)
    dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
    rnd = random.Random(13)
    texts = [misspell(t, rnd) for t in synthetic(sm, 150, 593, min_jamo=5, max_jamo=120)]
    for t, y in zip(texts, _analyze_typo(dev, prod, texts, 2.5)):
        assert _norm(orc.analyze_typo(orc_t, t, 2.5, 0)) == _norm(y), t
    dev.close(); prod.close()


@pytest.mark.parametrize("lanes,top_n,continual,lengthening", [("pos", 1, 1.0, 0.25), ("pos8", 1, 1.0, 0.25), ("16", 1, 1.0, float("inf")), ("64", 2, 1.0, 0.25)])
def test_typo_correction_with_a_cong_model(small_cong_model, monkeypatch, lanes, top_n, continual, lengthening):
    """The reference's default model type with typo correction: viterbi_kernel_cong_typo.hip (CoNgram scoring + node typo costs) on the MI355X
    against the oracle (pinned for this combination to the reference's SSE4.1 build) and, where it travelled, the real reference itself."""
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_cong_model
    force_lanes(monkeypatch, lanes)
    prod, orc_t = _typo_pair(LIB, continual, lengthening)
    dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
    rnd = random.Random(19)
    texts = [misspell(t, rnd, True, continual == 1.0, lengthening < 1e9) for t in synthetic(sm, 400, 751, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 200, 752)] + EDGE_TEXTS
    got = _analyze_typo(dev, prod, texts, 2.5, top_n)
    corrected = 0
    for t, y in zip(texts, got):
        want = orc.analyze_typo(orc_t, t, 2.5, 0, top_n=top_n)
        assert _norm(want) == _norm(y), t
        corrected += any(x.typo_cost > 0 for x in want[0][0])
    assert corrected >= 50
    dev.close(); prod.close()


def test_device_typo_graphs_equal_the_host_module(small_model):
    """k_typo_graph (typo_graph_kernel.hip: the graphs the analyze path builds its lattices over, count pass + write pass) against the host
    module's graphs -- byte-identical to the real reference's (tests/test_typo_product.py) -- for the repo's own rules in both directions,
    every built-in set, dialect masks; node records, links, continual indices and the last-character facts the lattice build reads."""
    from test_hipemu import check_device_typo_graphs
    check_device_typo_graphs(LIB, small_model[1], n_random=400)
