"""GPU (-m gpu): the BENCHMARKED configuration itself -- the synthetic 'full' model and the bench corpora of kiwi_amd/workloads.py
(BASELINE config 2: 8192 x 40-jamo; config 3's corpus: mixed 5-200 jamo) -- device vs the real reference translation units
(oracle/_ref, when the prebuilt library travelled) and vs the CPU oracle: tokens, positions and fp32 path scores, bit for bit.
bench.py's numbers are for exactly these outputs."""
from dataclasses import astuple

import pytest

pytestmark = pytest.mark.gpu


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


@pytest.fixture(scope="module")
def full():
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, c2, _ = get_workload("c2")
    _, c3, _ = get_workload("c3")
    dev = KiwiAmd(path)
    orc = oraclelib.OracleKiwi(path)
    ref = refbridge.RefKiwi(path) if refbridge.available() else None
    yield dev, orc, ref, c2, c3
    dev.close()


def _check(dev, cpu, texts, top_n=1):
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    assert len(got) == len(texts)
    bad = [s for s, y in zip(texts, got) if _norm(cpu.analyze(s, top_n=top_n)) != _norm(y)]
    assert not bad, (len(bad), bad[:3])


def test_c2_all_sentences_bit_exact_vs_oracle(full):
    dev, orc, _, c2, _ = full
    _check(dev, orc, c2)


def test_c2_all_sentences_bit_exact_vs_real_reference(full):
    dev, _, ref, c2, _ = full
    if ref is None:
        pytest.skip("oracle/_ref/libkiwi_ref.so not built")
    _check(dev, ref, c2)


def test_c3_4k_sentences_bit_exact_vs_oracle_and_reference(full):
    dev, orc, ref, _, c3 = full
    _check(dev, orc, c3[:4096])
    if ref is not None:
        _check(dev, ref, c3[:4096])
