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


def test_c4_cong_4k_sentences_bit_exact_vs_oracle_and_reference():
    """BASELINE config 4's model type on the benchmarked 'full-cong' model (bench.py --workload c4-cong): device vs the CPU oracle and, where the
    prebuilt x86 reference library travelled, vs the REAL src/CoNgramModel.cpp (SSE4.1 build)."""
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, c4, _ = get_workload("c4-cong")
    dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
    _check(dev, orc, c4[:4096])
    if refbridge.x86_available():
        _check(dev, refbridge.RefKiwi(path, arch=3, x86=True), c4[:4096])
    dev.close()
