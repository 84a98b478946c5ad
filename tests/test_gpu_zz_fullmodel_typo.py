"""GPU (-m gpu): BASELINE config 5 on the benchmarked model itself -- the 'full' synthetic model, the c2 corpus misspelt, the typo transformer of
`bench.py --workload c5` (kiwi_amd.workloads.TYPO_RULES, continual cost 1, threshold 2.5): typo graphs from k_typo_graph, lattices over them,
the typo search kernel -- against the CPU oracle (pinned to the real reference's typo path by tests/test_typo_oracle.py), tokens, positions,
fp32 scores and per-token typo costs.  (Sorted last on purpose: it is the one GPU test of the round that was written after the last GPU call.)"""
import os

import pytest

from test_hipemu import _analyze_typo, _norm, _typo_pair

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip.so")


def check_c5(lib, n):
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload, workload_typo
    path, c5, _ = get_workload("c5")
    cont, leng, threshold = workload_typo("c5")
    prod, orc_t = _typo_pair(lib, cont, leng)
    dev, orc = KiwiAmd(path, lib_path=lib), oraclelib.OracleKiwi(path)
    texts = c5[:n]
    got = _analyze_typo(dev, prod, texts, threshold)
    bad = [t for t, y in zip(texts, got) if _norm(orc.analyze_typo(orc_t, t, threshold, 0)) != _norm(y)]
    corrected = sum(any(tok.typo_cost > 0 for tok in y[0][0]) for y in got)
    dev.close(); prod.close()
    assert not bad, (len(bad), bad[:3])
    assert corrected > n // 20
    return corrected


def test_c5_2k_sentences_bit_exact_vs_oracle():
    check_c5(LIB, 2048)


def test_c5_position_step_kernel_equals_general_kernel_on_the_whole_corpus(monkeypatch):
    """All 8192 misspelt sentences of c5 through the typo compilation of the position-step kernel and through the general kernel alone
    (KAMD_POS_PATH=0): the same packed token records, typo costs included."""
    import numpy as np
    from kiwi_amd.api import KiwiAmd, Typo
    from kiwi_amd.workloads import fill_typo_rules, get_workload, workload_typo
    path, c5, _ = get_workload("c5")
    cont, leng, threshold = workload_typo("c5")
    out = []
    for pos in ("2", "0"):      # (2: the step kernel also for typo correction -- the engine itself takes the general kernel there, which is faster on such lattices)
        monkeypatch.setenv("KAMD_POS_PATH", pos)
        dev = KiwiAmd(path, lib_path=LIB)
        typo = Typo(dev.lib, cont, leng)
        fill_typo_rules(typo)
        typo.prepare(True)
        r = dev.analyze_batch_opt(c5, typo=typo, typo_threshold=threshold)
        out.append(r.pack()); r.close()
        typo.close(); dev.close()
    assert out[0].nbytes == out[1].nbytes and np.array_equal(out[0], out[1])


@pytest.mark.parametrize("lanes", ["64", "16"])
def test_typo_correction_with_a_skipbigram_model(small_sbg_model, monkeypatch, lanes):
    """viterbi_kernel_sbg_typo.hip: the SkipBigram search kernel over lattices with typo costs, against the oracle (compared with the real
    reference on this combination by tests/test_typo_oracle.py)."""
    import random
    import oraclelib
    from corpora import EDGE_TEXTS, dictionary_mix, force_lanes, synthetic
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_sbg_model
    force_lanes(monkeypatch, lanes)
    prod, orc_t = _typo_pair(LIB, 1.0)
    dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
    rnd = random.Random(25)
    texts = [misspell(t, rnd, True, True) for t in synthetic(sm, 120, 941, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 60, 942)] + EDGE_TEXTS
    got = _analyze_typo(dev, prod, texts, 2.5)
    for t, y in zip(texts, got):
        assert _norm(orc.analyze_typo(orc_t, t, 2.5, 0)) == _norm(y), t
    dev.close(); prod.close()


def test_model_variant_goldens_from_the_real_reference():
    """History-transformed quantised sj.knlm; CoNgram without / with Match::oovChrModel: the device against the committed outputs of the real
    reference (tests/golden/model_variants_golden.json)."""
    from test_hipemu import check_device_against_model_variant_goldens
    check_device_against_model_variant_goldens(LIB)
