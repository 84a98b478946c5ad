"""`-m gpu`: the GLOBAL CoNgram model (ModelType::congGlobal, window 7) searched on the MI355X -- viterbi_kernel_congg.hip,
viterbi_kernel_congg_typo.hip behind kamd_open_mode(lm_mode = 4) and kiwi_init(KIWI_BUILD_MODEL_TYPE_CONG_GLOBAL / _LARGEST) -- against the golden analyses of
the real reference (tests/golden/cong_global_*.json) and against the oracle, which tests/test_cong_global.py pins to the real reference's SSE4.1 build."""
import ctypes as C
import os

import pytest

from corpora import EDGE_TEXTS, force_lanes, fuzzed
from test_cong_global import _model, _oracle, _rows, check_device_goldens, check_device_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["32", "16"])
def test_golden_analyses_of_the_reference(which):
    assert check_device_goldens(None, which) == 271


@pytest.mark.parametrize("which,lanes", [("32", "64"), ("16", "16")])
def test_random_sentences_vs_oracle(monkeypatch, which, lanes):
    """2000 synthetic sentences + 500 dictionary mixes + edge texts + fuzzed input, top-1; the first 300 top-3.  A quarter of the sentences put more than 64
    entries into a path container: the replay of the reference's container behaviour there runs thousands of times."""
    from kiwi_amd import synth
    force_lanes(monkeypatch, lanes)
    sm = synth.SynthModel(synth.SMALL_CONG_GLOBAL_SPEC if which == "32" else synth.SMALL_CONG_GLOBAL16_SPEC)
    n, past64 = check_device_vs_oracle(None, which, 2000, 300, max_jamo=150, extra=EDGE_TEXTS + fuzzed(sm, 300, 929))
    assert n >= 2800 and past64 > 50000, (n, past64)


def test_long_sentences_vs_oracle():
    """Sentences of up to 400 jamo: hundreds of paths per node (the medium and large containers, histories through the HBM arena)."""
    n, past64 = check_device_vs_oracle(None, "32", 200, 40, max_jamo=400, seed=951)
    assert n >= 240 and past64 > 20000, (n, past64)


def test_typo_correction_vs_oracle():
    """The typo-correcting analysis with the global model (viterbi_kernel_congg_typo.hip): the built-in set basicTypoSetWithContinual, misspelt sentences; the
    oracle gets the same entries out of oracle/_ref (skipped where that library did not travel: the golden typo analyses above still ran)."""
    import random
    import oraclelib
    import refbridge
    from corpora import synthetic
    from kiwi_amd.api import KiwiAmd, Typo
    from typo_cases import misspell
    if not refbridge.available():
        pytest.skip("oracle/_ref not built: no built-in typo set for the oracle")
    sm, path = _model("32")
    orc = _oracle(path)
    ents, cont, leng = refbridge.default_typo_entries("basic_with_continual")
    ot = oraclelib.OracleTypo(); ot.update_entries(ents, cont, leng); ot.prepare(True)
    dev = KiwiAmd(path, lm_mode=4)
    ty = Typo.from_default(dev.lib, 3).prepare(True)
    rnd = random.Random(11)
    texts = [misspell(t, rnd, True, True) for t in synthetic(sm, 250, 961, min_jamo=5, max_jamo=80)]
    for top_n in (1, 2):
        got = dev.analyze_batch_opt(texts[:250 if top_n == 1 else 60], top_n=top_n, typo=ty, typo_threshold=2.5).to_python()
        corrected = 0
        for s, y in zip(texts, got):
            want = orc.analyze_typo(ot, s, 2.5, 0, top_n=top_n)
            assert _rows(want) == _rows(y), s
            corrected += any(t.typo_cost > 0 for t in want[0][0])
        assert corrected > 10
    dev.close()


def test_kiwi_init_model_types():
    """kiwi_init: CONG_GLOBAL and LARGEST resolve to the global scoring when the CoNgram blob has window sections (KiwiBuilder.cpp:939-946), CONG and the default
    to the local one; CONG_GLOBAL on a blob without the sections is refused."""
    import oraclelib
    from corpora import synthetic
    from test_gpu_capi import LIB, MATCH_ALL_WITH_NORMALIZING, Option
    sm, path = _model("32")
    L = C.CDLL(LIB)
    L.kiwi_init.restype = C.c_void_p
    L.kiwi_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.kiwi_analyze.restype = C.c_void_p
    L.kiwi_analyze.argtypes = [C.c_void_p, C.c_char_p, C.c_int, Option, C.c_void_p]
    L.kiwi_res_prob.restype = C.c_float
    L.kiwi_res_prob.argtypes = [C.c_void_p, C.c_int]
    L.kiwi_res_close.argtypes = [C.c_void_p]
    L.kiwi_close.argtypes = [C.c_void_p]
    L.kiwi_error.restype = C.c_char_p
    opt = Option(MATCH_ALL_WITH_NORMALIZING, None, 0, 0, 3.0, None, 2.5)
    glob = _oracle(path)
    local = oraclelib.OracleKiwi(path)
    texts = synthetic(sm, 60, 971, min_jamo=10, max_jamo=80)
    differ = 0
    for options, want in ((15 | 0x0500, glob), (15 | 0x0100, glob), (15 | 0x0400, local), (15, local)):
        k = L.kiwi_init(path.encode(), 0, options, 0)
        assert k, L.kiwi_error()
        for s in texts:
            r = L.kiwi_analyze(k, s.encode("utf-8"), 1, opt, None)
            assert r, L.kiwi_error()
            assert L.kiwi_res_prob(r, 0) == want.analyze(s)[0][1], (hex(options), s)
            differ += glob.analyze(s)[0][1] != local.analyze(s)[0][1]
            L.kiwi_res_close(r)
        L.kiwi_close(k)
    assert differ > 40      # (the two scorings do give different analyses on this corpus)
    from kiwi_amd import synth
    plain = os.path.join(os.path.dirname(path), "small-cong.raw")
    synth.SynthModel(synth.SMALL_CONG_SPEC).raw.save(plain)
    assert not L.kiwi_init(plain.encode(), 0, 15 | 0x0500, 0) and b"window" in L.kiwi_error()
