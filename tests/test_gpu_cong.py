"""GPU (-m gpu): the search kernel for CoNgram models (kiwi_amd/csrc/viterbi_kernel_cong.hip) against the CPU oracle, whose CoNgram path is
pinned to the REAL reference (src/CoNgramModel.cpp in its SSE4.1 build, tests/test_cong_oracle.py) -- and, where the prebuilt
oracle/_ref/libkiwi_ref_x86.so travelled, against that reference directly: tokens, positions, fp32 scores, bit for bit."""
import os
from dataclasses import astuple

import pytest

from corpora import EDGE_TEXTS, dictionary_mix, force_lanes, fuzzed, synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


@pytest.mark.parametrize("lanes,top_n", [("pos", 1), ("pos8", 1), ("16", 1), ("64", 1), ("16", 2), ("64", 2)])
def test_cong_tokens_bit_exact_vs_oracle(small_cong_model, monkeypatch, lanes, top_n):
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_model
    force_lanes(monkeypatch, lanes)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path)
    texts = synthetic(sm, 1200, 921, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 500, 922) + EDGE_TEXTS + fuzzed(sm, 400, 923)
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (lanes, top_n, s)
    dev.close()


def test_cong_against_real_reference_when_present(small_cong_model):
    import refbridge
    from kiwi_amd.api import KiwiAmd
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    sm, path = small_cong_model
    ref = refbridge.RefKiwi(path, arch=3, x86=True)       # the reference's SSE4.1 build: the pin (its AVX2 build rounds batched scores differently)
    dev = KiwiAmd(path)
    texts = synthetic(sm, 1500, 924, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 500, 925) + fuzzed(sm, 300, 926)
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(ref.analyze(s)) == _norm(y), s
    dev.close()


def test_cong_fallback_paths_with_small_capacities(small_cong_model, monkeypatch):
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_model
    lib = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip_smallcaps.so")
    if not os.path.exists(lib):
        pytest.skip("libkiwi_hip_smallcaps.so not built (make -C kiwi_amd/csrc smallcaps)")
    monkeypatch.setenv("KAMD_CONTAINER_LIMITS", "3,8,2")
    orc = oraclelib.OracleKiwi(path)
    orc.set_container_limits(3, 8, 2)
    texts = synthetic(sm, 300, 927, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 150, 928)
    for lanes in ("pos", "16", "64"):
        force_lanes(monkeypatch, lanes)
        dev = KiwiAmd(path, lib_path=lib)      # (the switch is read when an engine is opened)
        got = dev.analyze_batch(texts).to_python()
        for s, y in zip(texts, got):
            assert _norm(orc.analyze(s)) == _norm(y), (lanes, s)
        dev.close()


def test_kiwi_init_selects_the_cong_model(small_cong_model):
    """kiwi_init on a container that carries a CoNgram blob: default options pick it (the reference's default model type), KNLM picks the Knlm."""
    import ctypes as C
    import oraclelib
    from test_gpu_capi import LIB, Option, MATCH_ALL_WITH_NORMALIZING
    sm, path = small_cong_model
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
    orc = oraclelib.OracleKiwi(path)                                   # CoNgram (the container has the blob)
    knlm = oraclelib.OracleKiwi(os.path.join(os.path.dirname(path), "small.raw"))   # same lexicon and Knlm, no blob
    texts = synthetic(sm, 40, 929, min_jamo=10, max_jamo=60)
    for options, want in ((15, orc), (15 | 0x0400, orc), (15 | 0x0200, knlm)):
        k = L.kiwi_init(path.encode(), 0, options, 0)
        assert k, L.kiwi_error()
        for s in texts:
            r = L.kiwi_analyze(k, s.encode("utf-8"), 1, opt, None)
            assert r, L.kiwi_error()
            assert L.kiwi_res_prob(r, 0) == want.analyze(s)[0][1], (options, s)
            L.kiwi_res_close(r)
        L.kiwi_close(k)
    assert not L.kiwi_init(path.encode(), 0, 15 | 0x0500, 0)


def test_cong_file_as_the_reference_builder_writes_it(mid_cong_vl4_model):
    """keySize 3 (two trie steps for LM ids >= 63488), 4-bit grouped embeddings, the global model's sections present (skipped): device vs oracle,
    and vs the real reference's SSE4.1 build where it travelled."""
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    sm, path = mid_cong_vl4_model
    orc = oraclelib.OracleKiwi(path)
    ref = refbridge.RefKiwi(path, arch=3, x86=True) if refbridge.x86_available() else None
    dev = KiwiAmd(path)
    texts = synthetic(sm, 1500, 927, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 500, 928) + EDGE_TEXTS + fuzzed(sm, 300, 929)
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s)) == _norm(y), s
        if ref is not None:
            assert _norm(ref.analyze(s)) == _norm(y), s
    dev.close()


@pytest.mark.parametrize("lanes,top_n,bias", [("16", 1, 0.0), ("64", 2, 2.5)])
def test_unknown_forms_scored_by_the_character_model(small_cong_chr_model, monkeypatch, lanes, top_n, bias):
    """Match::oovChrModel: k_unk_chr + the search kernel reading its scores, against the oracle and -- where it
    travelled -- the real reference (UnkFormScorer + CoNgramModel, SSE4.1 build) on the same nounchr.mdl."""
    import ctypes as C
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_chr_model
    force_lanes(monkeypatch, lanes)
    match = oraclelib.MATCH_ALL_WITH_NORMALIZING | (1 << 8)
    orc = oraclelib.OracleKiwi(path)
    orc.lib.korc_set_oov_chr_bias.argtypes = [C.c_void_p, C.c_float]
    orc.lib.korc_set_oov_chr_bias(orc.h, bias)
    ref = None
    if refbridge.x86_available() and top_n == 1:
        ref = refbridge.RefKiwi(path, arch=3, x86=True)
        ref.lib.kref_set_oov_chr_bias.argtypes = [C.c_void_p, C.c_float]
        ref.lib.kref_set_oov_chr_bias(ref.h, bias)
    dev = KiwiAmd(path)
    dev.set_oov_chr_bias(bias)
    texts = synthetic(sm, 1200, 931, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 500, 932) + EDGE_TEXTS + fuzzed(sm, 400, 933)
    got = dev.analyze_batch(texts, top_n=top_n, match=match).to_python()
    plain = dev.analyze_batch(texts, top_n=top_n).to_python()
    differ = 0
    for s, y, p in zip(texts, got, plain):
        assert _norm(orc.analyze(s, top_n=top_n, match=match)) == _norm(y), (lanes, top_n, s)
        if ref is not None:
            assert _norm(ref.analyze(s, match=match)) == _norm(y), s
        differ += _norm(y) != _norm(p)
    assert differ > 100
    dev.close()


@pytest.mark.parametrize("lanes,top_n,mode,bias,params", [("16", 1, 2, 0.0, None), ("64", 1, 3, 2.5, (60.0, 1.5, 1.0)), ("64", 2, 2, 1.0, None)])
def test_unknown_forms_scored_with_substring_frequencies(small_cong_chr_model, monkeypatch, lanes, top_n, mode, bias, params):
    """Match::oovChrFreqModel / oovChrFreqBranchModel: k_unk_chr_freq (substring counts of the filtered text + the character
    model, tanhf / expf / logf restated from glibc: csrc/chr_freq.hpp, exact_math.hpp) + the search reading its per-node scores, against the oracle and --
    where it travelled -- the real reference (UnkFormScorer + SubstringCounter + CoNgramModel, SSE4.1 build).  Long texts of many chunks count over the whole text."""
    import ctypes as C
    import oraclelib
    import refbridge
    from corpora import repeated_unknown_texts
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_chr_model
    force_lanes(monkeypatch, lanes)
    match = oraclelib.MATCH_ALL_WITH_NORMALIZING | (mode << 8)
    orc = oraclelib.OracleKiwi(path)
    orc.lib.korc_set_oov_chr_bias.argtypes = [C.c_void_p, C.c_float]
    orc.lib.korc_set_oov_chr_bias(orc.h, bias)
    orc.lib.korc_set_oov_freq_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    ref = None
    if refbridge.x86_available() and top_n == 1:
        ref = refbridge.RefKiwi(path, arch=3, x86=True)
        ref.lib.kref_set_oov_chr_bias.argtypes = [C.c_void_p, C.c_float]
        ref.lib.kref_set_oov_chr_bias(ref.h, bias)
        ref.lib.kref_set_oov_freq_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    dev = KiwiAmd(path)
    dev.set_oov_chr_bias(bias)
    if params:
        orc.lib.korc_set_oov_freq_params(orc.h, *params); dev.set_oov_freq_params(*params)
        if ref is not None:
            ref.lib.kref_set_oov_freq_params(ref.h, *params)
    try:
        texts = repeated_unknown_texts(sm, 1500, 941) + synthetic(sm, 300, 942, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 300, 943) + EDGE_TEXTS + fuzzed(sm, 300, 944)
        texts += [". ".join(texts[k:k + 25]) + "." for k in range(0, 200, 25)]      # several chunks per text
        texts += ["😀가나 😀가나 😀가나 ★다라★ ★다라★", "abc abc abc abcd abcd", "가" * 40 + " " + "가" * 40]
        got = dev.analyze_batch(texts, top_n=top_n, match=match).to_python()
        plain = dev.analyze_batch(texts, top_n=top_n, match=oraclelib.MATCH_ALL_WITH_NORMALIZING | (1 << 8)).to_python()
        differ = 0
        for s, y, p in zip(texts, got, plain):
            assert _norm(orc.analyze(s, top_n=top_n, match=match)) == _norm(y), (lanes, top_n, s)
            if ref is not None:
                assert _norm(ref.analyze(s, match=match)) == _norm(y), s
            differ += _norm(y) != _norm(p)
        assert differ > 100
    finally:
        if ref is not None:
            ref.lib.kref_set_oov_freq_params(ref.h, 35.0, 3.0, 4.0); ref.lib.kref_set_oov_chr_bias(ref.h, 0.0)
    dev.close()
