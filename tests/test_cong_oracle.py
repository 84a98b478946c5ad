"""CPU: CoNgram scoring, oracle side.  The REAL src/CoNgramModel.cpp (+ the SIMD architecture
translation units, which alone carry the reference's QUANTISED CoNgram path) compiled into oracle/_ref/libkiwi_ref_x86.so over functional
stand-ins for the two empty submodules it includes (oracle/standin: Eigen, streamvbyte), loading a synthetic cong.mdl in the reference's own
layout (kiwi_amd/synth.py build_cong), pins this repo's restatement (oracle/viterbi_oracle.hpp congContext / congNext / evalCong,
kiwi_amd/csrc/model.cpp loadCong): single LM steps and whole analyses, tokens and fp32 scores bit for bit.  Pin = the reference's SSE4.1
build; its AVX2 build rounds the batched scores differently (asserted below, so that nobody mistakes the pin for arch-independent)."""
import ctypes as C
import os
import random
from dataclasses import astuple

import pytest

from corpora import EDGE_TEXTS, dictionary_mix, fuzzed, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


@pytest.fixture(scope="module")
def cong_pair(small_cong_model):
    import oraclelib
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built (make -C oracle refx86; needs /root/reference)")
    sm, path = small_cong_model
    return sm, refbridge.RefKiwi(path, arch=3, x86=True), oraclelib.OracleKiwi(path), path


@pytest.fixture(scope="module")
def cong_vl4_pair(mid_cong_vl4_model):
    import oraclelib
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built (make -C oracle refx86; needs /root/reference)")
    sm, path = mid_cong_vl4_model
    return sm, refbridge.RefKiwi(path, arch=3, x86=True), oraclelib.OracleKiwi(path), path


def test_cong_file_as_the_builder_writes_it_for_a_large_vocabulary(cong_vl4_pair):
    """cong.mdl with keySize 3 (a word id >= 63488 is two 16-bit trie keys: CoNgramModel::progressContextNode), qbit 4 (nibble pairs + local
    scale / zero point per group, requantised to int8 at load time by requantizePackedU4 -- the SSE4.1 build's rounding) and the sections of the
    global model present (skipped in local mode): LM steps over random walks incl. ids on both sides of the two-key boundary, then whole analyses."""
    sm, ref, orc, _ = cong_vl4_pair
    ref.lib.kref_cong_next.restype = C.c_float
    ref.lib.kref_cong_next.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_uint32]
    orc.lib.korc_cong_next.restype = C.c_float
    orc.lib.korc_cong_next.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_uint32]
    rnd = random.Random(6)
    vocab = sm.raw.vocab_size
    assert vocab > 63488 + 1024
    seen_big = 0
    for _ in range(1500):
        n1, c1, n2, c2 = C.c_int32(0), C.c_uint32(0), C.c_int32(0), C.c_uint32(0)
        for _ in range(8):
            r = rnd.random()
            w = rnd.randrange(60000, vocab) if r < 0.6 else rnd.randrange(63400, 63600) if r < 0.7 else rnd.randrange(0, 200)
            a = ref.lib.kref_cong_next(ref.h, C.byref(n1), C.byref(c1), w)
            b = orc.lib.korc_cong_next(orc.h, C.byref(n2), C.byref(c2), w)
            assert (a, n1.value, c1.value) == (b, n2.value, c2.value), w
            seen_big += w >= 63488 and c1.value != 0
    assert seen_big > 200
    texts = synthetic(sm, 500, 841, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 300, 842) + EDGE_TEXTS + fuzzed(sm, 200, 843)
    for t in texts:
        assert _norm(ref.analyze(t)) == _norm(orc.analyze(t)), repr(t)


def test_cong_lm_steps_equal_reference(cong_pair):
    """CoNgramModel::progress (score in the current context, then progressContextNode) over random walks: log-likelihood bits, node, context id."""
    sm, ref, orc, _ = cong_pair
    ref.lib.kref_cong_next.restype = C.c_float
    ref.lib.kref_cong_next.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_uint32]
    orc.lib.korc_cong_next.restype = C.c_float
    orc.lib.korc_cong_next.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_uint32]
    rnd = random.Random(5)
    vocab = sm.raw.vocab_size
    for _ in range(1500):
        n1, c1, n2, c2 = C.c_int32(0), C.c_uint32(0), C.c_int32(0), C.c_uint32(0)
        for _ in range(8):
            w = rnd.randrange(0, vocab) if rnd.random() < 0.5 else rnd.randrange(0, 200)
            a = ref.lib.kref_cong_next(ref.h, C.byref(n1), C.byref(c1), w)
            b = orc.lib.korc_cong_next(orc.h, C.byref(n2), C.byref(c2), w)
            assert (a, n1.value, c1.value) == (b, n2.value, c2.value)


def test_cong_analyses_equal_reference(cong_pair):
    """Whole analyses through the transposed PathEvaluator + MorphemeEvaluator<CoNgramState> (regular candidates against the score matrix, then
    the halves of split stems; z-coda shortcuts first), including which kernel of the reference rounds each score."""
    sm, ref, orc, _ = cong_pair
    texts = synthetic(sm, 700, 831, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 400, 832) + EDGE_TEXTS + fuzzed(sm, 400, 833)
    for t in texts:
        assert _norm(ref.analyze(t)) == _norm(orc.analyze(t)), repr(t)


def test_the_pin_is_the_sse41_build(cong_pair):
    """The reference's AVX2 kernels round the batched scores in another order (src/archImpl/avx2_qgemm.hpp): same tokens and near-equal scores,
    but not the same bits -- the reference is not bit-reproducible across its own architectures for this model type."""
    import refbridge
    sm, ref, _, path = cong_pair
    avx2 = refbridge.RefKiwi(path, arch=4, x86=True)
    texts = synthetic(sm, 200, 834, min_jamo=20, max_jamo=100)
    differ = 0
    for t in texts:
        a, b = ref.analyze(t), avx2.analyze(t)
        assert abs(a[0][1] - b[0][1]) < 1e-2 * max(1.0, abs(a[0][1]))
        differ += _norm(a) != _norm(b)
    assert differ > 0


@pytest.mark.parametrize("name,threshold,carry", [("basic_with_continual", 2.5, True), ("basic_with_continual_and_lengthening", 4.0, True)])
def test_typo_correction_with_a_cong_model_equals_reference(cong_pair, name, threshold, carry):
    """The reference's default model type together with its --typo configurations: Kiwi::analyze with a typo transformer on a CoNgram model,
    SSE4.1 build of the real reference vs the oracle -- tokens, positions, fp32 scores, per-token typo costs."""
    import oraclelib
    import refbridge
    from typo_cases import misspell
    sm, ref, orc, _ = cong_pair
    ents, cont, leng = refbridge.default_typo_entries(name)
    rt = refbridge.RefTypo(); rt.update_default(name); rt.prepare(True)
    ot = oraclelib.OracleTypo(); ot.update_entries(ents, cont, leng); ot.prepare(True)
    rnd = random.Random(15)
    tt = [misspell(t, rnd, True, carry, "lengthening" in name) for t in synthetic(sm, 70, 731, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 40, 732)] + EDGE_TEXTS[:30]
    corrected = 0
    for t in tt:
        if not t.strip():
            continue
        a = ref.analyze_typo(rt, t, threshold, 0)
        assert _norm(a) == _norm(orc.analyze_typo(ot, t, threshold, 0)), t
        corrected += any(x.typo_cost > 0 for x in a[0][0])
    assert corrected >= 15
