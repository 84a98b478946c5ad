"""The GLOBAL CoNgram model (ModelType::congGlobal, window 7: a valid distant token is scored as a mixture over the context and the last seven such tokens of
the path; src/CoNgramModel.cpp:802-868, 1037-1490) -- oracle side: the restatement (csrc/cong_global.hpp: the shared arithmetic;
oracle/viterbi_oracle.hpp: candidate order, matrix shapes, state equality and hash, the path containers) against the REAL reference's SSE4.1 build with
the same cong.mdl loaded with useDistantTokens = true, and against golden analyses of it (tests/golden/cong_global_*.json, tools/make_golden_cong_global.py)
where the reference library is absent.  32-bit and 16-bit ids: the reference's state hash reads 8 BYTES of the history.

What the pin found in the reference and the oracle reproduces (without it 1 sentence in ~400 differs): once a bucket of the path container holds 64
entries, nst::findAll<sse4_1> returns nothing for its first half (a shift count of 64), and candidates of the second half are compared with the entry of
the FIRST half at the same offset (BestPathContainer.hpp:341-350).

Device side (round 5: viterbi_kernel_congg.hip / viterbi_kernel_congg_typo.hip, kamd_open_mode(lm_mode = 4), kiwi_init with CONG_GLOBAL or LARGEST): the whole
analysis through the lane emulator here -- golden analyses of the real reference (top-1, top-3, open ending, typo correction), random sentences against the
oracle, the replay of the container past 64 entries --, and on the MI355X by tests/test_gpu_cong_global.py with the same checkers.  kamd_debug_cong_global
evaluates the mixture arithmetic alone on the device over the window sections in HBM (tests/test_zzz_gpu_cong_global_probe.py)."""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _model(which):
    from kiwi_amd import synth
    spec = synth.SMALL_CONG_GLOBAL_SPEC if which == "32" else synth.SMALL_CONG_GLOBAL16_SPEC
    path = os.path.join(ROOT, "_data", f"small-cong-global{which}.raw")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    sm = synth.SynthModel(spec)
    sm.raw.save(path)
    return sm, path


def _rows(res):
    return [[[[t.form, t.tag, t.position, t.length, t.score, t.typo_cost] for t in toks], score] for toks, score in res]


def _oracle(path):
    import oraclelib
    orc = oraclelib.OracleKiwi(path)
    orc.set_cong_global(True)
    return orc


def _reference(path):
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    return refbridge.RefKiwi(path, arch=3, model_dir_sbg="cong_global", x86=True)


def test_window_sections_are_optional():
    """The same file scored locally (ModelType::cong) unless the global model is asked for; a file without the sections cannot be."""
    import oraclelib
    from kiwi_amd import synth
    _, path = _model("32")
    a = oraclelib.OracleKiwi(path)
    b = _oracle(path)
    t = "가나다라 마바사 아자차카 타파하"
    ra, rb = a.analyze(t), b.analyze(t)
    assert a.counters()["congGlobalScores"] == 0 and b.counters()["congGlobalScores"] > 0
    assert ra[0][1] != rb[0][1]
    p0 = os.path.join(ROOT, "_data", "small-cong.raw")
    synth.SynthModel(synth.SMALL_CONG_SPEC).raw.save(p0)
    with pytest.raises(RuntimeError):
        oraclelib.OracleKiwi(p0).set_cong_global(True)


def test_mixture_kernels_equal_the_reference_bit_for_bit():
    """logSoftmax / logSumExp over 8 terms, packet form (progress()) and per-lane form (progressMatrix*), against lm::logSoftmax<sse4_1> ... of the
    reference's own SSE4.1 translation unit: Cephes exp / log polynomials without FMA, the horizontal sums, libm's log in one place."""
    import ctypes as C
    import oraclelib
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    ref = C.CDLL(refbridge.LIB_PATH); orc = C.CDLL(oraclelib.LIB_PATH)
    for lib, fn in ((ref, "kref_congg_math"), (orc, "korc_congg_math")):
        getattr(lib, fn).restype = C.c_float
        getattr(lib, fn).argtypes = [C.c_int, C.c_void_p]
    rng = np.random.default_rng(11)
    n = 0
    for trial in range(4000):
        w = rng.uniform(-12, 2, 8).astype(np.float32)
        if trial % 3 == 0:
            w[rng.integers(1, 8, rng.integers(1, 7))] = np.float32(-99999.0) + rng.uniform(-1, 1)      # empty history slots
        if trial % 7 == 0:
            w = (w * 8).astype(np.float32)
        for which in range(4):
            a, b = w.copy(), w.copy()
            ra = ref.kref_congg_math(which, a.ctypes.data)
            rb = orc.korc_congg_math(which, b.ctypes.data)
            assert a.tobytes() == b.tobytes() and struct.pack("f", ra) == struct.pack("f", rb), (which, w.tolist())
            n += 1
    assert n == 16000


@pytest.mark.parametrize("which", ["32", "16"])
def test_oracle_equals_reference(which):
    """Top-1 on long sentences (containers beyond 64 entries per bucket), top-3, open ending, typo correction."""
    import random
    import oraclelib
    import refbridge
    from corpora import EDGE_TEXTS, dictionary_mix, synthetic
    from typo_cases import misspell
    sm, path = _model(which)
    ref, orc = _reference(path), _oracle(path)
    texts = synthetic(sm, 260, 941, min_jamo=5, max_jamo=140) + dictionary_mix(sm, 80, 942) + EDGE_TEXTS
    for t in texts:
        if t.strip():
            assert _rows(ref.analyze(t)) == _rows(orc.analyze(t)), t
    c = orc.counters()
    assert c["congGlobalScores"] > 100000 and c["nodesOver128"] > 50
    for t in texts[:60]:
        if t.strip():
            assert _rows(ref.analyze(t, top_n=3)) == _rows(orc.analyze(t, top_n=3)), t
            assert _rows(ref.analyze(t, open_ending=True)) == _rows(orc.analyze(t, open_ending=True)), t
    ents, cont, leng = refbridge.default_typo_entries("basic_with_continual")
    rt = refbridge.RefTypo(); rt.update_default("basic_with_continual"); rt.prepare(True)
    ot = oraclelib.OracleTypo(); ot.update_entries(ents, cont, leng); ot.prepare(True)
    rnd = random.Random(3)
    corrected = 0
    for t in [misspell(t, rnd, True, True, False) for t in texts[:70]]:
        if t.strip():
            a = ref.analyze_typo(rt, t, 2.5, 0)
            assert _rows(a) == _rows(orc.analyze_typo(ot, t, 2.5, 0)), t
            corrected += any(x.typo_cost > 0 for x in a[0][0])
    assert corrected >= 30


@pytest.mark.parametrize("which", ["32", "16"])
def test_oracle_equals_golden(which):
    g = json.load(open(os.path.join(HERE, "golden", f"cong_global_{which}.json"), encoding="utf-8"))
    _, path = _model(which)
    orc = _oracle(path)
    assert len(g["top1"]) == 161
    for it in g["top1"]:
        assert _rows(orc.analyze(it["text"])) == it["res"], it["text"]
    for it in g["top3"]:
        assert _rows(orc.analyze(it["text"], top_n=3)) == it["res"], it["text"]
    for it in g["open_ending"]:
        assert _rows(orc.analyze(it["text"], open_ending=True)) == it["res"], it["text"]
    import refbridge
    if refbridge.available():      # (the built-in typo set is read out of oracle/_ref)
        import oraclelib
        ents, cont, leng = refbridge.default_typo_entries("basic_with_continual")
        ot = oraclelib.OracleTypo(); ot.update_entries(ents, cont, leng); ot.prepare(True)
        for it in g["typo"]:
            assert _rows(orc.analyze_typo(ot, it["text"], 2.5, 0)) == it["res"], it["text"]


def test_the_sentence_that_needs_the_container_defects():
    """tests/golden/cong_global_32.json item 0: the reference keeps a lower-scored duplicate of an 'equal' state alive in a bucket of > 64 entries, and the
    duplicate's other history words win the sentence later; the oracle answers the same only because it reproduces that."""
    g = json.load(open(os.path.join(HERE, "golden", "cong_global_32.json"), encoding="utf-8"))
    _, path = _model("32")
    it = g["top1"][0]
    got = _rows(_oracle(path).analyze(it["text"]))
    assert got == it["res"]
    assert [t[0] for t in got[0][0]][5:7] == ["소갸", "탸"] and abs(got[0][1] - (-89.72936248779297)) < 1e-6


def probe_device_arithmetic(lib_path):
    """kamd_debug_cong_global (the device evaluating csrc/cong_global.hpp over the window sections it uploaded) against the host evaluation of the same
    header inside the oracle -- which the analyses above pin to the real reference: 24 000 random (context, history, next, kind) queries, bit for bit."""
    import ctypes as C
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    n_ok = 0
    for which in ("32", "16"):
        sm, path = _model(which)
        orc = _oracle(path)
        dev = KiwiAmd(path, lib_path=lib_path) if lib_path else KiwiAmd(path)
        L = dev.lib
        L.kamd_debug_cong_global.argtypes = [C.c_void_p] * 6 + [C.c_uint32]
        O = orc.lib
        O.korc_congg_scores.argtypes = [C.c_void_p] * 6 + [C.c_uint32]
        from kiwi_amd.container import read_container
        blob = read_container(path)[1]["cong"]
        vocab, n_ctx = struct.unpack_from("<QQ", blob.tobytes(), 0)
        rng = np.random.default_rng(5)
        n = 12000
        ctx = rng.integers(0, n_ctx, n, dtype=np.uint32)
        nxt = rng.integers(0, vocab, n, dtype=np.uint32)
        hist = rng.integers(1, vocab, (n, 7), dtype=np.uint32)
        hist[rng.random((n, 7)) < 0.35] = 0                      # empty slots
        hist[: n // 10] = 0                                       # ... and whole empty histories
        flags = rng.integers(0, 4, n, dtype=np.uint8)
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        assert L.kamd_debug_cong_global(dev.h, ctx.ctypes.data, hist.ctypes.data, nxt.ctypes.data, flags.ctypes.data, a.ctypes.data, n) == 0
        assert O.korc_congg_scores(orc.h, ctx.ctypes.data, hist.ctypes.data, nxt.ctypes.data, flags.ctypes.data, b.ctypes.data, n) == 0
        assert a.tobytes() == b.tobytes(), int((a.view(np.uint32) != b.view(np.uint32)).sum())
        assert np.isfinite(a).all() and len(set(a.tolist())) > n // 2
        dev.close()
        n_ok += n
    return n_ok


def test_emulated_device_arithmetic_equals_the_oracle():
    import subprocess
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    assert probe_device_arithmetic(os.path.join(emu, "_build", "libkiwi_hipemu.so")) == 24000



def check_device_goldens(lib_path, which, sections=("top1", "top3", "open_ending", "typo"), limit=None):
    """The device (lib_path: the emulated library; None: the product library on a GPU) with the global model against the golden analyses of the real reference
    (tests/golden/cong_global_*.json; item 0 of the 32-bit file is the sentence that needs the container replay).  Returns the number of analyses compared."""
    from kiwi_amd.api import KiwiAmd, Typo
    _, path = _model(which)
    g = json.load(open(os.path.join(HERE, "golden", f"cong_global_{which}.json"), encoding="utf-8"))
    dev = KiwiAmd(path, lib_path=lib_path, lm_mode=4) if lib_path else KiwiAmd(path, lm_mode=4)
    n = 0
    for sec in sections:
        items = g[sec][:limit]
        texts = [it["text"] for it in items]
        if sec == "typo":
            ty = Typo.from_default(dev.lib, 3).prepare(True)      # basicTypoSetWithContinual, as tools/make_golden_cong_global.py asked the reference for
            got = dev.analyze_batch_opt(texts, typo=ty, typo_threshold=2.5).to_python()
            ty.close()
        else:
            got = dev.analyze_batch(texts, top_n=3 if sec == "top3" else 1, open_ending=sec == "open_ending").to_python()
        for it, y in zip(items, got):
            assert _rows(y) == it["res"], (which, sec, it["text"])
        n += len(items)
    dev.close()
    return n


def check_device_vs_oracle(lib_path, which, n_random, n_top3, max_jamo=140, seed=941, extra=()):
    """Random sentences: device == oracle (top-1 on all, top-3 on the first n_top3).  The corpus must reach the container replay many times (the oracle counts
    the insertions past 64 entries).  Returns (sentences, insertions past 64 entries)."""
    import sys
    sys.path.insert(0, HERE)
    from corpora import dictionary_mix, synthetic
    from kiwi_amd.api import KiwiAmd
    sm, path = _model(which)
    orc = _oracle(path)
    dev = KiwiAmd(path, lib_path=lib_path, lm_mode=4) if lib_path else KiwiAmd(path, lm_mode=4)
    texts = [t for t in synthetic(sm, n_random, seed, min_jamo=5, max_jamo=max_jamo) + dictionary_mix(sm, n_random // 4, seed + 1) + list(extra) if t.strip()]
    orc.counters(reset=True)
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _rows(orc.analyze(s)) == _rows(y), (which, s)
    c = orc.counters()
    assert c["congGlobalScores"] > 0
    got = dev.analyze_batch(texts[:n_top3], top_n=3).to_python()
    for s, y in zip(texts[:n_top3], got):
        assert _rows(orc.analyze(s, top_n=3)) == _rows(y), (which, s)
    dev.close()
    return len(texts), c["congPast64"]


@pytest.fixture(scope="module")
def emu_lib():
    import subprocess
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    return os.path.join(emu, "_build", "libkiwi_hipemu.so")


@pytest.mark.parametrize("which,lanes", [("32", "64"), ("16", "16")])
def test_emulated_device_equals_the_golden_analyses_of_the_reference(emu_lib, monkeypatch, which, lanes):
    """viterbi_kernel_congg.hip (+ _typo) compiled for the host: every golden analysis of the real reference's global model, 64-lane and 16-lane groups."""
    monkeypatch.setenv("KAMD_GROUP_LANES", lanes)
    assert check_device_goldens(emu_lib, which) == 271


def test_emulated_device_equals_the_oracle_on_random_sentences(emu_lib, monkeypatch):
    monkeypatch.setenv("KAMD_GROUP_LANES", "16")
    n, past64 = check_device_vs_oracle(emu_lib, "32", 60, 12)
    assert n >= 70 and past64 > 1000, (n, past64)


def test_emulated_device_with_small_capacities(monkeypatch):
    """The `make smallcaps` configuration (tiny LDS staging capacities, constant history digest: every equal-key pair reaches the exact comparison of the
    history words) on the first golden sentences -- the one that needs the container replay among them."""
    import subprocess
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "smallcaps", "-j8"], stdout=subprocess.DEVNULL)
    assert check_device_goldens(os.path.join(emu, "_build", "libkiwi_hipemu_smallcaps.so"), "32", sections=("top1", "top3", "typo"), limit=24) == 72
