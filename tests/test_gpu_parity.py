"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs --
bit-exact lattices, tokens, positions and fp32 path scores -- plus the committed golden vectors produced by
the real reference, and size-independent properties at the benchmark batch size."""
import json
import os
from dataclasses import astuple

import numpy as np
import pytest

from corpora import EDGE_TEXTS, dictionary_mix, force_lanes, fuzzed, synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


def test_native_library_is_loaded(engine):
    maps = open("/proc/self/maps").read()
    assert "libkiwi_hip.so" in maps


def test_lattices_bit_exact_vs_oracle(engine, oracle, small_model):
    sm, _ = small_model
    for s in synthetic(sm, 150, 31, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 100, 32) + EDGE_TEXTS:
        if not s.strip():
            continue
        assert engine.split(s) == oracle.split(s), s


@pytest.mark.parametrize("kind", ["synthetic", "mix", "edge"])
def test_tokens_bit_exact_vs_oracle(engine, oracle, small_model, kind):
    sm, _ = small_model
    texts = {"synthetic": lambda: synthetic(sm, 2000, 41, min_jamo=5, max_jamo=200),
             "mix": lambda: dictionary_mix(sm, 1000, 42),
             "edge": lambda: EDGE_TEXTS}[kind]()
    got = engine.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(oracle.analyze(s)) == _norm(y), s


def test_fuzzed_texts_bit_exact_vs_oracle(engine, oracle, small_model):
    """Fuzzed inputs (URLs, e-mail, hashtags, emoji sequences, other scripts, jamo, lone surrogates, random code points): device
    lattices and analyses against the oracle (which the CPU suite checks against the real reference on the same generator)."""
    sm, _ = small_model
    texts = fuzzed(sm, 1500, 221)
    got = engine.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(oracle.analyze(s)) == _norm(y), repr(s)
    for s in texts[:150]:
        if s.strip():
            assert engine.split(s) == oracle.split(s), repr(s)


def test_golden_vectors_from_reference(engine):
    g = json.load(open(os.path.join(HERE, "golden", "small_model_golden.json"), encoding="utf-8"))
    texts = [it["text"] for it in g["items"]]
    got = engine.analyze_batch(texts).to_python()
    for it, y in zip(g["items"], got):
        toks = [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.sense_id, t.morph_id] for t in y[0][0]]
        assert toks == it["tokens"], it["text"]
        assert y[0][1] == it["score"], it["text"]


def test_against_real_reference_when_present(engine, reference, small_model):
    sm, _ = small_model
    texts = synthetic(sm, 500, 51, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 300, 52)
    got = engine.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(reference.analyze(s)) == _norm(y), s


def test_config_and_match_variants(engine, oracle, small_model):
    sm, _ = small_model
    texts = synthetic(sm, 200, 61, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 100, 62)
    try:
        for cfg in (dict(cut_off=5.0), dict(space_tol=2), dict(max_unk=3), dict(integrate_allomorph=False)):
            oracle.set_config(**cfg)
            engine.set_config(**cfg)
            got = engine.analyze_batch(texts).to_python()
            for s, y in zip(texts, got):
                assert _norm(oracle.analyze(s)) == _norm(y), (cfg, s)
        oracle.set_config()
        engine.set_config()
        for match in (0, (1 << 23) | (1 << 22), (1 << 23) | (1 << 25), (1 << 23) | (1 << 26), (1 << 23) | (1 << 17) | (1 << 18)):
            got = engine.analyze_batch(texts, match=match).to_python()
            for s, y in zip(texts, got):
                assert _norm(oracle.analyze(s, match=match)) == _norm(y), (match, s)
    finally:
        oracle.set_config()
        engine.set_config()


def test_batch_properties_at_benchmark_size(engine, small_model):
    """8k x 40-jamo batch (BASELINE config 2 shape): results do not depend on batch composition or order,
    tokens tile the text left to right, and re-running the staged batch is idempotent."""
    sm, _ = small_model
    texts = synthetic(sm, 8192, 71, exact_jamo=40)
    fields = [f for f in engine.analyze_batch(texts[:1]).token_array(0).dtype.names if f != "form_off"]   # form_off is batch-relative

    def same(a, b):
        return len(a) == len(b) and all((a[f] == b[f]).all() for f in fields)

    res = engine.analyze_batch(texts)
    arrs = [res.token_array(i) for i in range(len(texts))]
    scores = np.array([res.lib.kamd_res_prob(res.h, i, 0) for i in range(len(texts))], np.float32)
    for t, a in zip(texts, arrs):
        assert len(a) > 0
        pos = a["position"].astype(np.int64)
        assert (np.diff(pos) >= 0).all()
        assert int((pos + a["length"]).max()) <= len(t)
    perm = np.random.default_rng(5).permutation(len(texts))
    res2 = engine.analyze_batch([texts[i] for i in perm])
    for j, i in enumerate(perm[:2000]):
        assert same(arrs[i], res2.token_array(j))
    sub = engine.analyze_batch(texts[100:164])
    for j in range(64):
        assert same(arrs[100 + j], sub.token_array(j))
    b = engine.stage(texts)
    engine.run(b)
    engine.run(b)
    r3 = engine.fetch(b)
    s3 = np.array([r3.lib.kamd_res_prob(r3.h, i, 0) for i in range(len(texts))], np.float32)
    assert (s3 == scores).all()
    assert b.info()["chunks"] >= len(texts)


def test_lattice_hbm_kernel_matches_lds_kernel(engine, oracle, small_model, monkeypatch):
    """The wave-per-chunk lattice build (working set in LDS) and the thread-per-chunk build (HBM) it hands large chunks to
    produce the same lattices and tokens; KAMD_LATTICE_LDS=0 forces every chunk through the HBM kernel, a tiny budget mixes both."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    texts = synthetic(sm, 300, 91, min_jamo=5, max_jamo=200) + dictionary_mix(sm, 200, 92)
    want = engine.analyze_batch(texts).to_python()
    for budget in ("0", "6000"):
        monkeypatch.setenv("KAMD_LATTICE_LDS", budget)
        other = KiwiAmd(path)
        got = other.analyze_batch(texts).to_python()
        for s, x, y in zip(texts, want, got):
            assert _norm(x) == _norm(y), (budget, s)
        for s in texts[:40]:
            assert other.split(s) == oracle.split(s), (budget, s)
        other.close()
    monkeypatch.delenv("KAMD_LATTICE_LDS")


@pytest.mark.parametrize("lanes,wps", [("pos", "2"), ("pos", "3"), ("pos8", "2"), ("pos8", "3"), ("8", "3"), ("8", "2"), ("16", "3"), ("4", "2"), ("32", "2"), ("64", "2")])
def test_lane_group_variants_match_oracle(oracle, small_model, monkeypatch, lanes, wps):
    """Every instantiation of the search kernel (lanes per chunk x register budget) gives the oracle's result; large batches
    select 8 lanes / 3 waves per SIMD on their own (engine.hip), the others are reachable through KAMD_GROUP_LANES / KAMD_WPS."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    texts = synthetic(sm, 300, 121, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 150, 122) + EDGE_TEXTS
    force_lanes(monkeypatch, lanes)
    if lanes in ("8", "16", "pos", "pos8"):
        monkeypatch.setenv("KAMD_WPS", wps)
    other = KiwiAmd(path)
    for top_n in (1, 2):
        got = other.analyze_batch(texts, top_n=top_n).to_python()
        for s, y in zip(texts, got):
            assert _norm(oracle.analyze(s, top_n=top_n)) == _norm(y), (lanes, wps, top_n, s)
    other.close()


@pytest.mark.parametrize("lanes", ["pos", "pos8", "16", "8", "64"])
def test_fallback_paths_with_small_capacities(small_model, monkeypatch, lanes):
    """The synthetic lattices are too unambiguous to reach the medium / large path containers (> 128 / > 512 incoming paths),
    the HBM work-item queue (> 32 items of one candidate), the HBM pruning path (> 32 new paths of a node) or the far-back node
    lookup (> 32 nodes back) on their own.  This runs a build of the same kernels with those capacities cut to 4
    (`make smallcaps`) and the container limits cut to 3 / 8 incoming paths and 2 keys per bucket on BOTH sides, device and
    oracle (a separate oracle instance), and demands identical output -- for top-1 and top-2."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    lib = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip_smallcaps.so")
    if not os.path.exists(lib):
        pytest.skip("libkiwi_hip_smallcaps.so not built (make -C kiwi_amd/csrc smallcaps)")
    texts = synthetic(sm, 300, 141, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 200, 142) + EDGE_TEXTS
    force_lanes(monkeypatch, lanes)
    monkeypatch.setenv("KAMD_CONTAINER_LIMITS", "3,8,2")
    orc = oraclelib.OracleKiwi(path)
    orc.set_container_limits(3, 8, 2)
    dev = KiwiAmd(path, lib_path=lib)
    for top_n in (1, 2):
        got = dev.analyze_batch(texts, top_n=top_n).to_python()
        for s, y in zip(texts, got):
            assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (lanes, top_n, s)
    dev.close()


def test_overflow_rerun_ladder(oracle, small_model, monkeypatch):
    """Per-chunk HBM regions are sized linearly in the chunk length; a chunk that outgrows one reports an error code and
    the host re-runs it with 4x, 16x, 64x the capacity.  KAMD_TEST_TINY_ARENAS makes the regions far too small at scale 1, so
    most chunks climb that ladder; the results must not change."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    texts = synthetic(sm, 150, 151, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 100, 152)
    monkeypatch.setenv("KAMD_TEST_TINY_ARENAS", "1")
    dev = KiwiAmd(path)
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(oracle.analyze(s)) == _norm(y), s
    dev.close()


def test_empty_and_degenerate_batches(engine):
    assert engine.analyze_batch([]).n_texts() == 0
    r = engine.analyze_batch(["", " ", "\n"]).to_python()
    assert [len(x[0][0]) for x in r] == [0, 0, 0]


@pytest.mark.parametrize("top_n", [2, 3, 4, 8, 16])
def test_top_n_bit_exact_vs_oracle(engine, oracle, small_model, top_n):
    """top-N: the N best paths per key, N-th-best pruning, 2N end candidates -- same analyses, order and fp32 scores as the
    oracle (its hand-on order: DESIGN.md, top-N)."""
    sm, _ = small_model
    texts = synthetic(sm, 400, 101, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 300, 102) + EDGE_TEXTS
    got = engine.analyze_batch(texts, top_n=top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(oracle.analyze(s, top_n=top_n)) == _norm(y), s


@pytest.mark.parametrize("lanes", ["pos", "16", "64"])
def test_order_4_knlm_bit_exact_vs_oracle_and_reference(small_order4_model, monkeypatch, lanes):
    """An order-4 Knlm (order <= 4): contexts of three words, so a back-off chain is one node longer than the pair a search
    state carries (ModelView::lmChain) and the one-round-trip probe hands its tail to the general walk -- position-step kernel, 16- and 64-lane general
    kernels, top-1 and top-3, against the oracle and the real reference."""
    import oraclelib
    import refbridge
    from corpora import force_lanes
    from kiwi_amd.api import KiwiAmd
    sm, path = small_order4_model
    force_lanes(monkeypatch, lanes)
    orc = oraclelib.OracleKiwi(path)
    ref = refbridge.RefKiwi(path) if refbridge.available() else None
    dev = KiwiAmd(path)
    texts = synthetic(sm, 1500, 1101, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 500, 1102) + EDGE_TEXTS + fuzzed(sm, 300, 1103)
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s)) == _norm(y), (lanes, s)
        if ref is not None:
            assert _norm(ref.analyze(s)) == _norm(y), (lanes, s)
    got3 = dev.analyze_batch(texts[:600], top_n=3).to_python()
    for s, y in zip(texts[:600], got3):
        assert _norm(orc.analyze(s, top_n=3)) == _norm(y), (lanes, s)
    dev.close()


def test_top_n_beyond_the_device_limit_is_refused_loudly(engine):
    with pytest.raises(RuntimeError):
        engine.analyze_batch(["가나다"], top_n=17)


@pytest.mark.parametrize("model", ["knlm", "cong"])
def test_blocklist_bit_exact_vs_oracle(small_model, small_cong_model, model):
    """AnalyzeOption::blocklist (kamd_morphset_* + kamd_analyze_batch_opt): blocked candidates dropped by k_expand_cands -- device vs the oracle,
    whose blocklist path equals the real reference's (tests/test_oracle_vs_ref.py::test_blocklist_matches_reference)."""
    import oraclelib
    from corpora import pick_blocklist
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model if model == "knlm" else small_cong_model
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path)
    texts = synthetic(sm, 600, 951, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 300, 952) + EDGE_TEXTS
    items = pick_blocklist(orc, texts[:300], 32)
    ms, found = dev.morphset(items)
    assert found == orc.set_blocklist(items)
    for top_n in (1, 2):
        got = dev.analyze_batch_opt(texts, top_n=top_n, blocklist=ms).to_python()
        for s, y in zip(texts, got):
            assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (model, top_n, s)
    orc.set_blocklist([])
    plain = dev.analyze_batch(texts).to_python()
    assert sum(_norm(orc.analyze(s)) != _norm(y) for s, y in zip(texts[:50], plain[:50])) == 0
    ms.close(); dev.close()


def test_quantised_knlm_bit_exact_vs_oracle(small_quantised_model):
    """A model whose sj.knlm is quantised / compressed as the reference's builder writes it: device vs oracle (= the real reference on that blob)."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path, _ = small_quantised_model
    orc, dev = oraclelib.OracleKiwi(path), KiwiAmd(path)
    texts = synthetic(sm, 800, 981, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 400, 982) + EDGE_TEXTS
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s)) == _norm(y), s
    dev.close()
