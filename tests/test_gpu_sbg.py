"""GPU (-m gpu): the search kernel for SkipBigram models (kiwi_amd/csrc/viterbi_kernel_sbg.hip) against the CPU oracle, whose
SkipBigram path is pinned to the real reference by tests/test_oracle_vs_ref.py.

First run on an MI355X in round 2 (profiles/r02_a_*): all green at first contact; the experimental gate is gone."""
import os
from dataclasses import astuple

import pytest

from corpora import EDGE_TEXTS, dictionary_mix, synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


def _texts(sm, seed, small=False):
    if small:      # the small-capacity build sends every batch through the staged (HBM scratch) path: minutes per hundred long texts
        return synthetic(sm, 60, seed, min_jamo=5, max_jamo=70) + dictionary_mix(sm, 40, seed + 1) + EDGE_TEXTS[:40]
    return synthetic(sm, 300, seed, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 150, seed + 1) + EDGE_TEXTS


@pytest.mark.parametrize("lanes", ["16", "64"])
@pytest.mark.parametrize("top_n", [1, 2, 3])
def test_skipbigram_tokens_bit_exact_vs_oracle(small_sbg_model, monkeypatch, lanes, top_n):
    """Tokens, positions and fp32 scores under Knlm + skip-bigram mixture scoring: history rings in the LM state (container keys
    compare the whole ring for top-1, the last four words and no root for top-N), glibc-exact exp / log on the device."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_sbg_model
    monkeypatch.setenv("KAMD_GROUP_LANES", lanes)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path)
    texts = _texts(sm, 301)
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (lanes, top_n, s)
    dev.close()


@pytest.mark.parametrize("lanes", ["16", "64"])
def test_skipbigram_fallback_paths_with_small_capacities(small_sbg_model, monkeypatch, lanes):
    """The `make smallcaps` build: LDS capacities of 4, container limits 3 / 8 / 2 on both sides (medium container: the bucket
    hash chains the ring words), and a constant ring digest, so that every pair of items with equal packed keys reaches the
    exact ring comparison and the register path hands colliding batches over to the scanning path."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_sbg_model
    lib = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip_smallcaps.so")
    if not os.path.exists(lib):
        pytest.skip("libkiwi_hip_smallcaps.so not built (make -C kiwi_amd/csrc smallcaps)")
    monkeypatch.setenv("KAMD_GROUP_LANES", lanes)
    monkeypatch.setenv("KAMD_CONTAINER_LIMITS", "3,8,2")
    orc = oraclelib.OracleKiwi(path)
    orc.set_container_limits(3, 8, 2)
    dev = KiwiAmd(path, lib_path=lib)
    texts = _texts(sm, 311, small=True)
    for top_n in (1, 2):
        got = dev.analyze_batch(texts, top_n=top_n).to_python()
        for s, y in zip(texts, got):
            assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (lanes, top_n, s)
    dev.close()


def test_skipbigram_golden_sequence_from_reference(small_sbg_model):
    """The committed analyses of the real reference under the SkipBigram model (tests/golden/small_sbg_model_sequence.json, top-1):
    where the reference's large container decided the order of equal-score paths the device may differ (top-N /
    container order); everything else must be identical, so at least the scores of the best analysis are compared exactly."""
    import json
    from kiwi_amd.api import KiwiAmd
    g = json.load(open(os.path.join(HERE, "golden", "small_sbg_model_sequence.json"), encoding="utf-8"))
    dev = KiwiAmd(small_sbg_model[1])
    texts = [it["text"] for it in g["items"]]
    got = dev.analyze_batch(texts, top_n=g.get("top_n", 1)).to_python()
    same = sum(1 for it, y in zip(g["items"], got) if y and y[0][1] == it["analyses"][0]["score"])
    assert same >= 0.99 * len(texts), (same, len(texts))
    dev.close()


@pytest.mark.parametrize("model,lanes,top_n,pool", [("sbg", "64", 1, "4096"), ("sbg", "64", 3, "4096"), ("sbg", "64", 2, "6"), ("knlm", "pos", 1, "2048"), ("knlm", "16", 2, "2048")])
def test_state_arenas_grow_into_the_pool(small_model, small_sbg_model, monkeypatch, model, lanes, top_n, pool):
    """Arenas of 1/64 of the worst case (KAMD_STATE_SCALE=1): most chunks fill theirs and carry on in arenas from the batch's pool (growArena: one atomic add on
    the pool's counter, the states so far move, the node that met the full arena is evaluated again; the end stage grows the same way) -- same analyses as the
    oracle's.  A pool of 6/64 of the arenas runs out: the chunks it could not serve go through the re-run ladder.  SkipBigram: arenas of the lane groups, the end
    stage inside the search kernel (finishPathsSolo)."""
    import oraclelib
    from corpora import force_lanes
    from kiwi_amd.api import KiwiAmd
    sm, path = small_sbg_model if model == "sbg" else small_model
    force_lanes(monkeypatch, lanes)
    monkeypatch.setenv("KAMD_STATE_SCALE", "1")
    monkeypatch.setenv("KAMD_STATE_POOL", pool)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path)
    texts = [t for t in synthetic(sm, 400, 571, min_jamo=20, max_jamo=120) + dictionary_mix(sm, 100, 572) if t.strip()]
    b = dev.stage(texts)
    got = dev.fetch(b, top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), s
    p = b.pool()
    reruns, _ = dev.reruns(b)
    assert p["pool_states"] >= p["arena_states"] * int(pool) // 64 and p["pool_asked"] > 0, p
    if pool == "6":      # (a pool this small: chunks it cannot serve end with an overflow status and are re-run)
        assert reruns > 0 or p["pool_asked"] <= p["pool_states"], (p, reruns)
    else:
        assert p["pool_asked"] > p["arena_states"] // 2 and p["pool_asked"] <= p["pool_states"] and (reruns == 0 or model == "knlm"), (p, reruns)
    b.close()
    dev.close()
