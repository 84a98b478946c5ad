"""SURVEY.md section 8 row a4, first half: the typo graph (TypoTransformer rule container -> prepare() -> generateGraph).
oracle/typo_oracle.hpp is a CPU restatement; here it is pinned, byte for byte, to the REAL reference translation unit
(src/TypoTransformer.cpp through oracle/ref_bridge.cpp), to the reference's own known-answer test, and to committed graphs
generated from the reference (tests/golden/typo_graphs.json).  Second half: the lattice built OVER such a graph
(oracle/typo_lattice_oracle.hpp: per-branch search states, typo costs, continual-typo positions) and the whole analysis with a typo
transformer -- lattices, tokens, fp32 scores and typo costs equal the reference's (kref_split_typo / kref_analyze_typo) on misspelt
texts -- including lengthening typos.  Nothing of this is on the device yet."""
import json
import os

import numpy as np
import pytest

from typo_cases import COND, HAND_TEXTS, INF, fill, texts

HERE = os.path.dirname(os.path.abspath(__file__))


def _ref_bytes(ref, t, dialect, norm_coda=True):
    import refbridge
    u = refbridge._u16(t)
    need = ref.lib.kref_typo_graph(ref.h, u.ctypes.data, len(u), dialect, int(norm_coda), None, 0)
    buf = np.zeros(need, np.uint8)
    ref.lib.kref_typo_graph(ref.h, u.ctypes.data, len(u), dialect, int(norm_coda), buf.ctypes.data, need)
    return buf.tobytes()


def test_reference_known_answer_generate_graph():
    """test/test_typo.cpp:8-29 (KiwiTypo.GenerateGraph): three rules, inverse preparation, 11 graph nodes."""
    import oraclelib
    import refbridge
    t = oraclelib.OracleTypo()
    t.add("ㅐ", "ㅚ"); t.add("레", "뢰"); t.add("뢨", "룄")
    t.prepare(True)
    norm, nodes, _ = refbridge.parse_typo_graph(t.graph_bytes("그럼 내괴다룄네", 0, norm_coda=False))
    assert len(nodes) == 11
    assert nodes[0][0] == "" and nodes[-1][1] == len(norm)
    fixes = [n for n in nodes if n[2] > 0]      # the corrections it offers: 괴->개, 뢰->레, 뢰->래, 룄->뢨, one typo each
    assert [n[0][0] for n in fixes] == ["개", "레", "래", "뢔"] and all(n[2] == 1.0 for n in fixes)


@pytest.mark.parametrize("inverse", [True, False])
def test_own_rules_graphs_equal_reference(inverse):
    """Rule by rule through addTypo on both sides (normalisation, jamo expansion, the 14 applosive variants, min-cost merging), then
    prepare(inverse) and generateGraph: the byte dumps of the graphs must be identical -- node order, links, costs, continual indices,
    dialect tags, maxContinualTypoIdx -- with the continual / lengthening costs off and on and for three dialect masks."""
    import oraclelib
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    tt = texts(150, 17 + inverse)
    for cont, leng in ((INF, INF), (1.0, 0.25)):
        ref = refbridge.RefTypo(cont, leng); fill(ref, True); ref.prepare(inverse)
        orc = oraclelib.OracleTypo(cont, leng); fill(orc, False); orc.prepare(inverse)
        for dia in (0, 8, 0xFFFF):
            for t in tt:
                assert _ref_bytes(ref, t, dia) == orc.graph_bytes(t, dia, True), (inverse, cont, dia, t)


@pytest.mark.parametrize("name", ["basic", "continual", "basic_with_continual", "basic_with_continual_and_lengthening", "dialect"])
def test_builtin_sets_graphs_equal_reference(name):
    """The reference's built-in typo sets (up to 12842 expanded rules), replayed entry by entry in the iteration order of the
    reference's map on both sides (TypoTransformer::update): prepare() walks the rule map in iteration order, so this also pins the
    restated hash of the rule key and the order of the replacements inside every pattern."""
    import oraclelib
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    from corpora import EDGE_TEXTS
    ents, cont, leng = refbridge.default_typo_entries(name)
    ref = refbridge.RefTypo(); ref.update_default(name); ref.prepare(True)
    orc = oraclelib.OracleTypo(); orc.update_entries(ents, cont, leng); orc.prepare(True)
    dia = 0xFFFF if name == "dialect" else 0
    n_nodes = 0
    for t in texts(120, 29) + EDGE_TEXTS:
        rb = _ref_bytes(ref, t, dia)
        assert rb == orc.graph_bytes(t, dia, True), (name, t)
        n_nodes += len(refbridge.parse_typo_graph(rb)[1])
    assert n_nodes > 3000


def test_golden_typo_graphs():
    """Graphs of the real reference for this repo's test rules (tools/make_golden.py), replayed without oracle/_ref."""
    import oraclelib
    g = json.load(open(os.path.join(HERE, "golden", "typo_graphs.json"), encoding="utf-8"))
    for case in g["cases"]:
        orc = oraclelib.OracleTypo(case["continual"] if case["continual"] is not None else INF, case["lengthening"] if case["lengthening"] is not None else INF)
        fill(orc, False)
        orc.prepare(case["inverse"])
        for it in case["items"]:
            assert orc.graph_bytes(it["text"], case["dialect"], True).hex() == it["graph"], (case["inverse"], it["text"])


def _norm(res):
    from dataclasses import astuple
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


@pytest.mark.parametrize("name,threshold,carry", [("basic", 2.5, False), ("basic", 6.0, False), ("continual", 2.5, True), ("basic_with_continual", 2.5, True),
                                                   ("basic_with_continual", 1.2, True), ("dialect", 2.5, False), ("lengthening", 2.5, False),
                                                   ("basic_with_continual_and_lengthening", 2.5, True), ("basic_with_continual_and_lengthening", 4.0, True)])
def test_typo_lattices_and_analyses_equal_reference(small_model, name, threshold, carry):
    """Misspelt texts of the synthetic model (confusable vowels, codas carried over to the next syllable) through Kiwi::analyze with a
    typo transformer, reference vs oracle: the lattice of every chunk (nodes, links, typo costs) and the analyses (tokens, positions,
    fp32 scores, per-token typo costs)."""
    import random
    import oraclelib
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    from corpora import EDGE_TEXTS, dictionary_mix, synthetic
    from typo_cases import misspell
    sm, path = small_model
    ref, orc = refbridge.RefKiwi(path), oraclelib.OracleKiwi(path)
    ents, cont, leng = refbridge.default_typo_entries(name)
    rt = refbridge.RefTypo(); rt.update_default(name); rt.prepare(True)
    ot = oraclelib.OracleTypo(); ot.update_entries(ents, cont, leng); ot.prepare(True)
    rnd = random.Random(5)
    dia = 0xFFFF if name == "dialect" else 0
    tt = [misspell(t, rnd, True, carry, "lengthening" in name) for t in synthetic(sm, 70, 701, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 40, 702)] + EDGE_TEXTS
    corrected = 0
    for t in tt:
        if not t.strip():
            continue
        assert ref.split_typo(rt, t, threshold, dia) == orc.split_typo(ot, t, threshold, dia), t
        a = ref.analyze_typo(rt, t, threshold, dia)
        assert _norm(a) == _norm(orc.analyze_typo(ot, t, threshold, dia)), t
        corrected += any(x.typo_cost > 0 for x in a[0][0])
    assert corrected >= (20 if name.startswith("basic") or carry or "lengthening" in name else 1)


def test_golden_typo_analyses(small_model):
    """Analyses of the real reference with this repo's test rules on misspelt texts (tools/make_golden_typo.py), replayed without oracle/_ref."""
    import oraclelib
    g = json.load(open(os.path.join(HERE, "golden", "typo_analyses.json"), encoding="utf-8"))
    orc = oraclelib.OracleKiwi(small_model[1])
    ot = oraclelib.OracleTypo(g["continual"], INF)
    fill(ot, False)
    ot.prepare(True)
    for it in g["items"]:
        res = orc.analyze_typo(ot, it["text"], g["threshold"], 0)
        got = [[t.form, t.tag, t.position, t.length, t.score, t.typo_cost] for t in res[0][0]]
        assert got == it["tokens"] and res[0][1] == it["score"], it["text"]


def test_typo_correction_with_a_skipbigram_model_equals_reference(small_sbg_model):
    """The reference's typo transformers work with every model type: SkipBigram model + typo transformer, real reference vs oracle.  Texts whose
    lattices stay within the small / medium containers must agree exactly; the large container hands tied paths on in a history-dependent order
    (tests/test_oracle_vs_ref.py::test_skipbigram_analyses_match_reference), so there the best score is what is compared."""
    import random
    from dataclasses import astuple
    import oraclelib
    import refbridge
    from corpora import dictionary_mix, synthetic
    from typo_cases import misspell
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    sm, path = small_sbg_model
    orc, ref = oraclelib.OracleKiwi(path), refbridge.RefKiwi(path)
    name = "basic_with_continual"
    ents, cont, leng = refbridge.default_typo_entries(name)
    rt = refbridge.RefTypo(); rt.update_default(name); rt.prepare(True)
    ot = oraclelib.OracleTypo(); ot.update_entries(ents, cont, leng); ot.prepare(True)
    rnd = random.Random(21)
    tt = [misspell(t, rnd, True, True, False) for t in synthetic(sm, 50, 751, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 30, 752)]

    def norm(res):
        return [([astuple(t) for t in a[0]], a[1]) for a in res]
    prev = orc.counters()
    exact = corrected = 0
    for t in tt:
        if not t.strip():
            continue
        a, b = ref.analyze_typo(rt, t, 2.5, 0), orc.analyze_typo(ot, t, 2.5, 0)
        c = orc.counters()
        large = c["nodesOver512"] > prev["nodesOver512"]
        prev = c
        assert a[0][1] == b[0][1], t
        if not large:
            assert norm(a) == norm(b), t
            exact += 1
        corrected += any(x.typo_cost > 0 for x in a[0][0])
    assert exact >= 20 and corrected >= 5
