"""SURVEY.md section 8 row a4, first half: the typo graph (TypoTransformer rule container -> prepare() -> generateGraph).
oracle/typo_oracle.hpp is a CPU restatement; here it is pinned, byte for byte, to the REAL reference translation unit
(src/TypoTransformer.cpp through oracle/ref_bridge.cpp), to the reference's own known-answer test, and to committed graphs
generated from the reference (tests/golden/typo_graphs.json).  The lattice search over such graphs (the second half of a4,
a5's per-branch search states) is not built yet."""
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
