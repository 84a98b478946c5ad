"""CPU: pretokenized spans -- the argument of kiwi_analyze* the product still refuses.  What exists is the pin: analyses of the REAL
reference with spans on the small synthetic model, committed as tests/golden/pretokenized_small.json (tools/make_golden_pretokenized.py).  Where the reference library
travelled, the golden vectors are replayed against it (the fixture is what the reference answers, not a stale copy); the structural facts the restatement and the device
path will have to reproduce are asserted on the fixture itself, so that they hold on the GPU box too."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "pretokenized_small.json"), encoding="utf-8"))


def test_golden_vectors_are_what_the_reference_answers(golden, small_model):
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref/libkiwi_ref.so not built (needs /root/reference)")
    import make_golden_pretokenized as gen
    sm, path = small_model
    ref = refbridge.RefKiwi(path)
    cases = gen.make_cases(sm)
    assert len(cases) == len(golden["cases"])
    for c, g in zip(cases, golden["cases"]):
        assert c["text"] == g["text"] and c["spans"] == g["spans"] and c["top_n"] == g["top_n"]
        assert json.loads(json.dumps(gen.run(ref, c))) == g["results"], c["text"]


def test_what_a_span_does_to_an_analysis(golden):
    """Facts of the fixture: every token either lies inside exactly one span and carries a non-zero typo_form_id (the span's index + 1 counted from the first span of
    its CHUNK: findPretokenizedGroupOfNode, src/Kiwi.cpp:949-969, is handed the chunk's spans), or overlaps none and carries 0; the tokens of a span cover it from
    its first unit; a span given with tokens comes back as those tokens (forms and tags)."""
    n_tok_cases = 0
    for g in golden["cases"]:
        best = g["results"][0]["tokens"]
        for t in best:
            within = [i for i, (b, e, _) in enumerate(g["spans"]) if b <= t[2] and t[2] + t[3] <= e]
            overlap = [i for i, (b, e, _) in enumerate(g["spans"]) if t[2] < e and t[2] + t[3] > b]
            if t[7]:
                assert len(within) == 1, (g["text"], t)
            else:
                assert not overlap, (g["text"], t)
        for b, e, toks in g["spans"]:
            inside = [t for t in best if b <= t[2] and t[2] + t[3] <= e and t[7]]
            assert inside and min(t[2] for t in inside) == b, (g["text"], g["spans"])
            if toks:
                n_tok_cases += 1
                assert [(t[0], t[1] & 0x7F) for t in inside] == [(tk[0], tk[3]) for tk in toks], (g["text"], inside, toks)
    assert n_tok_cases > 80


def test_oracle_equals_the_reference_where_it_restates_spans(golden, small_model, monkeypatch):
    """oracle/oracle.cpp analyzeOne with spans (the chunk cut stepping over a span, the span's node in the lattice, typoFormId of the tokens inside) -- ALL cases of
    makePretokenizedSpanGroup since round 5: spans that point at a form of the model (no tokens given, or one token that is a single-candidate dictionary entry) and
    spans that need temporary forms / morphemes (any other single token; several tokens: one morpheme whose chunks are the tokens), for which the oracle bakes the
    model once more with them behind its own entries (model.cpp bakeModelWithTemps).  Every golden case: the reference's best analysis and every score bit for bit;
    beyond the best one up to exactly tied analyses (the top-N rule, include/kiwi_capi.h)."""
    import oraclelib
    monkeypatch.setenv("KORC_QUIET", "1")
    # (... and the product's per-batch overlay of the temporaries -- model.cpp bakeTempsOverlay, computed from the baked model alone -- is compared with what the
    # second bake appended to every table, case by case: a mismatch refuses the case)
    monkeypatch.setenv("KORC_CHECK_OVERLAY", "1")
    sm, path = small_model
    orc = oraclelib.OracleKiwi(path)
    done = refused = 0
    for g in golden["cases"]:
        spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in g["spans"]]
        res = orc.analyze_pretokenized(g["text"], spans, top_n=g["top_n"])
        if res is None:
            refused += 1
            assert any(toks for _, _, toks in g["spans"]), g["spans"]      # (a span without tokens is always restated)
            continue
        got = json.loads(json.dumps([{"score": r[1], "tokens": [[x.form, x.tag, x.position, x.length, x.word_position, x.sent_position, x.score, x.typo_form_id, x.morph_id >= 0] for x in r[0]]} for r in res]))
        assert [r["score"] for r in got] == [r["score"] for r in g["results"]], g["text"]
        if g["top_n"] == 1:
            assert got == g["results"], g["text"]
        # (beyond the best analysis: which of exactly tied analyses come out, and in which order, is the reference's hash-bucket order -- the unknown-noun
        # readings NNG / NNP of one form tie exactly; an analysis whose score is unique in both lists has to be the same one)
        for a, b in zip(got, g["results"]):
            if [r["score"] for r in got].count(a["score"]) == 1 and a is not got[-1]:      # (the last one may tie with an analysis beyond the cut)
                assert a == b, g["text"]
        done += 1
    assert done == len(golden["cases"]) == 190 and refused == 0


def test_oracle_equals_the_live_reference_on_fresh_span_cases(small_model, monkeypatch):
    """Beyond the committed fixture: 2 x 300 freshly generated cases (other seeds; every case kind, temporary morphemes included) through the REAL reference
    (oracle/_ref, kref_analyze_pretokenized) and the oracle -- best analysis and score lists bit for bit, the rest up to exact ties."""
    import oraclelib
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref/libkiwi_ref.so not built (needs /root/reference)")
    import make_golden_pretokenized as gen
    monkeypatch.setenv("KORC_QUIET", "1")
    monkeypatch.setenv("KORC_CHECK_OVERLAY", "1")
    sm, path = small_model
    ref = refbridge.RefKiwi(path)
    orc = oraclelib.OracleKiwi(path)
    kinds = set()
    for seed in (2301, 3301):
        for c in gen.make_cases(sm, n=240, seed=seed):
            want = json.loads(json.dumps(gen.run(ref, c)))
            spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in c["spans"]]
            res = orc.analyze_pretokenized(c["text"], spans, top_n=c["top_n"])
            assert res is not None, c
            got = json.loads(json.dumps([{"score": r[1], "tokens": [[x.form, x.tag, x.position, x.length, x.word_position, x.sent_position, x.score, x.typo_form_id, x.morph_id >= 0] for x in r[0]]} for r in res]))
            assert [r["score"] for r in got] == [r["score"] for r in want], c["text"]
            if c["top_n"] == 1:
                assert got == want, c["text"]
            for a, b in zip(got, want):
                if [r["score"] for r in got].count(a["score"]) == 1 and a is not got[-1]:
                    assert a == b, c["text"]
            kinds.update("none" if not toks else "one" if len(toks) == 1 else "multi" for _, _, toks in c["spans"])
    assert kinds == {"none", "one", "multi"}
