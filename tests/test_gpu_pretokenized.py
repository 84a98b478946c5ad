"""GPU (-m gpu): pretokenized spans ON THE DEVICE PATH -- the `pretokenized` argument of kiwi_analyze{,_w} (include/kiwi_capi.h <- reference capi.h:1351-1407;
Kiwi::analyze src/Kiwi.cpp:785-946, 1043-1051; KTrie.cpp:782-790, 1177-1210): every golden case of the REAL reference (tests/golden/pretokenized_small.json, 190
cases: spans without tokens, one token reused / with a temporary form and morpheme, several tokens as one temporary morpheme with chunks, neighbouring spans, spans
over spaces and at both ends, top-3) through the low-level batch ABI (inferRegularity as the case gives it) and through kiwi_analyze_w with a kiwi_pretokenized_h
built by kiwi_pt_* (the C API always infers); freshly generated cases against the live reference library and the oracle; byte offsets through kiwi_analyze; the
error convention.  tests/test_hipemu.py re-runs this file on the CPU against the lane-emulated build of the same sources (KAMD_TEST_LIB)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from test_gpu_capi import LIB, Option, capi, kiwi, opt  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "pretokenized_small.json"), encoding="utf-8"))


@pytest.fixture(scope="module")
def dev(small_model):
    from kiwi_amd.api import KiwiAmd
    d = KiwiAmd(small_model[1], lib_path=LIB)
    yield d
    d.close()


def _rows(res):
    return json.loads(json.dumps([{"score": r[1], "tokens": [[x.form, x.tag, x.position, x.length, x.word_position, x.sent_position, x.score, x.typo_form_id, x.morph_id >= 0] for x in r[0]]} for r in res]))


def _same_up_to_exact_ties(got, want, top_n, what):
    """The reference's best analysis and every score bit for bit; beyond the best one up to exactly tied analyses (the top-N rule, include/kiwi_capi.h)."""
    assert [r["score"] for r in got] == [r["score"] for r in want], what
    if top_n == 1:
        assert got == want, what
    for a, b in zip(got, want):
        if [r["score"] for r in got].count(a["score"]) == 1 and a is not got[-1]:
            assert a == b, what


def test_golden_cases_through_the_batch_abi(golden, dev):
    temp = inside = 0
    for g in golden["cases"]:
        spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in g["spans"]]
        got = _rows(dev.analyze_pretokenized(g["text"], spans, top_n=g["top_n"]))
        _same_up_to_exact_ties(got, g["results"], g["top_n"], g["text"])
        temp += any(not t[8] for t in got[0]["tokens"])
        inside += any(t[7] for t in got[0]["tokens"])
    assert len(golden["cases"]) == 190 and temp > 80 and inside == 190


def _pt_api(L):
    L.kiwi_pt_init.restype = C.c_void_p
    L.kiwi_pt_add_span.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_pt_add_token_to_span.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.kiwi_pt_add_token_to_span_w.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.kiwi_pt_close.argtypes = [C.c_void_p]


def _make_pt(L, k, spans, wide=True, offset=lambda x: x):
    pt = L.kiwi_pt_init()
    assert pt
    for sb, se, toks in spans:
        sid = L.kiwi_pt_add_span(pt, offset(sb), offset(se))
        assert sid >= 0, L.kiwi_error()
        for form, tb, te, tag, _infer in toks:
            name = L.kiwi_tag_to_string(k, tag)
            if wide:
                u = np.frombuffer((form + "\0").encode("utf-16-le"), np.uint16).copy()
                assert L.kiwi_pt_add_token_to_span_w(pt, sid, u.ctypes.data, name, tb, te) == 0, L.kiwi_error()
            else:
                assert L.kiwi_pt_add_token_to_span(pt, sid, form.encode("utf-8"), name, tb, te) == 0, L.kiwi_error()
    return pt


def _read(L, k, r):
    out = []
    for i in range(L.kiwi_res_size(r)):
        toks = []
        for j in range(L.kiwi_res_word_num(r, i)):
            ti = L.kiwi_res_token_info(r, i, j).contents
            toks.append([L.kiwi_res_form(r, i, j).decode("utf-8"), ti.tag, ti.chr_position, ti.length, ti.word_position, ti.sent_position, ti.score, ti.typo_form_id, L.kiwi_res_morpheme_id(r, i, j, k) >= 0])
        out.append({"score": L.kiwi_res_prob(r, i), "tokens": toks})
    return json.loads(json.dumps(out))


def test_golden_cases_through_kiwi_analyze_w_with_a_pretokenized_handle(golden, capi, kiwi, dev):
    """kiwi_pt_init / kiwi_pt_add_span / kiwi_pt_add_token_to_span_w / kiwi_analyze_w(..., pt) / kiwi_pt_close -- the reference's own entry points.  A token added
    through the C API infers its regularity (BasicToken's default): cases whose fixture says otherwise are compared with the batch ABI's answer for the inferring
    token instead of the fixture."""
    _pt_api(capi)
    n_fixture = 0
    for g in golden["cases"]:
        pt = _make_pt(capi, kiwi, g["spans"])
        u = np.frombuffer((g["text"] + "\0").encode("utf-16-le"), np.uint16).copy()
        r = capi.kiwi_analyze_w(kiwi, u.ctypes.data, g["top_n"], opt(), pt)
        assert r, capi.kiwi_error()
        got = _read(capi, kiwi, r)
        capi.kiwi_res_close(r)
        assert capi.kiwi_pt_close(pt) == 0
        if all(tk[4] == 1 for _, _, toks in g["spans"] for tk in toks):
            _same_up_to_exact_ties(got, g["results"], g["top_n"], g["text"])
            n_fixture += 1
        else:
            spans = [(sb, se, [tuple(tk[:4]) + (1,) for tk in toks]) for sb, se, toks in g["spans"]]
            _same_up_to_exact_ties(got, _rows(dev.analyze_pretokenized(g["text"], spans, top_n=g["top_n"])), g["top_n"], g["text"])
    assert n_fixture > 150


def test_fresh_cases_against_the_live_reference_and_the_oracle(dev, small_model, monkeypatch):
    """2 x 240 freshly generated cases (other seeds than the fixture's; every case kind): the device path == the oracle, and == the real reference library where
    it travelled (oracle/_ref, kref_analyze_pretokenized)."""
    import oraclelib
    import refbridge
    import make_golden_pretokenized as gen
    monkeypatch.setenv("KORC_QUIET", "1")
    sm, path = small_model
    orc = oraclelib.OracleKiwi(path)
    ref = refbridge.RefKiwi(path) if refbridge.available() else None
    kinds, n = set(), 0
    for seed in (4301, 5301):
        for c in gen.make_cases(sm, n=240, seed=seed):
            spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in c["spans"]]
            got = _rows(dev.analyze_pretokenized(c["text"], spans, top_n=c["top_n"]))
            want = orc.analyze_pretokenized(c["text"], spans, top_n=c["top_n"])
            assert want is not None, c
            assert got == _rows(want), c["text"]      # (device and oracle keep tied analyses in the same -- insertion -- order)
            if ref is not None:
                _same_up_to_exact_ties(got, json.loads(json.dumps(gen.run(ref, c))), c["top_n"], c["text"])
            kinds.update("none" if not toks else "one" if len(toks) == 1 else "multi" for _, _, toks in c["spans"])
            n += 1
    assert kinds == {"none", "one", "multi"} and n >= 300


def test_utf8_entry_point_takes_byte_offsets(golden, capi, kiwi):
    """kiwi_analyze: a span's begin / end are BYTE offsets into the UTF-8 text (Kiwi::mapPretokenizedSpansToU16, src/Kiwi.cpp:34-44); the tokens' offsets inside a
    span are not mapped (the reference takes them as they are).  Same analyses as the UTF-16 entry point."""
    _pt_api(capi)
    done = 0
    for g in golden["cases"][:60]:
        b8 = lambda x, t=g["text"]: len(t[:x].encode("utf-8"))      # noqa: E731
        pt8 = _make_pt(capi, kiwi, g["spans"], wide=False, offset=b8)
        r8 = capi.kiwi_analyze(kiwi, g["text"].encode("utf-8"), g["top_n"], opt(), pt8)
        assert r8, capi.kiwi_error()
        pt16 = _make_pt(capi, kiwi, g["spans"])
        u = np.frombuffer((g["text"] + "\0").encode("utf-16-le"), np.uint16).copy()
        r16 = capi.kiwi_analyze_w(kiwi, u.ctypes.data, g["top_n"], opt(), pt16)
        assert _read(capi, kiwi, r8) == _read(capi, kiwi, r16), g["text"]
        for h in (r8, r16):
            capi.kiwi_res_close(h)
        for h in (pt8, pt16):
            capi.kiwi_pt_close(h)
        done += 1
    assert done == 60


def test_span_errors_and_plain_calls_next_to_span_calls(golden, capi, kiwi, dev, oracle):
    """Overlapping spans: NULL + the reference's message; an unknown tag is refused when the token is added; a handle without spans is no constraint; a plain
    analysis right after one with temporary morphemes is untouched by what that call left behind the model's tables."""
    _pt_api(capi)
    g = next(c for c in golden["cases"] if any(not t[8] for t in c["results"][0]["tokens"]))
    text = g["text"]
    u = np.frombuffer((text + "\0").encode("utf-16-le"), np.uint16).copy()
    pt = capi.kiwi_pt_init()
    capi.kiwi_pt_add_span(pt, 0, 3)
    capi.kiwi_pt_add_span(pt, 2, 5)
    assert not capi.kiwi_analyze_w(kiwi, u.ctypes.data, 1, opt(), pt)
    assert b"overlapped" in capi.kiwi_error()
    capi.kiwi_pt_close(pt)
    pt = capi.kiwi_pt_init()
    sid = capi.kiwi_pt_add_span(pt, 0, 2)
    assert capi.kiwi_pt_add_token_to_span(pt, sid, "가".encode("utf-8"), b"NOSUCHTAG", 0, 2) != 0
    capi.kiwi_pt_close(pt)
    empty = capi.kiwi_pt_init()
    r = capi.kiwi_analyze_w(kiwi, u.ctypes.data, 1, opt(), empty)
    assert r, capi.kiwi_error()
    plain = _read(capi, kiwi, r)
    capi.kiwi_res_close(r)
    capi.kiwi_pt_close(empty)
    spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in g["spans"]]
    assert _rows(dev.analyze_pretokenized(text, spans))[0] == g["results"][0]
    again = dev.analyze_batch([text]).to_python()[0]
    want = oracle.analyze(text)
    assert [([(t.form, t.tag, t.position, t.length, t.score, t.morph_id) for t in a[0]], a[1]) for a in again] == [([(t.form, t.tag, t.position, t.length, t.score, t.morph_id) for t in a[0]], a[1]) for a in want]
    assert [t[:7] for t in plain[0]["tokens"]] == [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.score] for t in want[0][0]]


@pytest.mark.parametrize("kind", ["sbg", "cong", "cong-chr"])
def test_spans_under_the_other_language_models(kind, small_sbg_model, small_cong_model, small_cong_chr_model, monkeypatch):
    """The span's forced node and its temporary morphemes under SkipBigram and CoNgram scoring (the temporaries' LM id is their tag's default morpheme), and with
    Match::oovChrModel on a model that carries the character model (a temporary form's own string is scored like a dictionary form's, per batch): device == oracle."""
    import oraclelib
    import make_golden_pretokenized as gen
    from kiwi_amd.api import KiwiAmd, MATCH_ALL_WITH_NORMALIZING
    monkeypatch.setenv("KORC_QUIET", "1")
    sm, path = {"sbg": small_sbg_model, "cong": small_cong_model, "cong-chr": small_cong_chr_model}[kind]
    match = MATCH_ALL_WITH_NORMALIZING | ((1 << 8) if kind == "cong-chr" else 0)
    d = KiwiAmd(path, lib_path=LIB)
    orc = oraclelib.OracleKiwi(path)
    n = temp = 0
    for c in gen.make_cases(sm, n=80 if kind != "sbg" else 48, seed=6301):
        spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in c["spans"]]
        want = orc.analyze_pretokenized(c["text"], spans, top_n=c["top_n"], match=match)
        assert want is not None, c
        got = _rows(d.analyze_pretokenized(c["text"], spans, top_n=c["top_n"], match=match))
        assert got == _rows(want), (kind, c["text"])
        temp += any(not t[8] for t in got[0]["tokens"])
        n += 1
    d.close()
    assert n >= 30 and temp >= 10
