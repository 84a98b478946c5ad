"""Dialects: kiwi_init's enabled_dialects and AnalyzeOption::allowedDialects / dialectCost on a model WITH dialect morphemes -- the small synthetic
model + the gold lexicon of eval_data + the gold (form, tag) pairs of the reference's eval_data/dialect files as morphemes of those files' dialects
(kiwi_amd.workloads.dialect_model; tests/golden/eval_dialect_lexicon.json) -- on sentences of those files, against what the REAL reference answered
(tests/golden/eval_dialect.json, written by tools/make_golden_dialect.py in the build container: enabled = all with allowed 0 / the file's own dialect / all / all at
cost 1.5, enabled = none with allowed all; no transformer given, so the reference corrects with its built-in `dialect` typo set; tokens, positions, fp32 scores,
typo costs, the dialect of every token).  CPU: the oracle, and the lane-emulated kernels (default choice, position steps forced, one chunk per wavefront);
`-m gpu`: every sentence on the MI355X through the low-level ABI (kiwi_init / kiwi_analyze with the reference's own option struct: tests/test_gpu_capi.py)."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ALL = 1023      # Dialect::all


def _golden():
    return json.load(open(os.path.join(HERE, "golden", "eval_dialect.json"), encoding="utf-8"))["items"]


def _rows(res):
    toks, score = res[0]
    return {"score": score, "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.line_number, t.score, t.typo_cost, t.dialect] for t in toks]}


def _sample(items, per_file):
    seen, out = {}, []
    for it in items:
        if seen.setdefault(it["file"], 0) < per_file:
            seen[it["file"]] += 1
            out.append(it)
    return out


def test_golden_files_hold_dialect_analyses_of_the_reference():
    items = _golden()
    lex = json.load(open(os.path.join(HERE, "golden", "eval_dialect_lexicon.json"), encoding="utf-8"))["entries"]
    assert len(items) >= 300 and len(lex) > 2000 and len({it["file"] for it in items}) == 9
    assert sum(1 for it in items for t in it["enabled_all"]["allowed_all"]["tokens"] if t[9]) > 800      # dialect tokens under allowed = all
    assert sum(1 for it in items for t in it["enabled_all"]["allowed_0"]["tokens"] if t[9]) == 0         # none when only the standard language is allowed
    assert sum(1 for it in items if it["enabled_all"]["allowed_all"]["tokens"] != it["enabled_all"]["allowed_0"]["tokens"]) > 300
    assert sum(1 for it in items if it["enabled_all"]["allowed_all"]["score"] != it["enabled_all"]["allowed_all_cost_1.5"]["score"]) > 250


def test_oracle_equals_reference_with_dialects():
    import oraclelib
    from kiwi_amd.workloads import dialect_model
    path = dialect_model()
    orc_all, orc_std = oraclelib.OracleKiwi(path, enabled_dialects=ALL), oraclelib.OracleKiwi(path, enabled_dialects=0)
    for it in _sample(_golden(), 14):
        text, g = it["text"], it["enabled_all"]
        assert _rows(orc_all.analyze_dialect(text, 0)) == g["allowed_0"], text
        assert _rows(orc_all.analyze_dialect(text, it["bit"])) == g["allowed_own"], text
        assert _rows(orc_all.analyze_dialect(text, ALL)) == g["allowed_all"], text
        assert _rows(orc_all.analyze_dialect(text, ALL, 1.5)) == g["allowed_all_cost_1.5"], text
        assert _rows(orc_std.analyze_dialect(text, ALL)) == it["enabled_none"]["allowed_all"], text


def _check_device(lib_path, items, with_capi=False):
    from kiwi_amd.api import KiwiAmd, Typo
    from kiwi_amd.workloads import dialect_model
    path = dialect_model()
    kw = {"lib_path": lib_path} if lib_path else {}
    dev_all, dev_std = KiwiAmd(path, enabled_dialects=ALL, **kw), KiwiAmd(path, enabled_dialects=0, **kw)
    texts = [it["text"] for it in items]

    def run(dev, allowed, cost, want, typo=None, idx=None):
        idx = list(range(len(items))) if idx is None else idx
        res = dev.analyze_batch_dialect([texts[i] for i in idx], allowed, cost, typo=typo)
        got = res.to_python()
        res.close()
        for i, y in zip(idx, got):
            assert _rows(y) == want(items[i]), (allowed, cost, texts[i])
    run(dev_all, 0, 3.0, lambda it: it["enabled_all"]["allowed_0"])
    run(dev_all, ALL, 3.0, lambda it: it["enabled_all"]["allowed_all"])
    run(dev_all, ALL, 1.5, lambda it: it["enabled_all"]["allowed_all_cost_1.5"])
    run(dev_std, ALL, 3.0, lambda it: it["enabled_none"]["allowed_all"])
    for bit in sorted({it["bit"] for it in items}):
        run(dev_all, bit, 3.0, lambda it: it["enabled_all"]["allowed_own"], idx=[i for i, it in enumerate(items) if it["bit"] == bit])
    # a transformer of the caller's beside the allowed dialect (DefaultTypoSet::basicTypoSetWithContinual)
    typo = Typo.from_default(dev_all.lib, 3)
    typo.prepare(True)
    key = "allowed_own_typo_basic_with_continual"
    for bit in sorted({it["bit"] for it in items}):
        idx = [i for i, it in enumerate(items) if it["bit"] == bit and key in it["enabled_all"]]
        if idx:
            run(dev_all, bit, 3.0, lambda it: it["enabled_all"][key], typo=typo, idx=idx)
    typo.close()
    dev_all.close(); dev_std.close()
    return len(items)


@pytest.mark.parametrize("variant", ["default", "pos", "64"])
def test_emulated_device_equals_reference_with_dialects(monkeypatch, variant):
    from corpora import force_lanes
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    if variant != "default":
        force_lanes(monkeypatch, variant)
    assert _check_device(os.path.join(emu, "_build", "libkiwi_hipemu.so"), _sample(_golden(), 7 if variant == "default" else 4)) >= 30


@pytest.mark.gpu
def test_device_equals_reference_with_dialects():
    assert _check_device(None, _golden()) >= 300
