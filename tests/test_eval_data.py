"""Real text: column 1 of the reference's eval_data files (web, written, web_with_typos, web_with_cont_typos) analysed on the small synthetic model
extended by the files' gold (form, tag) pairs as dictionary entries (kiwi_amd.workloads.eval_model), against what the REAL reference answered on
that model -- tests/golden/eval_data_*.json, written by tools/make_golden_eval.py in the build container: tokens, positions, word / sentence /
line numbers, fp32 scores, per-token typo costs.  CPU: the oracle (plain files) and the lane-emulated kernels (a sample of every file, typo files
with the built-in set basicTypoSetWithContinual); `-m gpu`: every line of every file on the MI355X.  What this does NOT pin is accuracy against
the gold annotations: the language model is synthetic (the shipped model files are git-LFS pointers here)."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ("web", "written", "web_with_typos", "web_with_cont_typos")
TYPO_SET = 3      # DefaultTypoSet::basicTypoSetWithContinual


def _golden(name):
    return json.load(open(os.path.join(HERE, "golden", f"eval_data_{name}.json"), encoding="utf-8"))


def _rows(tokens):
    return [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.line_number, t.score, t.typo_cost] for t in tokens]


def _check_device(lib_path, name, limit=None):
    from kiwi_amd.api import KiwiAmd, Typo
    from kiwi_amd.workloads import eval_model
    path, _ = eval_model()
    g = _golden(name)
    items = g["items"][:limit] if limit else g["items"]
    dev = KiwiAmd(path, lib_path=lib_path) if lib_path else KiwiAmd(path)
    typo = None
    if g["typo"]:
        typo = Typo.from_default(dev.lib, TYPO_SET)
        typo.prepare(True)
    texts = [it["text"] for it in items]
    res = dev.analyze_batch_opt(texts, typo=typo, typo_threshold=2.5) if typo is not None else dev.analyze_batch(texts)
    got = res.to_python()
    res.close()
    for it, y in zip(items, got):
        assert _rows(y[0][0]) == it["tokens"] and y[0][1] == it["score"], it["text"]
    if typo is not None:
        typo.close()
    dev.close()
    return len(items)


def test_golden_files_are_the_eval_data_of_the_reference():
    lex = json.load(open(os.path.join(HERE, "golden", "eval_data_lexicon.json"), encoding="utf-8"))["entries"]
    assert len(lex) > 2000
    n = {name: len(_golden(name)["items"]) for name in FILES}
    assert n == {"web": 158, "written": 33, "web_with_typos": 97, "web_with_cont_typos": 97}, n
    assert sum(len(it["tokens"]) for it in _golden("web")["items"]) > 3000


@pytest.mark.parametrize("name", ["web", "written"])
def test_oracle_equals_reference_on_eval_data(name):
    import oraclelib
    from kiwi_amd.workloads import eval_model
    path, _ = eval_model()
    orc = oraclelib.OracleKiwi(path)
    for it in _golden(name)["items"]:
        got = orc.analyze(it["text"])
        assert _rows(got[0][0]) == it["tokens"] and got[0][1] == it["score"], it["text"]


@pytest.mark.parametrize("name", FILES)
def test_emulated_device_equals_reference_on_eval_data(name):
    import subprocess
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    assert _check_device(os.path.join(emu, "_build", "libkiwi_hipemu.so"), name, limit=48) >= 33


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILES)
def test_device_equals_reference_on_eval_data(name):
    assert _check_device(None, name) >= 33
