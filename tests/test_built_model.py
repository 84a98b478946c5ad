"""A model as the REAL KiwiBuilder builds it (src/KiwiBuilder.cpp compiled unmodified into oracle/_ref/libkiwi_ref_x86.so): the small synthetic
sj.morph / sj.knlm + the eval_data gold lexicon, then the reference's own shipped combiningRule.txt, default.dict (113 k entries) and typo.dict through
loadDictionary / buildCombinedMorphemes / build -- exported after that step by tools/export_built.cpp as a raw-model container (tests/golden/eval_built_model.raw.xz, 15 k
rule-combined morphemes) together with what the built Kiwi itself answered on column 1 of the eval_data files (tests/golden/eval_built_*.json);
both written by tools/make_golden_built.py in the build container.

  * where /root/reference and the x86 reference build exist: the export is reproduced byte for byte, and the dictionary the reference bakes from the
    container equals the one KiwiBuilder::build() baked from the directory -- the container route the parity tests use loses nothing of a real build;
  * everywhere: the oracle and the product bake the same dictionary from the container, the oracle and the lane-emulated kernels answer what the built
    Kiwi answered (typo files with the built-in set basicTypoSetWithContinual); `-m gpu`: tests/test_zz_gpu_built_model.py, every line on the MI355X.

What this covers beyond tests/test_eval_data.py: pre-combined and allomorph morphemes with chunks, combineSocket / combined links over 15 k rule
products, pre-analysed multi-morpheme dictionary words, typo.dict's pre-analysed corrections.  The language model stays synthetic."""
import json
import lzma
import os
import struct

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FILES = ("web", "written", "web_with_typos", "web_with_cont_typos")
TYPO_SET = 3      # DefaultTypoSet::basicTypoSetWithContinual
MODEL_TYPE, OPTIONS = 2, 1 | 2 | 4


def _golden(name):
    return json.load(open(os.path.join(HERE, "golden", f"eval_built_{name}.json"), encoding="utf-8"))


def _rows(tokens):
    return [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.line_number, t.score, t.typo_cost] for t in tokens]


def built_model_path():
    """tests/golden/eval_built_model.raw.xz unpacked under _data/ (kept between runs)."""
    src = os.path.join(HERE, "golden", "eval_built_model.raw.xz")
    dst = os.path.join(ROOT, "_data", "eval-built.raw")
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with lzma.open(src) as f, open(dst + ".tmp", "wb") as g:
            g.write(f.read())
        os.replace(dst + ".tmp", dst)
    return dst


def mask_unused(dump: bytes) -> bytes:
    """A dictionary dump (layout: oracle/ref_bridge.cpp kref_dump_dict) with Morpheme::origMorphemeId zeroed: nothing on the analysis path reads it
    (it serves KiwiBuilder's word extraction and the joiner), the product does not store it."""
    b = bytearray(dump)
    n_forms, n_morphs = struct.unpack_from("<II", b, 0)
    o = 8
    for _ in range(n_forms):
        n, = struct.unpack_from("<I", b, o); o += 4 + 2 * n + 4 + 4 + 2
        n, = struct.unpack_from("<I", b, o); o += 4 + 4 * n
    for _ in range(n_morphs):
        o += 4 + 7 + 4 + 4 + 4
        b[o:o + 4] = b"\0\0\0\0"; o += 4 + 2
        n, = struct.unpack_from("<I", b, o); o += 4 + 6 * n
    assert len(b) - o < 4 * 64, "not the layout of kref_dump_dict"
    return bytes(b)


def check_device(lib_path, name, limit=None):
    from kiwi_amd.api import KiwiAmd, Typo
    g = _golden(name)
    items = g["items"][:limit] if limit else g["items"]
    path = built_model_path()
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


def test_fixture_is_a_built_model():
    import numpy as np
    from kiwi_amd.container import read_container
    from kiwi_amd.synth import MORPH_DTYPE
    n = {name: len(_golden(name)["items"]) for name in FILES}
    assert n == {"web": 158, "written": 33, "web_with_typos": 97, "web_with_cont_typos": 97}, n
    kind, sec = read_container(built_model_path())
    assert kind == b"KAMDRAW1"
    morphs = np.frombuffer(sec["morph"].tobytes(), MORPH_DTYPE)
    combined = int((morphs["combined"] != 0).sum())
    with_chunks = int((morphs["n_chunks"] != 0).sum())
    assert len(morphs) > 100000 and combined > 300 and with_chunks > 15000, (len(morphs), combined, with_chunks)
    # the rules matter: the same texts on the model without them (tests/golden/eval_data_web.json) come out differently
    plain = json.load(open(os.path.join(HERE, "golden", "eval_data_web.json"), encoding="utf-8"))["items"]
    differ = sum(a["tokens"] != b["tokens"] for a, b in zip(plain, _golden("web")["items"]))
    assert differ > 40, differ


@pytest.mark.skipif(not os.path.isdir("/root/reference/models/cong/base"), reason="needs the reference checkout (build container)")
def test_real_kiwibuilder_build_is_what_the_container_holds(tmp_path):
    import sys
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import shutil
    import make_golden_built as tool
    from kiwi_amd.workloads import eval_model
    raw, _ = eval_model(for_builder=True)
    lib, d = tool.shipped_dir(raw)
    try:
        out = str(tmp_path / "export.raw")
        report = tool.export(d, out)                                                     # tools/export_built.cpp, the program a maintainer would run
        assert "15272 rule-combined" in report and "135014 morphemes" in report, report
        assert open(out, "rb").read() == open(built_model_path(), "rb").read()          # the committed fixture is this build
        real = refbridge.RefKiwi.built(d, MODEL_TYPE, OPTIONS)                          # KiwiBuilder::build() itself
        via_container = refbridge.RefKiwi(built_model_path())                          # the bridge's bake of the exported tables
        assert real.dump_dict() == via_container.dump_dict()
        for it in _golden("web")["items"][:40]:
            a, b = real.analyze(it["text"]), via_container.analyze(it["text"])
            assert _rows(a[0][0]) == _rows(b[0][0]) == it["tokens"] and a[0][1] == b[0][1] == it["score"], it["text"]
    finally:
        shutil.rmtree(d)


def test_oracle_and_product_bake_the_reference_dictionary():
    import subprocess
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    path = built_model_path()
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    orc = oraclelib.OracleKiwi(path).dump_dict()
    dev = KiwiAmd(path, lib_path=os.path.join(emu, "_build", "libkiwi_hipemu.so"))
    prod = dev.dump_dict()
    dev.close()
    assert mask_unused(orc) == mask_unused(prod)
    import refbridge
    if refbridge.available():
        ref = refbridge.RefKiwi(path).dump_dict()
        a, b = mask_unused(ref), mask_unused(orc)
        assert len(a) == len(b)
        # the reference's sentinel form (last form record) carries uninitialised flag bits: compare around it (tests/test_oracle_vs_ref.py)
        assert sum(x != y for x, y in zip(a, b)) <= 1


@pytest.mark.parametrize("name", FILES)
def test_oracle_equals_the_built_reference(name):
    import oraclelib
    g = _golden(name)
    orc = oraclelib.OracleKiwi(built_model_path())
    typo = None
    if g["typo"]:
        import refbridge
        if not refbridge.available():
            pytest.skip("the built-in typo set is read out of oracle/_ref")
        ents, cont, leng = refbridge.default_typo_entries("basic_with_continual")
        typo = oraclelib.OracleTypo(); typo.update_entries(ents, cont, leng); typo.prepare(True)
    for it in g["items"]:
        got = orc.analyze_typo(typo, it["text"], 2.5, 0) if typo is not None else orc.analyze(it["text"])
        assert _rows(got[0][0]) == it["tokens"] and got[0][1] == it["score"], it["text"]


@pytest.mark.parametrize("name", FILES)
def test_emulated_device_equals_the_built_reference(name):
    import subprocess
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    assert check_device(os.path.join(emu, "_build", "libkiwi_hipemu.so"), name) >= 33            # every line


@pytest.mark.parametrize("lanes", ["pos", "16", "64"])
def test_emulated_kernel_variants_on_the_built_model(monkeypatch, lanes):
    """The real dictionary through the search kernels one by one -- the position-step kernel ("pos"), the general kernel in groups of 16 and 64 lanes --:
    top-1 and top-3 on the plain eval_data lines, typo correction top-1 and top-2 on the misspelt ones (built-in set basicTypoSetWithContinual entered
    rule by rule in the same order on both sides: the order of the rules decides which of exactly tied paths is kept), all against the oracle."""
    import subprocess
    import oraclelib
    import refbridge
    import test_typo_product
    from corpora import force_lanes
    from kiwi_amd.api import KiwiAmd
    from test_hipemu import _analyze_typo
    if not refbridge.available():
        pytest.skip("the built-in typo set is read out of oracle/_ref")
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    lib = os.path.join(emu, "_build", "libkiwi_hipemu.so")
    force_lanes(monkeypatch, lanes)
    path = built_model_path()
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=lib)

    def rows(res):
        return [([(t.form, t.tag, t.position, t.length, t.score, t.typo_cost) for t in toks], sc) for toks, sc in res]
    plain = [it["text"] for n in ("web", "written") for it in _golden(n)["items"]][::4]
    for top_n in (1, 3):
        for s, y in zip(plain, dev.analyze_batch(plain, top_n=top_n).to_python()):
            assert rows(orc.analyze(s, top_n=top_n)) == rows(y), (top_n, s)
    ents, cont, leng = refbridge.default_typo_entries("basic_with_continual")
    test_typo_product.LIB = lib
    prod = test_typo_product.ProductTypo(cont, leng)
    ot = oraclelib.OracleTypo(cont, leng)
    for orig, err, cost, cond, dialect in ents:
        prod.add_entry(orig, err, cost, cond, dialect)
    ot.update_entries(ents, cont, leng)
    prod.prepare(True); ot.prepare(True)
    misspelt = [it["text"] for it in _golden("web_with_typos")["items"]][::3]
    corrected = 0
    for top_n in (1, 2):
        for s, y in zip(misspelt, _analyze_typo(dev, prod, misspelt, 2.5, top_n)):
            want = orc.analyze_typo(ot, s, 2.5, 0, top_n=top_n)
            assert rows(want) == rows(y), (top_n, s)
            corrected += any(t.typo_cost > 0 for t in want[0][0])
    assert corrected >= 12
    dev.close(); prod.close()


@pytest.mark.skipif(not os.path.isdir("/root/reference/models/cong/base"), reason="needs the reference checkout (build container)")
def test_built_cong_model_reference_oracle_and_emulated_device(tmp_path, monkeypatch):
    """The same directory with a (synthetic, local) cong.mdl as its language model, the layout of the reference's models/cong/base: built by the real
    KiwiBuilder as ModelType::cong on its SSE4.1 dispatch, exported by tools/export_built.cpp, analysed by the oracle and the lane-emulated CoNgram
    kernels -- all three identical on the eval_data lines.  (The SkipBigram twin is not run: the synthetic skip-bigram tables let paths multiply on
    462-character lines until reference, oracle and device arenas alike take minutes per line -- DESIGN.md section 8, state arenas.)"""
    import json as _json
    import shutil
    import subprocess
    import sys
    from dataclasses import replace
    import oraclelib
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden_built as tool
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.synth import SMALL_CONG_SPEC, SynthModel
    from kiwi_amd.workloads import EVAL_BUILDER_REQUIRED
    monkeypatch.setenv("KIWI_ARCH_TYPE", "sse4_1")      # the reference's own switch (src/ArchUtils.cpp:105-117); the CoNgram pin is its SSE4.1 arithmetic
    entries = _json.load(open(os.path.join(HERE, "golden", "eval_data_lexicon.json"), encoding="utf-8"))["entries"]
    raw = os.path.join(ROOT, "_data", "small-eval-builder-cong.raw")
    if not os.path.exists(raw):
        SynthModel(replace(SMALL_CONG_SPEC, extra_words=tuple((f, t) for f, t in entries) + EVAL_BUILDER_REQUIRED)).raw.save(raw)
    lib, d = tool.shipped_dir(raw)
    try:
        assert "cong.mdl" in os.listdir(d)
        out = str(tmp_path / "built_cong.raw")
        monkeypatch.setattr(tool, "MODEL_TYPE", 4)      # ModelType::cong
        assert "15272 rule-combined" in tool.export(d, out)
        ref = refbridge.RefKiwi.built(d, 4, OPTIONS)
        orc = oraclelib.OracleKiwi(out)
        emu = os.path.join(HERE, "hipemu")
        subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
        dev = KiwiAmd(out, lib_path=os.path.join(emu, "_build", "libkiwi_hipemu.so"))

        def rows(res):
            return [([(t.form, t.tag, t.position, t.length, t.score) for t in toks], sc) for toks, sc in res]
        texts = [it["text"] for n in ("web", "written") for it in _golden(n)["items"]][::3]
        got = dev.analyze_batch(texts).to_python()
        for s, y in zip(texts, got):
            a = rows(ref.analyze(s))
            assert a == rows(orc.analyze(s)) == rows(y), s
        assert orc.counters()["congScores"] > 10000
        dev.close()
    finally:
        shutil.rmtree(d)


def test_match_options_on_the_built_model():
    """open ending and the Match bits that change lattices or candidate sets (normalisation off, splitComplex, zCoda off, splitSaisiot, mergeSaisiot) on the
    real dictionary: emulated device == oracle, and == the real reference analysing the same container where oracle/_ref is present."""
    import subprocess
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    path = built_model_path()
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=os.path.join(emu, "_build", "libkiwi_hipemu.so"))
    ref = refbridge.RefKiwi(path) if refbridge.available() else None

    def rows(res):
        return [([(t.form, t.tag, t.position, t.length, t.score) for t in toks], sc) for toks, sc in res]
    texts = [it["text"] for n in ("web", "written") for it in _golden(n)["items"]][::4]
    M = oraclelib.MATCH_ALL_WITH_NORMALIZING
    cases = [("open_ending", M, True), ("no normalisation", M & ~(1 << 16), False), ("splitComplex", M | (1 << 17), False), ("zCoda off", M & ~(1 << 23), False),
             ("splitSaisiot", M | (1 << 19), False), ("mergeSaisiot", M | (1 << 21), False)]
    base = [rows(orc.analyze(s)) for s in texts]
    for name, match, open_ending in cases:
        got = dev.analyze_batch(texts, match=match, open_ending=open_ending).to_python()
        want = [rows(orc.analyze(s, match=match, open_ending=open_ending)) for s in texts]
        assert [rows(y) for y in got] == want, name
        if ref is not None:
            assert [rows(ref.analyze(s, match=match, open_ending=open_ending)) for s in texts] == want, name
        if name == "open_ending":      # (sanity: the option reaches the search; the Match bits need text the sample may not hold)
            assert want != base, name
    dev.close()


def test_blocklist_on_the_built_model():
    """AnalyzeOption::blocklist with frequent real morphemes (kiwi_morphset_add = Kiwi::findMorphemes over the built dictionary, pre-analysed and combined
    morphemes included through Morpheme::hasMorpheme): emulated device == oracle for top-1 and top-2, == the real reference where oracle/_ref is present."""
    import subprocess
    import oraclelib
    import refbridge
    from corpora import pick_blocklist
    from kiwi_amd.api import KiwiAmd
    emu = os.path.join(HERE, "hipemu")
    subprocess.check_call(["make", "-C", emu, "-j8"], stdout=subprocess.DEVNULL)
    path = built_model_path()
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=os.path.join(emu, "_build", "libkiwi_hipemu.so"))

    def rows(res):
        return [([(t.form, t.tag, t.position, t.length, t.score) for t in toks], sc) for toks, sc in res]
    texts = [it["text"] for n in ("web", "written") for it in _golden(n)["items"]][::4]
    base = [rows(orc.analyze(s)) for s in texts]
    items = pick_blocklist(orc, texts, 25) + [("없는형태", 1)]
    ms, found = dev.morphset(items)
    assert found == orc.set_blocklist(items) and found[-1] == 0 and sum(found) >= 25
    for top_n in (1, 2):
        got = dev.analyze_batch_opt(texts, top_n=top_n, blocklist=ms).to_python()
        want = [rows(orc.analyze(s, top_n=top_n)) for s in texts]
        assert [rows(y) for y in got] == want, top_n
        if top_n == 1:
            assert sum(a != b for a, b in zip(want, base)) > len(texts) // 2      # the blocked morphemes were on most best paths
            if refbridge.available():
                ref = refbridge.RefKiwi(path)
                assert ref.set_blocklist(items) == found
                assert [rows(ref.analyze(s)) for s in texts] == want
    dev.close()
