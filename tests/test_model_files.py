"""The reference's on-disk model files: sj.morph (the reference's serializer, "KIWI" key + raw forms + raw
morphemes), sj.knlm and skipbigram.mdl (memory images).  The synthetic model is written in those formats BY THE REFERENCE'S OWN WRITER
(oracle/_ref: serializer::writeMany over FormRaw / MorphemeRaw), then loaded on both sides: the product's directory loader
(kiwi_amd/csrc/model.cpp loadModelDir, behind kamd_open / kiwi_init) and the reference's serializer::readMany + KnLangModelBase::create.
CPU part: the baked dictionary of the directory equals the baked dictionary of the raw container byte for byte, and analyses through the
emulated kernels equal the reference loading the same files.  GPU part: the same through kiwi_init(directory) on the device."""
import os
from dataclasses import astuple

import pytest

from corpora import dictionary_mix, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


def _model_dir(raw_path, name):
    """_data/<name>.files/: written by the reference where oracle/_ref can run; on the GPU box the directory travels with the repo."""
    import refbridge
    d = os.path.join(ROOT, "_data", name + ".files")
    if refbridge.available():
        refbridge.write_model_dir(raw_path, d)
    if not os.path.exists(os.path.join(d, "sj.morph")):
        pytest.skip("no model files (oracle/_ref not built)")
    return d


def test_model_files_are_what_the_loader_expects(small_model):
    """Format check independent of both loaders: the file the reference wrote parses by hand (struct) into the raw model's records."""
    import struct
    import numpy as np
    from kiwi_amd.container import read_container
    sm, path = small_model
    d = _model_dir(path, "small")
    b = open(os.path.join(d, "sj.morph"), "rb").read()
    assert b[:4] == b"KIWI"
    _, sec = read_container(path)
    n_forms, = struct.unpack_from("<I", b, 4)
    assert n_forms == int(np.frombuffer(sec["meta"], "<u4")[0])
    n0, = struct.unpack_from("<I", b, 8)                       # first form: u32 length + UTF-16 units
    fp = np.frombuffer(sec["form_ptr"], "<u4"); fc = np.frombuffer(sec["form_chars"], "<u2")
    assert n0 == fp[1] - fp[0] and list(struct.unpack_from("<%dH" % n0, b, 12)) == list(fc[fp[0]:fp[1]])
    assert open(os.path.join(d, "sj.knlm"), "rb").read() == sec["knlm"].tobytes()


@pytest.mark.parametrize("sbg", [False, True])
def test_directory_loader_equals_container_and_reference(small_model, small_sbg_model, sbg):
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    from kiwi_amd.api import KiwiAmd
    emu = os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated library not built")
    sm, path = small_sbg_model if sbg else small_model
    d = _model_dir(path, "small-sbg" if sbg else "small")
    a, b = KiwiAmd(path, lib_path=emu), KiwiAmd(d, lib_path=emu)
    assert a.dump_dict() == b.dump_dict()                      # same baked dictionary, byte for byte
    ref = refbridge.RefKiwi(d, model_dir_sbg=sbg)              # the reference reading the same files
    texts = synthetic(sm, 40, 711, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 20, 712)
    got = b.analyze_batch(texts).to_python()
    if not sbg:
        for s, y in zip(texts, got):
            assert _norm(ref.analyze(s)) == _norm(y), s
    else:
        # SkipBigram lattices live in the reference's large container, whose hand-on order of exactly tied paths depends on what the thread
        # analysed before (top-N / container order): the loaders are compared like with like -- the reference from the files
        # against the reference from the container over the same sequence, the product from the files against the product from the container
        ref_c = refbridge.RefKiwi(path)
        for s in texts:
            assert _norm(ref.analyze(s)) == _norm(ref_c.analyze(s)), s
        want = a.analyze_batch(texts).to_python()
        assert [_norm(x) for x in want] == [_norm(y) for y in got]
        for s, y in zip(texts, got):
            assert ref.analyze(s)[0][1] == y[0][1], s          # the best score is order-independent
    a.close(); b.close()


def test_quantised_knlm_file_in_a_model_directory(small_quantised_model):
    """A directory whose sj.knlm is quantised / compressed and / or history-transformed (what the reference's builder writes by default): the product's directory loader against the reference's own loading of the same files."""
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    from kiwi_amd.api import KiwiAmd
    emu = os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated library not built")
    sm, path, name = small_quantised_model
    d = _model_dir(path, name)
    import struct
    blob = open(os.path.join(d, "sj.knlm"), "rb").read()
    assert (struct.unpack_from("<B", blob, 91)[0] != 0) == ("q" in name)      # KnLangModelHeader::quantized
    assert (struct.unpack_from("<Q", blob, 48)[0] != 0) == ("htx" in name)     # KnLangModelHeader::htx_offset
    dev = KiwiAmd(d, lib_path=emu)
    ref = refbridge.RefKiwi(d, model_dir_sbg=False)
    texts = synthetic(sm, 40, 723, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 20, 724)
    for s, y in zip(texts, dev.analyze_batch(texts).to_python()):
        assert _norm(ref.analyze(s)) == _norm(y), s
    dev.close()


def _cong_dir(raw_path):
    """A directory like the reference's models/cong/base: sj.morph + cong.mdl and NO sj.knlm."""
    import shutil
    d = _model_dir(raw_path, "small-cong-src")
    out = os.path.join(ROOT, "_data", "small-cong.files")
    if os.path.exists(os.path.join(d, "cong.mdl")):
        os.makedirs(out, exist_ok=True)
        for f in ("sj.morph", "cong.mdl"):
            shutil.copyfile(os.path.join(d, f), os.path.join(out, f))
    if not os.path.exists(os.path.join(out, "cong.mdl")):
        pytest.skip("no model files (oracle/_ref not built)")
    return out


def test_cong_directory_loader_equals_container_and_reference(small_cong_model):
    """kamd_open(directory with sj.morph + cong.mdl, no sj.knlm -- the layout of the reference's models/cong/base): same baked dictionary as the
    container, CoNgram scoring selected, analyses equal the reference's SSE4.1 build loading the same two files (KiwiBuilder.cpp:939-1031)."""
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    from kiwi_amd.api import KiwiAmd
    emu = os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated library not built")
    sm, path = small_cong_model
    d = _cong_dir(path)
    assert sorted(os.listdir(d)) == ["cong.mdl", "sj.morph"]
    a, b = KiwiAmd(path, lib_path=emu), KiwiAmd(d, lib_path=emu)
    assert a.dump_dict() == b.dump_dict()
    ref = refbridge.RefKiwi(d, arch=3, model_dir_sbg=2, x86=True)
    texts = synthetic(sm, 40, 721, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 20, 722)
    got = b.analyze_batch(texts).to_python()
    want = a.analyze_batch(texts).to_python()
    for s, y, w in zip(texts, got, want):
        assert _norm(ref.analyze(s)) == _norm(y) == _norm(w), s
    a.close(); b.close()


def test_directory_with_a_character_model(small_cong_chr_model):
    """sj.morph + cong.mdl + nounchr.mdl (KiwiBuilder.cpp:1094-1100): the directory loader picks the character model up; analyses with
    Match::oovChrModel equal the reference's SSE4.1 build loading the same three files."""
    import shutil
    import refbridge
    if not refbridge.x86_available():
        pytest.skip("oracle/_ref/libkiwi_ref_x86.so not built")
    from kiwi_amd.api import KiwiAmd
    emu = os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated library not built")
    sm, path = small_cong_chr_model
    src = _model_dir(path, "small-cong-chr-src")
    d = os.path.join(ROOT, "_data", "small-cong-chr.files")
    os.makedirs(d, exist_ok=True)
    for f in ("sj.morph", "cong.mdl", "nounchr.mdl"):
        shutil.copyfile(os.path.join(src, f), os.path.join(d, f))
    match = refbridge.MATCH_ALL_WITH_NORMALIZING | (1 << 8)
    dev = KiwiAmd(d, lib_path=emu)
    ref = refbridge.RefKiwi(d, arch=3, model_dir_sbg=2, x86=True)
    texts = synthetic(sm, 40, 725, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 20, 726)
    for s, y in zip(texts, dev.analyze_batch(texts, match=match).to_python()):
        assert _norm(ref.analyze(s, match=match)) == _norm(y), s
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["knlm", "cong"])
def test_kiwi_init_on_a_model_directory(small_model, small_cong_model, kind):
    """kiwi_init(directory with sj.morph + sj.knlm, or with sj.morph + cong.mdl like the reference's models/cong/base) on the device, against
    the reference loading the same files."""
    import refbridge
    if not refbridge.available() or (kind == "cong" and not refbridge.x86_available()):
        pytest.skip("oracle/_ref not built")
    import ctypes as C
    from test_gpu_capi import LIB, Option, MATCH_ALL_WITH_NORMALIZING
    sm, path = small_model if kind == "knlm" else small_cong_model
    d = _model_dir(path, "small") if kind == "knlm" else _cong_dir(path)
    L = C.CDLL(LIB)
    L.kiwi_init.restype = C.c_void_p
    L.kiwi_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.kiwi_analyze.restype = C.c_void_p
    L.kiwi_analyze.argtypes = [C.c_void_p, C.c_char_p, C.c_int, Option, C.c_void_p]
    L.kiwi_res_prob.restype = C.c_float
    L.kiwi_res_prob.argtypes = [C.c_void_p, C.c_int]
    L.kiwi_res_word_num.argtypes = [C.c_void_p, C.c_int]
    L.kiwi_res_form.restype = C.c_char_p
    L.kiwi_res_form.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_res_close.argtypes = [C.c_void_p]
    L.kiwi_close.argtypes = [C.c_void_p]
    L.kiwi_error.restype = C.c_char_p
    k = L.kiwi_init(d.encode(), 0, 15, 0)
    assert k, L.kiwi_error()
    ref = refbridge.RefKiwi(d, model_dir_sbg=False) if kind == "knlm" else refbridge.RefKiwi(d, arch=3, model_dir_sbg=2, x86=True)
    opt = Option(MATCH_ALL_WITH_NORMALIZING, None, 0, 0, 3.0, None, 2.5)
    for s in synthetic(sm, 60, 713, min_jamo=5, max_jamo=80):
        r = L.kiwi_analyze(k, s.encode("utf-8"), 1, opt, None)
        assert r, L.kiwi_error()
        want = ref.analyze(s)
        assert L.kiwi_res_prob(r, 0) == want[0][1] and L.kiwi_res_word_num(r, 0) == len(want[0][0]), s
        assert [L.kiwi_res_form(r, 0, j).decode("utf-8") for j in range(len(want[0][0]))] == [t.form for t in want[0][0]], s
        L.kiwi_res_close(r)
    L.kiwi_close(k)


def test_a_directory_with_the_builders_text_inputs_is_refused(small_model, tmp_path):
    """A real Kiwi model directory also holds combiningRule.txt (+ .dict files), which the reference's KiwiBuilder consumes at build time and this
    library does not: loading it is refused loudly (never a silently different dictionary), unless KAMD_ALLOW_UNEXPANDED_MODEL is set."""
    import shutil
    import subprocess
    import sys
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    emu = os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated library not built")
    src = _model_dir(small_model[1], "small")
    d = str(tmp_path / "real-like")
    shutil.copytree(src, d)
    open(os.path.join(d, "combiningRule.txt"), "w").write("# rules\n")
    code = ("import sys; sys.path.insert(0, %r); from kiwi_amd.api import KiwiAmd\n"
            "try:\n    KiwiAmd(%r, lib_path=%r); print('LOADED')\nexcept Exception as e:\n    print('REFUSED', 'combiningRule.txt' in str(e))\n") % (ROOT, d, emu)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "KAMD_ALLOW_UNEXPANDED_MODEL"}).stdout
    assert "REFUSED True" in out, out
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, KAMD_ALLOW_UNEXPANDED_MODEL="1")).stdout
    assert "LOADED" in out, out


def test_knlm_model_type_is_refused_on_a_directory_without_sj_knlm(small_cong_model):
    """kiwi_init(models/cong/base layout, KIWI_BUILD_MODEL_TYPE_KNLM): the reference cannot open sj.knlm there and fails; so does kiwi_init here --
    NULL plus kiwi_error() -- instead of searching with empty LM tables."""
    import ctypes as C
    emu = os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(emu):
        pytest.skip("emulated library not built")
    sm, path = small_cong_model
    d = _cong_dir(path)
    L = C.CDLL(emu)
    L.kiwi_init.restype = C.c_void_p
    L.kiwi_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.kiwi_error.restype = C.c_char_p
    L.kiwi_close.argtypes = [C.c_void_p]
    for model_type in (0x0200, 0x0300):      # KIWI_BUILD_MODEL_TYPE_KNLM, _SBG
        L.kiwi_clear_error()
        assert not L.kiwi_init(d.encode(), 0, 15 | model_type, 0)
        assert b"sj.knlm" in L.kiwi_error() or b"skipbigram" in L.kiwi_error().lower()
    h = L.kiwi_init(d.encode(), 0, 15 | 0x0400, 0)      # KIWI_BUILD_MODEL_TYPE_CONG still opens
    assert h
    L.kiwi_close(h)
