"""GPU (-m gpu): the drop-in boundary itself -- Kiwi's own C API (include/kiwi_capi.h <- reference include/kiwi/capi.h) bound
with ctypes exactly as a client of the reference would, against the CPU oracle: kiwi_init, kiwi_analyze{,_w,_m,_mw} with
reader / receiver callbacks in input order, the kiwi_res_* accessors, config / option setters, error convention."""
import ctypes as C
import os

import numpy as np
import pytest

from corpora import EDGE_TEXTS, dictionary_mix, synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
# KAMD_TEST_LIB: tests/test_hipemu.py re-runs this file on the CPU against the lane-emulated build of the same sources
LIB = os.environ.get("KAMD_TEST_LIB") or os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip.so")
MATCH_ALL_WITH_NORMALIZING = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 5) | (1 << 23) | (1 << 16)


class TokenInfo(C.Structure):      # kiwi_token_info_t (capi.h:43-61)
    _fields_ = [("chr_position", C.c_uint32), ("word_position", C.c_uint32), ("sent_position", C.c_uint32), ("line_number", C.c_uint32),
                ("length", C.c_uint16), ("tag", C.c_uint8), ("sense_id", C.c_uint8), ("score", C.c_float), ("typo_cost", C.c_float),
                ("typo_form_id", C.c_uint32), ("paired_token", C.c_uint32), ("sub_sent_position", C.c_uint32), ("dialect", C.c_uint16)]


class Config(C.Structure):         # kiwi_config_t (capi.h:72-86)
    _fields_ = [("integrate_allomorph", C.c_uint8), ("cut_off_threshold", C.c_float), ("oov_rule_scale", C.c_float), ("oov_rule_bias", C.c_float),
                ("oov_chr_bias", C.c_float), ("oov_global_weight", C.c_float), ("oov_local_weight", C.c_float), ("oov_global_min_freq", C.c_float),
                ("space_penalty", C.c_float), ("typo_cost_weight", C.c_float),
                ("max_unk_form_size", C.c_uint32), ("max_unk_form_size_followed_by_j_class", C.c_uint32), ("space_tolerance", C.c_uint32)]


class Option(C.Structure):         # kiwi_analyze_option_t (capi.h:662-670)
    _fields_ = [("match_options", C.c_int), ("blocklist", C.c_void_p), ("open_ending", C.c_int), ("allowed_dialects", C.c_int),
                ("dialect_cost", C.c_float), ("typo_transformer", C.c_void_p), ("typo_threshold", C.c_float)]


READER = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p)
READER_W = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p)
RECEIVER = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p)


@pytest.fixture(scope="module")
def capi():
    L = C.CDLL(LIB)
    L.kiwi_error.restype = C.c_char_p
    L.kiwi_version.restype = C.c_char_p
    L.kiwi_init.restype = C.c_void_p
    L.kiwi_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.kiwi_close.argtypes = [C.c_void_p]
    L.kiwi_set_global_config.argtypes = [C.c_void_p, Config]
    L.kiwi_get_global_config.restype = Config
    L.kiwi_get_global_config.argtypes = [C.c_void_p]
    L.kiwi_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_get_option.argtypes = [C.c_void_p, C.c_int]
    L.kiwi_analyze.restype = C.c_void_p
    L.kiwi_analyze.argtypes = [C.c_void_p, C.c_char_p, C.c_int, Option, C.c_void_p]
    L.kiwi_analyze_w.restype = C.c_void_p
    L.kiwi_analyze_w.argtypes = [C.c_void_p, C.c_void_p, C.c_int, Option, C.c_void_p]
    L.kiwi_analyze_m.argtypes = [C.c_void_p, READER, RECEIVER, C.c_void_p, C.c_int, Option]
    L.kiwi_analyze_mw.argtypes = [C.c_void_p, READER_W, RECEIVER, C.c_void_p, C.c_int, Option]
    for f in ("kiwi_res_size", "kiwi_res_close"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.kiwi_res_prob.restype = C.c_float
    L.kiwi_res_prob.argtypes = [C.c_void_p, C.c_int]
    L.kiwi_res_word_num.argtypes = [C.c_void_p, C.c_int]
    L.kiwi_res_token_info.restype = C.POINTER(TokenInfo)
    L.kiwi_res_token_info.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_res_morpheme_id.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.kiwi_res_form.restype = C.c_char_p
    L.kiwi_res_form.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_res_tag.restype = C.c_char_p
    L.kiwi_res_tag.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_res_form_w.restype = C.POINTER(C.c_uint16)
    L.kiwi_res_form_w.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for f in ("kiwi_res_position", "kiwi_res_length", "kiwi_res_word_position", "kiwi_res_sent_position"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_res_score.restype = C.c_float
    L.kiwi_res_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.kiwi_tag_to_string.restype = C.c_char_p
    L.kiwi_tag_to_string.argtypes = [C.c_void_p, C.c_uint8]
    return L


def opt(match=MATCH_ALL_WITH_NORMALIZING, open_ending=0):
    return Option(match, None, open_ending, 0, 0.0, None, 0.0)


def read_result(L, k, r):
    """kiwi_res_h -> [([(form, tag id, position, length, word pos, sent pos, score, morph id)], prob)]"""
    out = []
    for i in range(L.kiwi_res_size(r)):
        toks = []
        for j in range(L.kiwi_res_word_num(r, i)):
            ti = L.kiwi_res_token_info(r, i, j).contents
            form = L.kiwi_res_form(r, i, j).decode("utf-8")
            assert L.kiwi_res_position(r, i, j) == ti.chr_position and L.kiwi_res_length(r, i, j) == ti.length
            assert L.kiwi_res_word_position(r, i, j) == ti.word_position and L.kiwi_res_sent_position(r, i, j) == ti.sent_position
            assert L.kiwi_res_score(r, i, j) == ti.score
            assert L.kiwi_res_tag(r, i, j) == L.kiwi_tag_to_string(k, ti.tag)
            fw = L.kiwi_res_form_w(r, i, j)
            n = 0
            while fw[n]:
                n += 1
            assert np.array(fw[:n], np.uint16).tobytes().decode("utf-16-le", errors="surrogatepass") == form
            toks.append((form, ti.tag, ti.chr_position, ti.length, ti.word_position, ti.sent_position, ti.score, L.kiwi_res_morpheme_id(r, i, j, k)))
        out.append((toks, L.kiwi_res_prob(r, i)))
    return out


def from_oracle(res):
    return [([(t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.score, t.morph_id) for t in a[0]], a[1]) for a in res]


@pytest.fixture(scope="module")
def kiwi(capi, small_model):
    k = capi.kiwi_init(small_model[1].encode(), 0, 15, 0)      # KIWI_BUILD_DEFAULT
    assert k, capi.kiwi_error()
    yield k
    assert capi.kiwi_close(k) == 0


def test_version_and_init_errors(capi):
    assert capi.kiwi_version()
    assert not capi.kiwi_init(b"/nonexistent/model", 0, 0, 0)
    assert capi.kiwi_error()


def test_analyze_utf8_and_utf16(capi, kiwi, oracle, small_model):
    sm, _ = small_model
    for s in synthetic(sm, 40, 161, min_jamo=5, max_jamo=100) + [t for t in EDGE_TEXTS if t.strip() and "\x00" not in t][:30]:
        try:
            s.encode("utf-8")
        except UnicodeEncodeError:
            continue                     # lone surrogates have no UTF-8 form: covered by the UTF-16 entry point below
        for top_n in (1, 2):
            r = capi.kiwi_analyze(kiwi, s.encode("utf-8"), top_n, opt(), None)
            assert r, capi.kiwi_error()
            assert read_result(capi, kiwi, r) == from_oracle(oracle.analyze(s, top_n=top_n)), s
            assert capi.kiwi_res_close(r) == 0
        u = np.frombuffer((s + "\0").encode("utf-16-le", errors="surrogatepass"), np.uint16).copy()
        r = capi.kiwi_analyze_w(kiwi, u.ctypes.data, 1, opt(), None)
        assert r, capi.kiwi_error()
        assert read_result(capi, kiwi, r) == from_oracle(oracle.analyze(s)), s
        capi.kiwi_res_close(r)


def test_analyze_many_delivers_in_input_order(capi, kiwi, oracle, small_model):
    sm, _ = small_model
    texts = [t for t in synthetic(sm, 300, 162, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 100, 163)]
    enc = [t.encode("utf-8") for t in texts]
    got = []

    def reader(i, buf, ud):
        if i >= len(enc):
            return 0
        if not buf:
            return len(enc[i])                # capi.h:88-104: a null buffer asks for the size; 0 ends the stream
        C.memmove(buf, enc[i], len(enc[i]))
        return 0

    def receiver(i, r, ud):
        got.append((i, read_result(capi, kiwi, r)))
        capi.kiwi_res_close(r)
        return 0

    # the reference's own C test (test/test_c.cpp:48-55) counts and copies the terminator as part of each line
    with_nul = []

    def reader_nul(i, buf, ud):
        if i >= 50:
            return 0
        if not buf:
            return len(enc[i]) + 1
        C.memmove(buf, enc[i] + b"\0", len(enc[i]) + 1)
        return 0

    def receiver_nul(i, r, ud):
        with_nul.append(read_result(capi, kiwi, r))
        capi.kiwi_res_close(r)
        return 0

    assert capi.kiwi_analyze_m(kiwi, READER(reader_nul), RECEIVER(receiver_nul), None, 1, opt()) == 50
    for y, s in zip(with_nul, texts):
        want = [([(t[0].split("\0")[0],) + t[1:] for t in a[0]], a[1]) for a in from_oracle(oracle.analyze(s + "\0"))]   # C strings end at the NUL
        assert y == want, s

    capi.kiwi_set_option(kiwi, 0x9001, 128)   # KIWI_GPU_BATCH_SIZE: several device batches
    assert capi.kiwi_get_option(kiwi, 0x9001) == 128
    n = capi.kiwi_analyze_m(kiwi, READER(reader), RECEIVER(receiver), None, 1, opt())
    assert n == len(texts), capi.kiwi_error()
    assert [i for i, _ in got] == list(range(len(texts)))
    for (i, y), s in zip(got, texts):
        assert y == from_oracle(oracle.analyze(s)), s


def test_analyze_many_delivers_the_parts_of_a_batch_in_input_order(capi, kiwi, oracle, small_model, monkeypatch):
    """kiwi_analyze_m over ONE device batch that the engine cuts into parts (Engine::analyzeBatch: stage / launch on one host thread, wait / download /
    assembly on another): the parts' results are handed to the calling thread while the later parts are still on the device and delivered to the
    receiver in input order -- same analyses as the oracle's, line by line, with texts of several chunks among them."""
    sm, _ = small_model
    monkeypatch.setenv("KAMD_BATCH_PARTS", "3")
    texts = [t for t in synthetic(sm, 420, 171, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 60, 172)]
    texts += [". ".join(texts[k:k + 4]) + "." for k in range(0, 40, 4)]
    enc = [t.encode("utf-8") for t in texts]
    got = []

    def reader(i, buf, ud):
        if i >= len(enc):
            return 0
        if not buf:
            return len(enc[i])
        C.memmove(buf, enc[i], len(enc[i]))
        return 0

    def receiver(i, r, ud):
        got.append((i, read_result(capi, kiwi, r)))
        capi.kiwi_res_close(r)
        return 0

    capi.kiwi_set_option(kiwi, 0x9001, 65536)   # KIWI_GPU_BATCH_SIZE: everything in one device batch
    n = capi.kiwi_analyze_m(kiwi, READER(reader), RECEIVER(receiver), None, 1, opt())
    assert n == len(texts), capi.kiwi_error()
    assert [i for i, _ in got] == list(range(len(texts)))
    for (i, y), s in zip(got, texts):
        assert y == from_oracle(oracle.analyze(s)), s


def test_concurrent_callers_on_one_handle(capi, kiwi, oracle, small_model):
    """kiwi_analyze* is callable from many threads on one handle (reference capi threading contract): device work is
    serialised per engine, results are unaffected."""
    import threading
    sm, _ = small_model
    texts = synthetic(sm, 160, 181, min_jamo=5, max_jamo=80)
    want = [from_oracle(oracle.analyze(s)) for s in texts]
    got = [None] * len(texts)
    errs = []

    def work(t):
        try:
            for i in range(t, len(texts), 8):
                r = capi.kiwi_analyze(kiwi, texts[i].encode("utf-8"), 1, opt(), None)
                assert r, capi.kiwi_error()
                got[i] = read_result(capi, kiwi, r)
                capi.kiwi_res_close(r)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert got == want


def test_config_roundtrip_and_refusals(capi, kiwi, oracle):
    cfg = capi.kiwi_get_global_config(kiwi)
    assert cfg.cut_off_threshold == 8.0 and cfg.space_penalty == 7.0
    assert (cfg.oov_global_weight, cfg.oov_local_weight, cfg.oov_global_min_freq) == (35.0, 3.0, 4.0)      # include/kiwi/Kiwi.h:157-159
    cfg.oov_global_weight, cfg.oov_local_weight, cfg.oov_global_min_freq = 60.0, 1.5, 1.0
    capi.kiwi_set_global_config(kiwi, cfg)
    back = capi.kiwi_get_global_config(kiwi)
    assert (back.oov_global_weight, back.oov_local_weight, back.oov_global_min_freq) == (60.0, 1.5, 1.0)
    cfg.oov_global_weight, cfg.oov_local_weight, cfg.oov_global_min_freq = 35.0, 3.0, 4.0
    capi.kiwi_set_global_config(kiwi, cfg)
    s = "가나다라 마바사"
    try:
        cfg.cut_off_threshold = 5.0
        capi.kiwi_set_global_config(kiwi, cfg)
        oracle.set_config(cut_off=5.0)
        r = capi.kiwi_analyze(kiwi, s.encode(), 1, opt(), None)
        assert read_result(capi, kiwi, r) == from_oracle(oracle.analyze(s))
        capi.kiwi_res_close(r)
    finally:
        cfg.cut_off_threshold = 8.0
        capi.kiwi_set_global_config(kiwi, cfg)
        oracle.set_config()
    assert not capi.kiwi_analyze(kiwi, s.encode(), 17, opt(), None)         # top_n beyond the device limit (16): refused, not ignored
    assert capi.kiwi_error()
    # allowed_dialects / dialect_cost on a model without dialect morphemes: no morpheme is skipped or charged, but -- as in the reference -- the analysis is
    # corrected with the built-in `dialect` typo set (tests/test_dialect.py has the models WITH dialect morphemes)
    o = opt()
    o.allowed_dialects, o.dialect_cost = 0x3FF, 3.0
    r = capi.kiwi_analyze(kiwi, s.encode(), 1, o, None)
    assert r and read_result(capi, kiwi, r) == from_oracle(oracle.analyze_dialect(s, 0x3FF, 3.0))
    capi.kiwi_res_close(r)
    assert not capi.kiwi_init(b"/nonexistent", 0, 15, 1) and capi.kiwi_error()      # (no such model; enabled dialects themselves are accepted)
    # pretokenized objects can be built and closed; without spans they constrain nothing; with a span the analysis honours it (round 6: the whole argument is
    # tests/test_gpu_pretokenized.py's) -- what stays refused is a span together with a typo transformer
    import ctypes as C
    capi.kiwi_pt_init.restype = C.c_void_p
    capi.kiwi_pt_add_span.argtypes = [C.c_void_p, C.c_int, C.c_int]
    capi.kiwi_pt_add_token_to_span.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    capi.kiwi_pt_close.argtypes = [C.c_void_p]
    pt = capi.kiwi_pt_init()
    assert pt
    r = capi.kiwi_analyze(kiwi, s.encode(), 1, opt(), pt)
    assert r and read_result(capi, kiwi, r) == from_oracle(oracle.analyze(s))
    capi.kiwi_res_close(r)
    assert capi.kiwi_pt_add_span(pt, 0, 3) == 0 and capi.kiwi_pt_add_token_to_span(pt, 0, "가나다".encode(), b"NNP", 0, 3) == 0
    assert capi.kiwi_pt_add_token_to_span(pt, 5, "가".encode(), b"NNP", 0, 1) != 0
    r = capi.kiwi_analyze(kiwi, s.encode(), 1, opt(), pt)
    assert r, capi.kiwi_error()
    first = capi.kiwi_res_token_info(r, 0, 0).contents
    assert (capi.kiwi_res_form(r, 0, 0).decode(), capi.kiwi_res_tag(r, 0, 0), first.chr_position, first.typo_form_id) == ("가나다", b"NNP", 0, 1)
    capi.kiwi_res_close(r)
    capi.kiwi_typo_get_default.restype = C.c_void_p
    capi.kiwi_typo_get_default.argtypes = [C.c_int]
    o = opt()
    o.typo_transformer = capi.kiwi_typo_get_default(1)      # KIWI_TYPO_BASIC_TYPO_SET
    o.typo_threshold = 2.5
    assert o.typo_transformer and not capi.kiwi_analyze(kiwi, s.encode(), 1, o, pt) and b"pretokenized" in capi.kiwi_error()
    assert capi.kiwi_pt_close(pt) == 0


@pytest.mark.parametrize("header", ["kiwi_capi.h", "reference capi.h"])
def test_c_client_program(oracle, small_model, tmp_path, header):
    """a plain C program written against the C API header only is compiled with gcc, linked to the
    library and run on a corpus; its printed tokens equal the oracle's.  Once against this repo's header, once against the REFERENCE's
    own include/kiwi/capi.h (the actual drop-in claim): that binary is built where /root/reference exists (tests/c_client/Makefile,
    __graft_entry__.build()) and travels to the GPU box."""
    import subprocess
    sm, path = small_model
    root = os.path.dirname(HERE)
    if header == "kiwi_capi.h":
        exe = str(tmp_path / "kiwi_client")
        subprocess.check_call(["gcc", "-std=c99", "-D_GNU_SOURCE", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(HERE, "c_client", "client.c"),
                               LIB, "-Wl,-rpath," + os.path.dirname(LIB), "-o", exe])
    else:
        exe = os.path.join(HERE, "c_client", "_build", "client_refhdr")
        if not os.path.exists(exe):
            pytest.skip("tests/c_client/_build/client_refhdr not built (needs /root/reference/include/kiwi/capi.h)")
        if os.environ.get("KAMD_TEST_LIB"):
            pytest.skip("the prebuilt binary is linked with the product library, not with the library under test")
    texts = [t for t in synthetic(sm, 200, 171, min_jamo=5, max_jamo=100) if "\n" not in t and "\r" not in t and t.strip()]
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join(texts) + "\n", encoding="utf-8")
    out = subprocess.run([exe, path, str(corpus)], check=True, capture_output=True).stdout.decode("utf-8")
    want = []
    tag_names = None
    for i, t in enumerate(texts):
        res = oracle.analyze(t)
        for tok in res[0][0]:
            want.append((i, tok.form, tok.position, tok.length, "%.9g" % tok.score))
        want.append((i, "#", "%.9g" % res[0][1]))
    got = []
    for ln in out.splitlines():
        f = ln.split("\t")
        if f[1] == "#":
            got.append((int(f[0]), "#", f[2]))
        else:
            got.append((int(f[0]), f[1], int(f[3]), int(f[4]), f[5]))
    assert got == want


def test_typo_transformer_through_the_c_api(capi, kiwi, small_model):
    """kiwi_typo_init / _add / _copy / _update / _scale_cost / _set_*_cost / _prepare and kiwi_analyze with option.typo_transformer, as a client of the
    reference would call them (capi.h:459-588, 662-698), against the oracle with the same rules: tokens, scores, kiwi_res_typo_cost."""
    import random
    import oraclelib
    from typo_cases import COND, RULES, misspell
    sm, path = small_model
    L = capi
    L.kiwi_typo_init.restype = C.c_void_p
    L.kiwi_typo_copy.restype = C.c_void_p
    L.kiwi_typo_copy.argtypes = [C.c_void_p]
    L.kiwi_typo_get_default.restype = C.c_void_p
    L.kiwi_typo_get_basic.restype = C.c_void_p
    L.kiwi_typo_add.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_float, C.c_int]
    L.kiwi_typo_update.argtypes = [C.c_void_p, C.c_void_p]
    L.kiwi_typo_scale_cost.argtypes = [C.c_void_p, C.c_float]
    L.kiwi_typo_set_continual_typo_cost.argtypes = [C.c_void_p, C.c_float]
    L.kiwi_typo_set_lengthening_typo_cost.argtypes = [C.c_void_p, C.c_float]
    L.kiwi_typo_close.argtypes = [C.c_void_p]
    L.kiwi_typo_prepare.restype = C.c_void_p
    L.kiwi_typo_prepare.argtypes = [C.c_void_p]
    L.kiwi_prepared_typo_close.argtypes = [C.c_void_p]
    L.kiwi_res_typo_cost.restype = C.c_float
    L.kiwi_res_typo_cost.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert not L.kiwi_typo_get_default(7) and b"DefaultTypoSet" in L.kiwi_error()
    part, whole = L.kiwi_typo_init(), L.kiwi_typo_init()
    orc_t = oraclelib.OracleTypo(1.0, 0.25)
    for origs, errs, cost, cond, dia in RULES:
        if dia:
            continue          # (the C API's kiwi_typo_add has no dialect argument)
        o = (C.c_char_p * len(origs))(*[x.encode() for x in origs])
        e = (C.c_char_p * len(errs))(*[x.encode() for x in errs])
        assert L.kiwi_typo_add(part, o, len(origs), e, len(errs), cost / 2, COND[cond]) == 0      # half the cost here, doubled by scale_cost below
        for a in origs:
            for b in errs:
                orc_t.add(a, b, cost, COND[cond], 0)
    assert L.kiwi_typo_scale_cost(part, 2.0) == 0 and L.kiwi_typo_scale_cost(part, -1.0) == -1
    assert L.kiwi_typo_update(whole, part) == 0
    copy = L.kiwi_typo_copy(whole)
    assert L.kiwi_typo_set_continual_typo_cost(copy, 1.0) == 0 and L.kiwi_typo_set_lengthening_typo_cost(copy, 0.25) == 0
    prepared = L.kiwi_typo_prepare(copy)
    assert prepared
    orc_t.prepare(True)
    orc = oraclelib.OracleKiwi(path)
    rnd = random.Random(21)
    opt = Option(MATCH_ALL_WITH_NORMALIZING, None, 0, 0, 3.0, prepared, 2.5)
    corrected = 0
    for t in [misspell(x, rnd, True, True, True) for x in synthetic(sm, 60, 181, min_jamo=5, max_jamo=80)]:
        r = L.kiwi_analyze(kiwi, t.encode("utf-8"), 1, opt, None)
        assert r, L.kiwi_error()
        want = orc.analyze_typo(orc_t, t, 2.5, 0)
        toks = want[0][0]
        assert L.kiwi_res_word_num(r, 0) == len(toks) and L.kiwi_res_prob(r, 0) == want[0][1], t
        for k, tok in enumerate(toks):
            assert L.kiwi_res_form(r, 0, k).decode("utf-8") == tok.form and L.kiwi_res_typo_cost(r, 0, k) == tok.typo_cost, (t, k)
            corrected += tok.typo_cost > 0
        L.kiwi_res_close(r)
    assert corrected > 10
    L.kiwi_prepared_typo_close(prepared)
    for h in (part, whole, copy):
        assert L.kiwi_typo_close(h) == 0
    # the built-in sets (kiwi_typo_get_default; the reference's --typo configurations): against the REAL reference analysing with its own set
    import refbridge
    if refbridge.available():
        ref = refbridge.RefKiwi(path)
        for name, sid in (("basic", 1), ("basic_with_continual_and_lengthening", 5)):
            builtin = L.kiwi_typo_get_default(sid)
            assert builtin, L.kiwi_error()
            assert builtin == (L.kiwi_typo_get_basic() if sid == 1 else L.kiwi_typo_get_default(sid))      # one static object per set
            prepared = L.kiwi_typo_prepare(builtin)
            rt = refbridge.RefTypo.from_default(name); rt.prepare(True)
            opt = Option(MATCH_ALL_WITH_NORMALIZING, None, 0, 0, 3.0, prepared, 2.5)
            corrected = 0
            for t in [misspell(x, rnd, True, True, sid == 5) for x in synthetic(sm, 60, 183, min_jamo=5, max_jamo=80)]:
                r = L.kiwi_analyze(kiwi, t.encode("utf-8"), 1, opt, None)
                assert r, L.kiwi_error()
                want = ref.analyze_typo(rt, t, 2.5, 0)
                toks = want[0][0]
                assert L.kiwi_res_word_num(r, 0) == len(toks) and L.kiwi_res_prob(r, 0) == want[0][1], (name, t)
                for k, tok in enumerate(toks):
                    assert L.kiwi_res_form(r, 0, k).decode("utf-8") == tok.form and L.kiwi_res_typo_cost(r, 0, k) == tok.typo_cost, (name, t, k)
                    corrected += tok.typo_cost > 0
                L.kiwi_res_close(r)
            assert corrected > 5, name
            L.kiwi_prepared_typo_close(prepared)


def test_blocklist_through_the_c_api(capi, kiwi, oracle, small_model):
    """kiwi_new_morphset / kiwi_morphset_add / _add_w / _close and kiwi_analyze* with option.blocklist, as a client of the reference calls them
    (capi.h:655-664, 1243-1263), against the oracle with the same list (pinned to the real reference on the CPU): tokens, positions, scores."""
    from corpora import pick_blocklist
    sm, path = small_model
    L = capi
    L.kiwi_new_morphset.restype = C.c_void_p
    L.kiwi_new_morphset.argtypes = [C.c_void_p]
    L.kiwi_morphset_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.kiwi_morphset_add_w.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    L.kiwi_morphset_close.argtypes = [C.c_void_p]
    texts = synthetic(sm, 120, 941, min_jamo=5, max_jamo=100)
    items = pick_blocklist(oracle, texts, 16)
    ms = L.kiwi_new_morphset(kiwi)
    assert ms
    try:
        want_found = oracle.set_blocklist(items)
        for n, ((form, tag), wf) in enumerate(zip(items, want_found)):
            tag_s = None if tag < 0 else L.kiwi_tag_to_string(kiwi, tag)
            if n % 2:
                u = np.frombuffer((form + "\0").encode("utf-16-le"), np.uint16).copy()
                got = L.kiwi_morphset_add_w(ms, u.ctypes.data, tag_s)
            else:
                got = L.kiwi_morphset_add(ms, form.encode("utf-8"), tag_s.lower() if tag_s else None)      # (tags are case-insensitive)
            assert got == wf, (form, tag)
        assert L.kiwi_morphset_add(ms, "없는형태".encode("utf-8"), b"NNG") == 0
        assert L.kiwi_morphset_add(ms, texts[0][:1].encode("utf-8"), b"NOTATAG") < 0 and b"Unknown POSTag" in L.kiwi_error()
        o = Option(MATCH_ALL_WITH_NORMALIZING, ms, 0, 0, 0.0, None, 0.0)
        changed = 0
        for t in texts:
            r = L.kiwi_analyze(kiwi, t.encode("utf-8"), 2, o, None)
            assert r, L.kiwi_error()
            assert read_result(L, kiwi, r) == from_oracle(oracle.analyze(t, top_n=2)), t
            L.kiwi_res_close(r)
        # through the batch driver as well, and unconstrained again without the option
        got = {}
        enc = [t.encode("utf-8") for t in texts]

        def reader(idx, buf, ud):
            if idx >= len(enc):
                return 0
            if not buf:
                return len(enc[idx])
            C.memmove(buf, enc[idx], len(enc[idx]))
            return 0

        def receiver(idx, res, ud):
            got[idx] = read_result(L, kiwi, res)
            L.kiwi_res_close(res)
            return 0

        rd, rc = READER(reader), RECEIVER(receiver)
        assert L.kiwi_analyze_m(kiwi, rd, rc, None, 1, o) == len(texts)
        for i, t in enumerate(texts):
            assert got[i] == from_oracle(oracle.analyze(t)), t
        oracle.set_blocklist([])
        for t in texts[:40]:
            r = L.kiwi_analyze(kiwi, t.encode("utf-8"), 1, opt(), None)
            res = read_result(L, kiwi, r)
            L.kiwi_res_close(r)
            assert res == from_oracle(oracle.analyze(t)), t
            changed += res != got[texts.index(t)]
        assert changed >= 10
    finally:
        oracle.set_blocklist([])
        assert L.kiwi_morphset_close(ms) == 0


def test_dialects_through_the_c_api(capi):
    """kiwi_init(..., enabled_dialects) and kiwi_analyze with the reference's kiwi_analyze_option_t (allowed_dialects, dialect_cost) on the model with dialect
    morphemes (kiwi_amd.workloads.dialect_model): tokens, scores and the dialect field of kiwi_token_info_t as the REAL reference answered
    (tests/golden/eval_dialect.json, tools/make_golden_dialect.py)."""
    import json
    from kiwi_amd.workloads import dialect_model
    items = json.load(open(os.path.join(HERE, "golden", "eval_dialect.json"), encoding="utf-8"))["items"]
    k = capi.kiwi_init(dialect_model().encode(), 1, 1 | 0x0200, 1023)      # integrate allomorphs, Knlm; every dialect enabled
    assert k, capi.kiwi_error()
    try:
        for it in items[::9]:
            for allowed, cost, key in ((0, 3.0, "allowed_0"), (it["bit"], 3.0, "allowed_own"), (1023, 1.5, "allowed_all_cost_1.5")):
                o = opt()
                o.allowed_dialects, o.dialect_cost = allowed, cost
                r = capi.kiwi_analyze(k, it["text"].encode(), 1, o, None)
                assert r, capi.kiwi_error()
                want = it["enabled_all"][key]
                assert capi.kiwi_res_word_num(r, 0) == len(want["tokens"]) and capi.kiwi_res_prob(r, 0) == C.c_float(want["score"]).value, (it["text"], key)
                for j, w in enumerate(want["tokens"]):
                    ti = capi.kiwi_res_token_info(r, 0, j).contents
                    assert (capi.kiwi_res_form(r, 0, j).decode(), ti.tag, ti.chr_position, ti.length, ti.dialect) == (w[0], w[1], w[2], w[3], w[9]), (it["text"], key, j)
                    assert ti.score == C.c_float(w[7]).value and ti.typo_cost == C.c_float(w[8]).value
                capi.kiwi_res_close(r)
    finally:
        assert capi.kiwi_close(k) == 0
