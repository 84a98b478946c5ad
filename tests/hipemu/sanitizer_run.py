"""Run by tests/test_hipemu.py in a subprocess with a sanitizer runtime preloaded: a small corpus through the sanitizer build of the
emulated library (Knlm with KAMD_TEST_TINY_ARENAS, i.e. with most chunks hitting a capacity limit first; the SkipBigram kernel; the
typo-correcting analysis).
Any report of the sanitizer ends the process with a non-zero status."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))      # (test_hipemu imports the oracle wrappers at module level; nothing of the oracle runs here)
from corpora import EDGE_TEXTS, dictionary_mix, synthetic   # noqa: E402
from kiwi_amd.api import KiwiAmd                              # noqa: E402
from kiwi_amd.synth import SMALL_CONG_CHR_SPEC, SMALL_SBG_SPEC, SMALL_SPEC, SynthModel   # noqa: E402

lib, kind = sys.argv[1], sys.argv[2]
sm = SynthModel(SMALL_SBG_SPEC if kind in ("sbg", "sbgtypo") else SMALL_CONG_CHR_SPEC if kind == "chr" else SMALL_SPEC)
path = os.path.join(ROOT, "_data", "small-sbg.raw" if kind in ("sbg", "sbgtypo") else "small-cong-chr.raw" if kind == "chr" else "small.raw")
if not os.path.exists(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    sm.raw.save(path)
if kind == "congg":
    # the global CoNgram model (viterbi_kernel_congg.hip, viterbi_kernel_congg_typo.hip): sentences long enough for containers past 64 entries (the replay),
    # top-1 and top-2, and the typo-correcting analysis
    import random
    from kiwi_amd.api import Typo
    from kiwi_amd.synth import SMALL_CONG_GLOBAL_SPEC
    from typo_cases import misspell
    sm = SynthModel(SMALL_CONG_GLOBAL_SPEC)
    path = os.path.join(ROOT, "_data", "small-cong-global32.raw")
    sm.raw.save(path)
    dev = KiwiAmd(path, lib_path=lib, lm_mode=4)
    texts = synthetic(sm, 8, 941, min_jamo=60, max_jamo=140) + dictionary_mix(sm, 3, 942)
    for top_n in (1, 2):
        assert len(dev.analyze_batch(texts, top_n=top_n).to_python()) == len(texts)
    ty = Typo.from_default(dev.lib, 3).prepare(True)
    rnd = random.Random(7)
    tt = [misspell(t, rnd, True, True) for t in synthetic(sm, 4, 943, min_jamo=10, max_jamo=40)]
    assert len(dev.analyze_batch_opt(tt, typo=ty, typo_threshold=2.5).to_python()) == len(tt)
    dev.close()
    print("sanitizer run complete:", kind, len(texts), "texts")
    sys.exit(0)
dev = KiwiAmd(path, lib_path=lib)
if kind == "chr":
    # Match::oovChrModel on a CoNgram model: k_unk_chr, the CoNgram search reading its scores -- plain and with a typo transformer (k_typo_graph,
    # typo lattices, the typo + CoNgram search)
    import random
    import test_hipemu
    from typo_cases import misspell
    texts = synthetic(sm, 10, 621, min_jamo=5, max_jamo=60) + dictionary_mix(sm, 5, 622) + EDGE_TEXTS[:25]
    import oraclelib
    match = oraclelib.MATCH_ALL_WITH_NORMALIZING | (1 << 8)
    assert len(dev.analyze_batch(texts, match=match).to_python()) == len(texts)
    # ... and Match::oovChrFreqModel: k_unk_chr_freq (substring counts over the filtered text, a column of LDS counters per lane); texts of several chunks
    from corpora import repeated_unknown_texts
    ft = repeated_unknown_texts(sm, 12, 623) + [". ".join(texts[:8]) + ".", "가" * 40 + " " + "가" * 40] + EDGE_TEXTS[:25]
    assert len(dev.analyze_batch(ft, match=oraclelib.MATCH_ALL_WITH_NORMALIZING | (2 << 8)).to_python()) == len(ft)
    prod, _ = test_hipemu._typo_pair(lib, 1.0)
    rnd = random.Random(5)
    tt = [misspell(t, rnd, True, True) for t in texts[:12]]
    assert len(test_hipemu._analyze_typo(dev, prod, tt, 2.5, match=match)) == len(tt)
    dev.close(); prod.close()
    print("sanitizer run complete:", kind, len(texts), "texts")
    sys.exit(0)
if kind in ("typo", "sbgtypo"):
    # (sbgtypo: the same on a SkipBigram model -- viterbi_kernel_sbg_typo.hip)
    # the typo-correcting analysis (typo lattice kernel, search with node typo costs) on misspelt texts
    import random
    import test_hipemu
    from typo_cases import misspell
    import test_typo_product
    from typo_cases import COND, INF, RULES
    test_typo_product.LIB = lib
    prod = test_typo_product.ProductTypo(1.0, INF)
    for origs, errs, cost, cond, dia in RULES:
        for o in origs:
            for e in errs:
                assert prod.add(o, e, cost, COND[cond], dia) == 0
    prod.prepare(True)
    rnd = random.Random(3)
    texts = [misspell(t, rnd, True, True) for t in synthetic(sm, 12 if kind == "typo" else 6, 611, min_jamo=5, max_jamo=60 if kind == "typo" else 40) + dictionary_mix(sm, 6 if kind == "typo" else 3, 612)]
    for top_n in (1, 2) if kind == "typo" else (1,):
        assert len(test_hipemu._analyze_typo(dev, prod, texts, 2.5, top_n)) == len(texts)
    dev.close(); prod.close()
    print("sanitizer run complete:", kind, len(texts), "texts")
    sys.exit(0)
n = 8 if kind == "sbg" else 9
texts = synthetic(sm, n, 601, min_jamo=5, max_jamo=50 if kind == "sbg" else 60) + dictionary_mix(sm, n // 2, 602) + (EDGE_TEXTS if kind != "sbg" else [])
for top_n in (1, 2):
    res = dev.analyze_batch(texts, top_n=top_n).to_python()
    assert len(res) == len(texts)
dev.close()
print("sanitizer run complete:", kind, len(texts), "texts")
