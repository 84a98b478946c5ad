"""Run by tests/test_hipemu.py in a subprocess with a sanitizer runtime preloaded: a small corpus through the sanitizer build of the
emulated library (Knlm with KAMD_TEST_TINY_ARENAS, i.e. with most chunks hitting a capacity limit first; then the SkipBigram kernel).
Any report of the sanitizer ends the process with a non-zero status."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from corpora import EDGE_TEXTS, dictionary_mix, synthetic   # noqa: E402
from kiwi_amd.api import KiwiAmd                              # noqa: E402
from kiwi_amd.synth import SMALL_SBG_SPEC, SMALL_SPEC, SynthModel   # noqa: E402

lib, kind = sys.argv[1], sys.argv[2]
sm = SynthModel(SMALL_SBG_SPEC if kind == "sbg" else SMALL_SPEC)
path = os.path.join(ROOT, "_data", "small-sbg.raw" if kind == "sbg" else "small.raw")
if not os.path.exists(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    sm.raw.save(path)
dev = KiwiAmd(path, lib_path=lib)
n = 8 if kind == "sbg" else 14
texts = synthetic(sm, n, 601, min_jamo=5, max_jamo=50 if kind == "sbg" else 80) + dictionary_mix(sm, n // 2, 602) + (EDGE_TEXTS if kind != "sbg" else [])
for top_n in (1, 2):
    res = dev.analyze_batch(texts, top_n=top_n).to_python()
    assert len(res) == len(texts)
dev.close()
print("sanitizer run complete:", kind, len(texts), "texts")
