"""Developer check: small corpus through the HIP path (library from $KAMD_LIB) vs the CPU oracle; prints OK/DIFF."""
import os, sys
from dataclasses import astuple
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kiwi_amd.synth import SynthModel, SMALL_SPEC, SMALL_SBG_SPEC
from kiwi_amd.api import KiwiAmd
import oraclelib
os.makedirs(os.path.join(ROOT, "_data"), exist_ok=True)
sbg = bool(os.environ.get("KAMD_EXPERIMENTAL_SBG"))      # the SkipBigram model and its (gated) search kernel
path = os.path.join(ROOT, "_data", "small-sbg.raw" if sbg else "small.raw")
sm = SynthModel(SMALL_SBG_SPEC if sbg else SMALL_SPEC); sm.raw.save(path)
o = oraclelib.OracleKiwi(path); k = KiwiAmd(path)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
corpus = sm.make_corpus(n, 77, min_jamo=5, max_jamo=120)
print("start", os.environ.get("KAMD_LIB"), flush=True)
res = k.analyze_batch(corpus).to_python()
bad = 0
for s, y in zip(corpus, res):
    x = o.analyze(s)
    if [([astuple(t) for t in a[0]], a[1]) for a in x] != [([astuple(t) for t in a[0]], a[1]) for a in y]: bad += 1
print("RESULT bad", bad, "/", n, flush=True)
