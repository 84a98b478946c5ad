"""Run by tests/test_oracle_vs_ref.py in a FRESH process (the reference keeps its path containers in thread_local storage that
is never shrunk, so only a fresh process gives both sides the same -- empty -- history): the oracle in its reference-faithful
mode against the real reference, same texts in the same sequence.  usage: faithful_check.py small|small-sbg <top_n>"""
import os
import sys
from dataclasses import astuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]
import oraclelib  # noqa: E402
import refbridge  # noqa: E402
from corpora import EDGE_TEXTS, dictionary_mix, synthetic  # noqa: E402
from kiwi_amd.synth import SMALL_SBG_SPEC, SMALL_SPEC, SynthModel  # noqa: E402

kind, top_n = sys.argv[1], int(sys.argv[2])
sm = SynthModel(SMALL_SBG_SPEC if kind == "small-sbg" else SMALL_SPEC)
path = os.path.join(ROOT, "_data", kind + ".raw")
os.makedirs(os.path.dirname(path), exist_ok=True)
sm.raw.save(path)
orc, ref = oraclelib.OracleKiwi(path), refbridge.RefKiwi(path)
orc.set_faithful_order(True)
texts = synthetic(sm, 300, 201, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 200, 202) + [t for t in EDGE_TEXTS if t.strip()]
bad = 0
for s in texts:
    x = [([astuple(t) for t in a[0]], a[1]) for a in orc.analyze(s, top_n=top_n)]
    y = [([astuple(t) for t in a[0]], a[1]) for a in ref.analyze(s, top_n=top_n)]
    if x != y:
        bad += 1
        print("MISMATCH", repr(s)[:100])
print("checked", len(texts), "mismatches", bad)
sys.exit(1 if bad else 0)
