"""Generates tests/golden/small_model_golden.json: outputs of the REAL reference translation units
(oracle/_ref/libkiwi_ref.so, built from /root/reference by oracle/Makefile) on the deterministic small
synthetic model.  Run in the container that has /root/reference; the JSON is committed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kiwi_amd.synth import SynthModel, SMALL_SPEC  # noqa: E402
import refbridge  # noqa: E402
from corpora import EDGE_TEXTS, dictionary_mix, synthetic  # noqa: E402

sm = SynthModel(SMALL_SPEC)
os.makedirs(os.path.join(ROOT, "_data"), exist_ok=True)
path = os.path.join(ROOT, "_data", "small.raw")
sm.raw.save(path)
r = refbridge.RefKiwi(path)
texts = synthetic(sm, 150, 101, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 100, 102) + EDGE_TEXTS
items = []
for s in texts:
    res = r.analyze(s)
    items.append({"text": s, "score": res[0][1],
                  "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.sense_id, t.morph_id] for t in res[0][0]]})
out = os.path.join(ROOT, "tests", "golden", "small_model_golden.json")
json.dump({"model": "kiwi_amd.synth.SMALL_SPEC", "reference": "bab2min/Kiwi v0.23.1 TUs via oracle/ref_bridge.cpp", "items": items},
          open(out, "w", encoding="utf-8"), ensure_ascii=True)
print(len(items), "items ->", out)

# ---- top-N and SkipBigram goldens.  The reference hands paths on in the order of thread_local containers whose state depends on
# what the thread analysed before, so these vectors are a SEQUENCE: generated here in one go from a fresh process state (the
# top-1 Knlm pass above does not touch the top-N / SkipBigram containers), to be replayed in the same order by a fresh oracle
# handle in its reference-faithful mode (tests/test_oracle_vs_ref.py::test_golden_sequences).
from kiwi_amd.synth import SMALL_SBG_SPEC  # noqa: E402


def dump(res):
    return [{"score": a[1], "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.sense_id, t.morph_id, t.score] for t in a[0]]} for a in res]


seq_texts = synthetic(sm, 120, 111, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 80, 112)
top3 = [{"text": s, "analyses": dump(r.analyze(s, top_n=3))} for s in seq_texts]
out = os.path.join(ROOT, "tests", "golden", "small_model_top3_sequence.json")
json.dump({"model": "kiwi_amd.synth.SMALL_SPEC", "top_n": 3, "reference": "bab2min/Kiwi v0.23.1 TUs via oracle/ref_bridge.cpp", "items": top3},
          open(out, "w", encoding="utf-8"), ensure_ascii=True)
print(len(top3), "items ->", out)

sm_sbg = SynthModel(SMALL_SBG_SPEC)
path_sbg = os.path.join(ROOT, "_data", "small-sbg.raw")
sm_sbg.raw.save(path_sbg)
r_sbg = refbridge.RefKiwi(path_sbg)
sbg = [{"text": s, "analyses": dump(r_sbg.analyze(s))} for s in seq_texts]
out = os.path.join(ROOT, "tests", "golden", "small_sbg_model_sequence.json")
json.dump({"model": "kiwi_amd.synth.SMALL_SBG_SPEC", "top_n": 1, "reference": "bab2min/Kiwi v0.23.1 TUs via oracle/ref_bridge.cpp (SkipBigramModel)", "items": sbg},
          open(out, "w", encoding="utf-8"), ensure_ascii=True)
print(len(sbg), "items ->", out)
