"""Generates tests/golden/model_variants_golden.json: outputs of the REAL reference translation units on the synthetic models of the round-2 file
variants -- a history-transformed, 8-bit quantised, compressed sj.knlm (oracle/_ref/libkiwi_ref.so), and a CoNgram model with the character model
of Match::oovChrModel, analysed without and with that option (oracle/_ref/libkiwi_ref_x86.so, SSE4.1 build: the pin of the CoNgram oracle).
Run in the container that has /root/reference; the JSON is committed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kiwi_amd.synth import SMALL_CONG_CHR_SPEC, SMALL_HTX_Q8_SPEC, SynthModel  # noqa: E402
import refbridge  # noqa: E402
from corpora import EDGE_TEXTS, dictionary_mix, synthetic  # noqa: E402

OOV_CHR_MODEL = 1 << 8
os.makedirs(os.path.join(ROOT, "_data"), exist_ok=True)
sets = {}
for name, spec, fname, x86, match in (("htx-q8c", SMALL_HTX_Q8_SPEC, "small-htx-q8.raw", False, refbridge.MATCH_ALL_WITH_NORMALIZING),
                                      ("cong", SMALL_CONG_CHR_SPEC, "small-cong-chr.raw", True, refbridge.MATCH_ALL_WITH_NORMALIZING),
                                      ("cong-oov-chr", SMALL_CONG_CHR_SPEC, "small-cong-chr.raw", True, refbridge.MATCH_ALL_WITH_NORMALIZING | OOV_CHR_MODEL)):
    sm = SynthModel(spec)
    path = os.path.join(ROOT, "_data", fname)
    sm.raw.save(path)
    r = refbridge.RefKiwi(path, arch=3, x86=True) if x86 else refbridge.RefKiwi(path)
    texts = synthetic(sm, 90, 111, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 60, 112) + EDGE_TEXTS
    items = []
    for s in texts:
        res = r.analyze(s, match=match)
        items.append({"text": s, "score": res[0][1],
                      "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.sense_id, t.morph_id] for t in res[0][0]]})
    sets[name] = {"match": match, "items": items}
out = os.path.join(ROOT, "tests", "golden", "model_variants_golden.json")
json.dump({"models": {"htx-q8c": "kiwi_amd.synth.SMALL_HTX_Q8_SPEC", "cong": "kiwi_amd.synth.SMALL_CONG_CHR_SPEC", "cong-oov-chr": "kiwi_amd.synth.SMALL_CONG_CHR_SPEC + Match::oovChrModel"},
           "reference": "bab2min/Kiwi v0.23.1 TUs via oracle/ref_bridge.cpp (CoNgram: SSE4.1 build)", "sets": sets}, open(out, "w", encoding="utf-8"), ensure_ascii=True)
print({k: len(v["items"]) for k, v in sets.items()}, "->", out)
