"""Generates tests/golden/pretokenized_small.json: analyses of the REAL reference with pretokenized spans (Kiwi::analyze(..., pretokenized), src/Kiwi.cpp:785-946,
1043-1051, KTrie.cpp:1177-1210; through oracle/ref_bridge.cpp kref_analyze_pretokenized) on the small synthetic model -- the argument of kiwi_analyze* this repo's
product still refuses.  The cases cover what makePretokenizedSpanGroup distinguishes: a span without tokens (a dictionary form reused, or the fallback NNP form with
the text as its own string), one token that IS a single-candidate dictionary entry, one token with a tag the dictionary does not have for the form (temporary form +
morpheme), several tokens (one temporary morpheme with chunks), spans next to each other, at the ends of the text, over spaces and special characters, and top-3.
Run in the container that has /root/reference; the JSON is committed (tests/test_pretokenized_golden.py replays it against the reference where that travelled, and
is what the restatement / device path of the next round are checked against)."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refbridge  # noqa: E402
from corpora import dictionary_mix, synthetic  # noqa: E402
from kiwi_amd.synth import SMALL_SPEC, SynthModel  # noqa: E402

NNG, NNP, VV, JKS, SL = 1, 2, 4, 39, 30      # tag ids (kiwi_amd/csrc/kchars.hpp)


def make_cases(sm, n=160, seed=1301):
    rnd = random.Random(seed)
    texts = synthetic(sm, n, seed, min_jamo=10, max_jamo=70) + dictionary_mix(sm, n // 4, seed + 1)
    cases = []
    for i, t in enumerate(texts):
        words, pos, at = [], [], 0
        for w in t.split(" "):
            if w:
                words.append(w); pos.append(at)
            at += len(w) + 1
        if len(words) < 3:
            continue
        kind = i % 8
        k = rnd.randrange(len(words))
        b, e = pos[k], pos[k] + len(words[k])
        w = words[k]
        if kind == 0:
            spans = [(b, e, [])]
        elif kind == 1:
            spans = [(b, e, [(w, 0, e - b, NNP, 1)])]
        elif kind == 2:
            spans = [(b, e, [(w, 0, e - b, rnd.choice([NNG, VV, SL]), rnd.randrange(2))])]
        elif kind == 3 and len(w) >= 2:
            h = rnd.randrange(1, len(w))
            spans = [(b, e, [(w[:h], 0, h, NNG, 1), (w[h:], h, e - b, JKS, 1)])]
        elif kind == 4 and k + 1 < len(words):
            e2 = pos[k + 1] + len(words[k + 1])
            spans = [(b, e2, [])]                                                  # over a space
        elif kind == 5 and k + 1 < len(words):
            spans = [(b, e, []), (pos[k + 1], pos[k + 1] + len(words[k + 1]), [(words[k + 1], 0, len(words[k + 1]), NNG, 1)])]      # neighbours
        elif kind == 6:
            spans = [(0, len(words[0]), []), (pos[-1], pos[-1] + len(words[-1]), [])]      # both ends (the last one takes the final punctuation along)
        else:
            spans = [(b, e, [(w, 0, e - b, NNP, 1)])]
        cases.append({"text": t, "spans": [[sb, se, [list(tk) for tk in toks]] for sb, se, toks in spans], "top_n": 3 if i % 5 == 0 else 1})
    return cases


def run(ref, case):
    spans = [(sb, se, [tuple(tk) for tk in toks]) for sb, se, toks in case["spans"]]
    res = ref.analyze_pretokenized(case["text"], spans, top_n=case["top_n"])
    return [{"score": r[1], "tokens": [[x.form, x.tag, x.position, x.length, x.word_position, x.sent_position, x.score, x.typo_form_id, x.morph_id >= 0] for x in r[0]]} for r in res]


if __name__ == "__main__":
    sm = SynthModel(SMALL_SPEC)
    os.makedirs(os.path.join(ROOT, "_data"), exist_ok=True)
    path = os.path.join(ROOT, "_data", "small.raw")
    sm.raw.save(path)
    ref = refbridge.RefKiwi(path)
    cases = make_cases(sm)
    for c in cases:
        c["results"] = run(ref, c)
    out = os.path.join(ROOT, "tests", "golden", "pretokenized_small.json")
    json.dump({"model": "kiwi_amd.synth.SMALL_SPEC", "reference": "bab2min/Kiwi v0.23.1 TUs via oracle/ref_bridge.cpp (kref_analyze_pretokenized)",
               "token_fields": ["form", "tag", "position", "length", "word_position", "sent_position", "score", "typo_form_id (span index + 1)", "morpheme of the model (false: a temporary one)"],
               "cases": cases}, open(out, "w", encoding="utf-8"), ensure_ascii=True)
    inside = sum(any(tok[7] for tok in r["tokens"]) for c in cases for r in c["results"][:1])
    temp = sum(any(not tok[8] for tok in r["tokens"]) for c in cases for r in c["results"][:1])
    print(len(cases), "cases ->", out, os.path.getsize(out), "bytes;", inside, "with a token inside a span,", temp, "with a temporary morpheme")
