#!/usr/bin/env python3
"""Writes tests/golden/cong_global_{32,16}.json: what the REAL reference (oracle/_ref/libkiwi_ref_x86.so, SSE4.1 dispatch) answers with the CoNgram blob
of kiwi_amd.synth.SMALL_CONG_GLOBAL_SPEC / SMALL_CONG_GLOBAL16_SPEC loaded as the GLOBAL model (CoNgramModelBase::create(useDistantTokens = true):
ModelType::congGlobal, window 7) -- top-1 on synthetic sentences long enough to fill its path containers beyond 64 entries, top-3, open ending, and
misspelt sentences with the built-in typo set basicTypoSetWithContinual.  Replayed by tests/test_cong_global.py without the reference.  Run in the
build container."""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
# the sentence on which the reference's container defects (BestPathContainer.hpp:316-351, search.cpp:555-597) first showed: with equal states merged
# as the code intends, the analysis comes out differently
DEFECT_TEXT = "햼붸쮀 햐겡 듸어쾌 킌너 소갸탸모슈 륲야아아 셔걠섀픠태"


def rows(res):
    return [[[[t.form, t.tag, t.position, t.length, t.score, t.typo_cost] for t in toks], score] for toks, score in res]


def main():
    import refbridge
    from kiwi_amd import synth
    from corpora import synthetic, dictionary_mix
    from typo_cases import misspell
    for which, spec in (("32", synth.SMALL_CONG_GLOBAL_SPEC), ("16", synth.SMALL_CONG_GLOBAL16_SPEC)):
        path = os.path.join(ROOT, "_data", f"small-cong-global{which}.raw")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        sm = synth.SynthModel(spec)
        sm.raw.save(path)
        ref = refbridge.RefKiwi(path, arch=3, model_dir_sbg="cong_global", x86=True)
        texts = [DEFECT_TEXT] + synthetic(sm, 120, 931, min_jamo=20, max_jamo=140) + dictionary_mix(sm, 40, 932)
        out = {"source": "real reference, SSE4.1 dispatch, ModelType::congGlobal (tools/make_golden_cong_global.py)",
               "top1": [{"text": t, "res": rows(ref.analyze(t))} for t in texts],
               "top3": [{"text": t, "res": rows(ref.analyze(t, top_n=3))} for t in texts[1:41]],
               "open_ending": [{"text": t, "res": rows(ref.analyze(t, open_ending=True))} for t in texts[41:71]]}
        rt = refbridge.RefTypo(); rt.update_default("basic_with_continual"); rt.prepare(True)
        rnd = random.Random(7)
        tt = [misspell(t, rnd, True, True, False) for t in texts[1:41]]
        out["typo"] = [{"text": t, "res": rows(ref.analyze_typo(rt, t, 2.5, 0))} for t in tt]
        fn = os.path.join(ROOT, "tests", "golden", f"cong_global_{which}.json")
        json.dump(out, open(fn, "w", encoding="utf-8"), ensure_ascii=False, separators=(",", ":"))
        print(fn, os.path.getsize(fn), "bytes;", sum(len(x["res"][0][0]) for x in out["top1"]), "top-1 tokens")


if __name__ == "__main__":
    main()
