#!/usr/bin/env python3
"""Writes the eval_data parity corpus: tests/golden/eval_data_lexicon.json -- the (form, tag) pairs of the gold annotations of
/root/reference/eval_data/{web,written,web_with_typos,web_with_cont_typos}.txt, which kiwi_amd.workloads adds to the small synthetic model as
dictionary entries ('small-eval') so that real text meets a real lattice -- and tests/golden/eval_data_<file>.json: column 1 of every line with
what the REAL reference (oracle/_ref) answers on that model: tokens, positions, fp32 scores (top-1; the two typo files with the built-in typo
set `basicTypoSetWithContinual`, threshold 2.5, typo costs included).  Accuracy against the gold column is NOT what this pins: the language
model is synthetic (the shipped model files are git-LFS pointers in this environment).  Run in the build container; the JSON travels."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
EVAL = "/root/reference/eval_data"
FILES = ("web", "written", "web_with_typos", "web_with_cont_typos")
GOLD = os.path.join(ROOT, "tests", "golden")


def lexicon():
    seen, out = set(), []
    for fn in FILES:
        for line in open(os.path.join(EVAL, fn + ".txt"), encoding="utf-8"):
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2:
                continue
            for tok in parts[1].split(" "):
                if "/" not in tok:
                    continue
                form, tag = tok.rsplit("/", 1)
                if form and (form, tag) not in seen:
                    seen.add((form, tag)); out.append([form, tag])
    return out


def main():
    lex = lexicon()
    json.dump({"source": "gold annotations of /root/reference/eval_data/*.txt (tools/make_golden_eval.py)", "entries": lex},
              open(os.path.join(GOLD, "eval_data_lexicon.json"), "w", encoding="utf-8"), ensure_ascii=False, separators=(",", ":"))
    from kiwi_amd.workloads import eval_model
    import refbridge
    path, added = eval_model()
    print(len(lex), "gold (form, tag) pairs,", added, "dictionary entries added ->", path)
    ref = refbridge.RefKiwi(path)
    typo = refbridge.RefTypo.from_default("basic_with_continual")
    typo.prepare(True)
    for fn in FILES:
        items = []
        with_typo = "typos" in fn
        for line in open(os.path.join(EVAL, fn + ".txt"), encoding="utf-8"):
            text = line.rstrip("\n").split("\t")[0]
            if not text:
                continue
            res = ref.analyze_typo(typo, text, 2.5, 0) if with_typo else ref.analyze(text)
            toks, score = res[0]
            items.append({"text": text, "score": score,
                          "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.line_number, t.score, t.typo_cost] for t in toks]})
        json.dump({"source": f"the real reference (oracle/_ref) on eval_data/{fn}.txt column 1, model 'small-eval'" + (", typo set basicTypoSetWithContinual, threshold 2.5" if with_typo else ""),
                   "typo": with_typo, "items": items}, open(os.path.join(GOLD, f"eval_data_{fn}.json"), "w", encoding="utf-8"), ensure_ascii=False, separators=(",", ":"))
        print(fn, len(items), "lines,", sum(len(i["tokens"]) for i in items), "tokens")


if __name__ == "__main__":
    main()
