#!/usr/bin/env python3
"""Writes the dialect fixtures: tests/golden/eval_dialect_lexicon.json -- the (form, tag) pairs of the gold annotations of
/root/reference/eval_data/dialect/*.txt that the standard eval_data files do not have, each with the Dialect bits of the files it occurs in, which
kiwi_amd.workloads.dialect_model() adds to the small synthetic model as DIALECT morphemes -- and tests/golden/eval_dialect.json: a sample of the files'
sentences with what the REAL reference (oracle/_ref; the bake with KiwiBuilder's enabledDialects) answers for

    enabled = all:  allowedDialects 0 (standard only), the file's own dialect, all dialects (dialectCost 3, and once 1.5);
    enabled = none: allowedDialects all (the dialect forms are not in the trie; the reference still takes its `dialect` typo set)

without a typo transformer (the reference then takes its built-in set DefaultTypoSet::dialect, src/Kiwi.cpp:1037-1041) and, for a few, with
basicTypoSetWithContinual.  The language model is synthetic: what is pinned is the mechanism, not dialect accuracy.  Run in the build container."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
EVAL = "/root/reference/eval_data"
GOLD = os.path.join(ROOT, "tests", "golden")
ALL = 1023      # Dialect::all (include/kiwi/Types.h:334)
PER_FILE = 40


def main():
    from kiwi_amd.workloads import DIALECT_BITS, dialect_model
    std = {tuple(e) for e in json.load(open(os.path.join(GOLD, "eval_data_lexicon.json"), encoding="utf-8"))["entries"]}
    seen, order = {}, []
    for name, bit in DIALECT_BITS.items():
        for line in open(os.path.join(EVAL, "dialect", name + ".txt"), encoding="utf-8"):
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2:
                continue
            for tok in parts[1].split(" "):
                if "/" not in tok:
                    continue
                form, tag = tok.rsplit("/", 1)
                form = form.split("__")[0]
                if not form or (form, tag) in std:
                    continue
                if (form, tag) not in seen:
                    seen[(form, tag)] = 0; order.append((form, tag))
                seen[(form, tag)] |= bit
    # a pair the gold of five or more dialects shares is common vocabulary the standard files happen not to hold: standard
    entries = [[f, t, seen[(f, t)]] for f, t in order if bin(seen[(f, t)]).count("1") < 5]
    json.dump({"source": "gold annotations of /root/reference/eval_data/dialect/*.txt not in eval_data_lexicon.json, Dialect bits of the files they occur in (tools/make_golden_dialect.py)",
               "entries": entries}, open(os.path.join(GOLD, "eval_dialect_lexicon.json"), "w", encoding="utf-8"), ensure_ascii=False, separators=(",", ":"))
    import refbridge
    path = dialect_model()
    print(len(entries), "dialect (form, tag) pairs ->", path)
    ref_all = refbridge.RefKiwi(path, model_dir_sbg=("dialects", ALL))
    ref_std = refbridge.RefKiwi(path, model_dir_sbg=("dialects", 0))
    typo = refbridge.RefTypo.from_default("basic_with_continual"); typo.prepare(True)
    items = []
    def rec(res):
        toks, score = res[0]
        return {"score": score, "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.line_number, t.score, t.typo_cost, t.dialect] for t in toks]}
    for name, bit in DIALECT_BITS.items():
        lines = [ln.rstrip("\n").split("\t")[0] for ln in open(os.path.join(EVAL, "dialect", name + ".txt"), encoding="utf-8")]
        lines = [t for t in lines if t][:PER_FILE]
        for k, text in enumerate(lines):
            it = {"text": text, "file": name, "bit": bit,
                  "enabled_all": {"allowed_0": rec(ref_all.analyze_dialect(text, 0)), "allowed_own": rec(ref_all.analyze_dialect(text, bit)), "allowed_all": rec(ref_all.analyze_dialect(text, ALL)),
                                  "allowed_all_cost_1.5": rec(ref_all.analyze_dialect(text, ALL, 1.5))},
                  "enabled_none": {"allowed_all": rec(ref_std.analyze_dialect(text, ALL))}}
            if k < 6:
                it["enabled_all"]["allowed_own_typo_basic_with_continual"] = rec(ref_all.analyze_dialect(text, bit, 3.0, typo=typo))
            items.append(it)
    json.dump({"source": "the real reference (oracle/_ref) on eval_data/dialect/*.txt column 1, model kiwi_amd.workloads.dialect_model() (tools/make_golden_dialect.py)", "items": items},
              open(os.path.join(GOLD, "eval_dialect.json"), "w", encoding="utf-8"), ensure_ascii=False, separators=(",", ":"))
    nd = sum(1 for it in items for t in it["enabled_all"]["allowed_all"]["tokens"] if t[9])
    diff = sum(1 for it in items if it["enabled_all"]["allowed_all"]["tokens"] != it["enabled_all"]["allowed_0"]["tokens"])
    print(len(items), "sentences;", nd, "dialect tokens under allowed = all;", diff, "sentences analysed differently with / without dialects allowed")


if __name__ == "__main__":
    main()
