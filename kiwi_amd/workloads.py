"""Benchmark / test workloads on the synthetic model.

The real Kiwi model binaries are not available, so every workload uses the deterministic synthetic model of
``kiwi_amd/synth.py``.  Models and corpora are cached under ``_data/`` (git-ignored, shipped to the GPU box).
"""
from __future__ import annotations

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "_data")

SEED_C5 = 0x4B495749 + 5
WORKLOADS = {
    # name: (spec name, n sentences, corpus kwargs, seed offset)  -- seeds follow BASELINE.md ("KIWI" + config index)
    "c2": ("full", 8192, dict(exact_jamo=40), 2),
    "c3": ("full", 65536, dict(min_jamo=5, max_jamo=200), 3),
    # BASELINE config 3 proper: SkipBigram model, top-3
    "c3-sbg": ("full-sbg", 65536, dict(min_jamo=5, max_jamo=200), 3),
    "small-c2": ("small", 8192, dict(exact_jamo=40), 2),
    # BASELINE config 5: the c2 corpus misspelt (confusable vowels, carried-over codas), analysed with a typo transformer (TYPO_RULES, continual cost 1,
    # typoCostWeight 6, threshold 2.5)
    "c5": ("full", 8192, dict(exact_jamo=40), 2),
    # BASELINE config 4's model type and length distribution on one GPU: CoNgram model (local, 8-bit, dim 64), sentence lengths log-normal
    # (median 60 jamo, sigma 0.6, clipped to 5..400); 131072 sentences = one GPU's share of the 1M-sentence corpus on 8 GPUs
    "c4-cong": ("full-cong", 131072, dict(min_jamo=5, max_jamo=400, lognormal=(60, 0.6)), 4),
    # ... and the reference's largest model type on the same lexicon and length distribution: CoNgram global (window 7: valid distant tokens are scored as a mixture over
    # the context and the last seven such tokens of the path; kiwi_init's LARGEST / CONG_GLOBAL); the c4 corpus itself (rounds 5: its first 32768 sentences)
    "c4-cong-global": ("full-cong-global", 131072, dict(min_jamo=5, max_jamo=400, lognormal=(60, 0.6)), 4),
    "small-cong-c2": ("small-cong", 8192, dict(exact_jamo=40), 2),
    # c2 at the batch size the north-star throughput target is quoted on (>= 64k sentences): the throughput regime
    "c2-64k": ("full", 65536, dict(exact_jamo=40), 12),
}


INF = float("inf")
COND = {"none": 0, "any": 1, "vowel": 2, "vocalic": 3, "vocalic_h": 4, "non_vowel": 5, "non_vocalic": 6, "non_vocalic_h": 7, "applosive": 8, "continual": 9, "boundary": 10}
# This repo's own typo rules (the reference's built-in sets are its data and are not shipped): (origs, errors, cost, left condition, dialect bits)
TYPO_RULES = [
    (["ㅐ", "ㅔ"], ["ㅐ", "ㅔ"], 1.0, "none", 0), (["ㅚ", "ㅙ"], ["ㅞ", "ㅐ"], 1.5, "none", 0), (["ㅟ", "ㅢ"], ["ㅣ"], 1.0, "none", 0),
    (["위", "의"], ["이"], INF, "none", 0), (["위", "의"], ["이"], 1.0, "any", 0), (["자", "쟈"], ["자", "쟈"], 1.0, "none", 0),
    (["ᆻ어"], ["ᆺ어", "ᆺ서"], 1.0, "none", 0), (["ᆫᄒ"], ["ᆫᄒ", "ᆭᄋ"], 2.0, "none", 0), (["ᄒ"], ["ᄋ"], 0.5, "vowel", 0),
    (["ᄒ", "ᄀ"], ["ᄏ", "ᄁ"], 1.0, "applosive", 0), (["ᆨᄋ"], ["ᄀ"], 1.0, "continual", 0), (["ᆫᄋ"], ["ᄂ"], 1.0, "continual", 0),
    (["ᆯᄋ"], ["ᄅ"], 1.0, "continual", 0), (["시어"], ["셔"], 0.25, "boundary", 8), (["지어"], ["져"], 0.25, "boundary", 0),
    (["안"], ["않"], 1.5, "none", 0), (["돼"], ["되"], 1.0, "none", 0), (["던"], ["든"], 1.0, "none", 16),
]
CODA2ONSET = {1: 0, 4: 2, 7: 3, 8: 5, 16: 6, 17: 7, 19: 9, 22: 12, 23: 14, 24: 15, 25: 16, 26: 17, 27: 18}
LENGTHENING_VOWEL = [0, 1, 0, 1, 4, 5, 4, 5, 8, 0, 1, 1, 8, 13, 4, 5, 20, 13, 18, 20, 20]


def misspell(text, rnd, vowels=True, carry=True, lengthen=False):
    """Injects the kinds of errors typo transformers correct into a text of the synthetic model: confusable vowels (ㅐ/ㅔ, ㅚ/ㅙ), a coda
    written as the onset of the following vowel-initial syllable (연철, what continual rules undo), and 1-3 syllables that merely lengthen
    the vowel of an open syllable ("가아아", what the lengthening cost pays for)."""
    o = list(text)
    if vowels:
        for i, ch in enumerate(o):
            c = ord(ch)
            if 0xAC00 <= c < 0xD7A4 and rnd.random() < 0.15:
                v = (c - 0xAC00) // 28 % 21
                if v == 1: c += 4 * 28
                elif v == 5: c -= 4 * 28
                elif v == 11: c -= 1 * 28
                o[i] = chr(c)
    if carry:
        for i in range(len(o) - 1):
            a, b = ord(o[i]), ord(o[i + 1])
            if 0xAC00 <= a < 0xD7A4 and 0xAC00 <= b < 0xD7A4 and rnd.random() < 0.5:
                coda = (a - 0xAC00) % 28
                onset = (b - 0xAC00) // 28 // 21
                if coda in CODA2ONSET and onset == 11:
                    o[i] = chr(a - coda)
                    o[i + 1] = chr(b + (CODA2ONSET[coda] - 11) * 21 * 28)
    if lengthen:
        p = []
        for ch in o:
            p.append(ch)
            c = ord(ch)
            if 0xAC00 <= c < 0xD7A4 and (c - 0xAC00) % 28 == 0 and rnd.random() < 0.2:
                p.append(chr(0xAC00 + (11 * 21 + LENGTHENING_VOWEL[(c - 0xAC00) // 28 % 21]) * 28) * rnd.randint(1, 3))
        o = p
    return "".join(o)


def fill_typo_rules(transformer, cond_by_name=False):
    """TYPO_RULES through `transformer.add(orig, error, cost, cond, dialect)`."""
    for origs, errs, cost, cond, dia in TYPO_RULES:
        for o in origs:
            for e in errs:
                transformer.add(o, e, cost, cond if cond_by_name else COND[cond], dia)


def workload_typo(name: str):
    """None, or (continual cost, lengthening cost, threshold) of the transformer the workload is analysed with (rules: TYPO_RULES)."""
    return (1.0, INF, 2.5) if name == "c5" else None


def _spec(name):
    from dataclasses import replace
    from .synth import FULL_CONG_SPEC, FULL_SBG_SPEC, FULL_SPEC, SMALL_CONG_SPEC, SMALL_SPEC
    return {"full": FULL_SPEC, "full-sbg": FULL_SBG_SPEC, "small": SMALL_SPEC, "full-cong": FULL_CONG_SPEC, "small-cong": SMALL_CONG_SPEC,
            "full-cong-global": replace(FULL_CONG_SPEC, cong_window=7)}[name]


# morphemes that the reference's shipped default.dict / typo.dict refer to as "original" morphemes (pre-analysed entries, allomorph definitions) beyond
# those of the eval_data gold lexicon: the real KiwiBuilder refuses the files without them (tools/make_golden_built.py)
EVAL_BUILDER_REQUIRED = (("으라", "EC"), ("ᆫ다", "EC"), ("편찮", "VA"), ("하찮", "VA"), ("시끄럽", "VA-I"))


def eval_model(for_builder=False):
    """The small synthetic model with the gold (form, tag) pairs of the reference's eval_data files as additional dictionary entries
    (tests/golden/eval_data_lexicon.json, written by tools/make_golden_eval.py): real text then meets a lattice of real dictionary words; the language
    model stays synthetic.  for_builder: plus EVAL_BUILDER_REQUIRED -- the input of the real KiwiBuilder run behind tests/golden/eval_built_model.raw.xz.
    Returns (raw model path, number of entries the lexicon file holds)."""
    import json
    from dataclasses import replace
    from .synth import SMALL_SPEC, SynthModel
    lex_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eval_data_lexicon.json")
    entries = json.load(open(lex_path, encoding="utf-8"))["entries"]
    os.makedirs(DATA, exist_ok=True)
    path = os.path.join(DATA, "small-eval-builder.raw" if for_builder else "small-eval.raw")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(lex_path):
        words = tuple((f, t) for f, t in entries) + (EVAL_BUILDER_REQUIRED if for_builder else ())
        SynthModel(replace(SMALL_SPEC, extra_words=words)).raw.save(path)
    return path, len(entries)


DIALECT_BITS = {"gyeonggi": 1, "chungcheong": 2, "gangwon": 4, "gyeongsang": 8, "jeolla": 16, "jeju": 32, "hwanghae": 64, "hamgyeong": 128, "pyeongan": 256}      # kiwi::Dialect (include/kiwi/Types.h:320-335)


def dialect_model():
    """The small synthetic model + the gold lexicon of eval_data (as 'small-eval') + the gold (form, tag) pairs of the reference's eval_data/dialect files
    that the standard files do not have, each tagged with the Dialect bits of the files it occurs in (tests/golden/eval_dialect_lexicon.json, written by
    tools/make_golden_dialect.py): a model WITH dialect morphemes -- what AnalyzeOption::allowedDialects / dialectCost and kiwi_init's enabled_dialects
    act on.  Returns the raw model path."""
    import json
    from dataclasses import replace
    from .synth import SMALL_SPEC, SynthModel
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    std = json.load(open(os.path.join(gold, "eval_data_lexicon.json"), encoding="utf-8"))["entries"]
    dia = json.load(open(os.path.join(gold, "eval_dialect_lexicon.json"), encoding="utf-8"))["entries"]
    os.makedirs(DATA, exist_ok=True)
    path = os.path.join(DATA, "small-eval-dialect.raw")
    newest = max(os.path.getmtime(os.path.join(gold, f)) for f in ("eval_data_lexicon.json", "eval_dialect_lexicon.json"))
    if not os.path.exists(path) or os.path.getmtime(path) < newest:
        words = tuple((f, t) for f, t in std) + tuple((f, t, d) for f, t, d in dia)
        SynthModel(replace(SMALL_SPEC, extra_words=words)).raw.save(path)
    return path


def _unpack_model(model_path: str) -> bool:
    """The 'full' models travel xz-compressed (`_data/<spec>.raw.xz`, 35 MB instead of 120: the snapshot of the repository that goes to a GPU box is capped; the
    uncompressed files are listed in .gpurunignore) and are unpacked on first use -- two seconds, against minutes for generating one.  True: the file is there."""
    if os.path.exists(model_path):
        return True
    if not os.path.exists(model_path + ".xz"):
        return False
    import lzma
    import shutil
    tmp = f"{model_path}.{os.getpid()}.tmp"      # (several ranks may get here at once: each unpacks its own copy, the rename is atomic)
    with lzma.open(model_path + ".xz", "rb") as src, open(tmp, "wb") as dst:
        shutil.copyfileobj(src, dst, 1 << 24)
    os.replace(tmp, model_path)
    return True


def _pack_model(model_path: str):
    import subprocess
    try:
        with open(model_path + ".xz.tmp", "wb") as out:
            subprocess.check_call(["xz", "-T0", "-3", "-k", "-c", model_path], stdout=out)
        os.replace(model_path + ".xz.tmp", model_path + ".xz")
    except (OSError, subprocess.CalledProcessError):      # (no xz binary: the uncompressed file alone)
        if os.path.exists(model_path + ".xz.tmp"):
            os.remove(model_path + ".xz.tmp")


def get_workload(name: str):
    """Returns (raw_model_path, list_of_texts, description)."""
    spec_name, n, kw, idx = WORKLOADS[name]
    os.makedirs(DATA, exist_ok=True)
    model_path = os.path.join(DATA, f"{spec_name}.raw")
    corpus_path = os.path.join(DATA, f"{name}.corpus.txt")
    def _stale(path):      # (a corpus cached before the workload's size changed)
        with open(path, encoding="utf-8") as f:
            return f.read().count("\n") + 1 != n
    if not (_unpack_model(model_path) and os.path.exists(corpus_path) and not (name != "c5" and _stale(corpus_path))):
        from .synth import SEED_BASE, SynthModel
        sm = SynthModel(_spec(spec_name))
        sm.raw.save(model_path)
        if spec_name.startswith("full"):
            _pack_model(model_path)
        for wname, (sname, wn, wkw, widx) in WORKLOADS.items():     # the grammar object is expensive: make every corpus of this model now
            if sname != spec_name:
                continue
            texts = sm.make_corpus(wn, SEED_BASE + widx, **wkw)
            with open(os.path.join(DATA, f"{wname}.corpus.txt"), "w", encoding="utf-8") as f:
                f.write("\n".join(texts))
    if name == "c5":      # the c2 corpus, misspelt deterministically
        import random
        _, texts, _ = get_workload("c2")
        rnd = random.Random(SEED_C5)
        texts = [misspell(t, rnd) for t in texts]
        return model_path, texts, f"c5: {n} synthetic sentences (c2 corpus misspelt: confusable vowels, carried-over codas), synthetic '{spec_name}' model (kiwi_amd/synth.py), Knlm, top-1, typo transformer (kiwi_amd.workloads.TYPO_RULES, continual cost 1, threshold 2.5)"
    with open(corpus_path, encoding="utf-8") as f:
        texts = f.read().split("\n")
    assert len(texts) == n, (len(texts), n)
    lm = ("Knlm + SkipBigram, top-3" if spec_name.endswith("-sbg") else "CoNgram (local, 8-bit), top-1" if spec_name.endswith("-cong")
          else "CoNgram global (window 7, 8-bit), top-1" if spec_name.endswith("-cong-global") else "Knlm, top-1")
    desc = f"{name}: {n} synthetic sentences ({kw}), synthetic '{spec_name}' model (kiwi_amd/synth.py), {lm}"
    return model_path, texts, desc


def workload_top_n(name: str) -> int:
    return 3 if WORKLOADS[name][0].endswith("-sbg") else 1


def workload_lm_mode(name: str) -> int:
    """kamd_open_mode's lm_mode for the workload's engine: 4 = CoNgram global (asked for explicitly, as kiwi_init's CONG_GLOBAL / LARGEST do), else 0 (the container's own)."""
    return 4 if WORKLOADS[name][0].endswith("-cong-global") else 0
