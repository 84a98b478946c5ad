"""Benchmark / test workloads on the synthetic model.

The real Kiwi model binaries are not available, so every workload uses the deterministic synthetic model of
``kiwi_amd/synth.py``.  Models and corpora are cached under ``_data/`` (git-ignored, shipped to the GPU box).
"""
from __future__ import annotations

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "_data")

WORKLOADS = {
    # name: (spec name, n sentences, corpus kwargs, seed offset)  -- seeds follow BASELINE.md ("KIWI" + config index)
    "c2": ("full", 8192, dict(exact_jamo=40), 2),
    "c3": ("full", 65536, dict(min_jamo=5, max_jamo=200), 3),
    # BASELINE config 3 proper: SkipBigram model, top-3 (device kernel experimental: needs KAMD_EXPERIMENTAL_SBG=1)
    "c3-sbg": ("full-sbg", 65536, dict(min_jamo=5, max_jamo=200), 3),
    "small-c2": ("small", 8192, dict(exact_jamo=40), 2),
    # c2 at the batch size the north-star throughput target is quoted on (>= 64k sentences): the throughput regime
    "c2-64k": ("full", 65536, dict(exact_jamo=40), 12),
}


def _spec(name):
    from .synth import FULL_SBG_SPEC, FULL_SPEC, SMALL_SPEC
    return {"full": FULL_SPEC, "full-sbg": FULL_SBG_SPEC, "small": SMALL_SPEC}[name]


def get_workload(name: str):
    """Returns (raw_model_path, list_of_texts, description)."""
    spec_name, n, kw, idx = WORKLOADS[name]
    os.makedirs(DATA, exist_ok=True)
    model_path = os.path.join(DATA, f"{spec_name}.raw")
    corpus_path = os.path.join(DATA, f"{name}.corpus.txt")
    if not (os.path.exists(model_path) and os.path.exists(corpus_path)):
        from .synth import SEED_BASE, SynthModel
        sm = SynthModel(_spec(spec_name))
        sm.raw.save(model_path)
        for wname, (sname, wn, wkw, widx) in WORKLOADS.items():     # the grammar object is expensive: make every corpus of this model now
            if sname != spec_name:
                continue
            texts = sm.make_corpus(wn, SEED_BASE + widx, **wkw)
            with open(os.path.join(DATA, f"{wname}.corpus.txt"), "w", encoding="utf-8") as f:
                f.write("\n".join(texts))
    with open(corpus_path, encoding="utf-8") as f:
        texts = f.read().split("\n")
    assert len(texts) == n, (len(texts), n)
    lm = "Knlm + SkipBigram, top-3" if spec_name.endswith("-sbg") else "Knlm, top-1"
    desc = f"{name}: {n} synthetic sentences ({kw}), synthetic '{spec_name}' model (kiwi_amd/synth.py), {lm}"
    return model_path, texts, desc


def workload_top_n(name: str) -> int:
    return 3 if WORKLOADS[name][0].endswith("-sbg") else 1
