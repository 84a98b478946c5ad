"""GPU (-m gpu): the BENCHMARKED configuration itself -- the synthetic 'full' model and the bench corpora of kiwi_amd/workloads.py
(BASELINE config 2: 8192 x 40-jamo; config 3's corpus: mixed 5-200 jamo) -- device vs the real reference translation units
(oracle/_ref, when the prebuilt library travelled) and vs the CPU oracle: tokens, positions and fp32 path scores, bit for bit.
bench.py's numbers are for exactly these outputs."""
import os
from dataclasses import astuple

import pytest

pytestmark = pytest.mark.gpu


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


@pytest.fixture(scope="module")
def full():
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, c2, _ = get_workload("c2")
    _, c3, _ = get_workload("c3")
    dev = KiwiAmd(path)
    orc = oraclelib.OracleKiwi(path)
    ref = refbridge.RefKiwi(path) if refbridge.available() else None
    yield dev, orc, ref, c2, c3
    dev.close()


def _check(dev, cpu, texts, top_n=1):
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    assert len(got) == len(texts)
    bad = [s for s, y in zip(texts, got) if _norm(cpu.analyze(s, top_n=top_n)) != _norm(y)]
    assert not bad, (len(bad), bad[:3])


def test_c2_all_sentences_bit_exact_vs_oracle(full):
    dev, orc, _, c2, _ = full
    _check(dev, orc, c2)


def test_c2_all_sentences_bit_exact_vs_real_reference(full):
    dev, _, ref, c2, _ = full
    if ref is None:
        pytest.skip("oracle/_ref/libkiwi_ref.so not built")
    _check(dev, ref, c2)


def test_c3_4k_sentences_bit_exact_vs_oracle_and_reference(full):
    dev, orc, ref, _, c3 = full
    _check(dev, orc, c3[:4096])
    if ref is not None:
        _check(dev, ref, c3[:4096])


def test_c4_cong_4k_sentences_bit_exact_vs_oracle_and_reference():
    """BASELINE config 4's model type on the benchmarked 'full-cong' model (bench.py --workload c4-cong): device vs the CPU oracle and, where the
    prebuilt x86 reference library travelled, vs the REAL src/CoNgramModel.cpp (SSE4.1 build)."""
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, c4, _ = get_workload("c4-cong")
    dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
    _check(dev, orc, c4[:4096])
    if refbridge.x86_available():
        _check(dev, refbridge.RefKiwi(path, arch=3, x86=True), c4[:4096])
    dev.close()


def test_c2_64k_sample_bit_exact_vs_oracle_and_reference(full):
    """The default bench workload (c2-64k: 65 536 x 40 jamo, the >= 64k-sentence regime of the north-star target): 4096 of its sentences analysed IN a
    batch of 16 384 (so that the launch takes the many-chunk configuration of the position-step kernel), device vs oracle and real reference."""
    from kiwi_amd.workloads import get_workload
    dev, orc, ref, _, _ = full
    _, c64, _ = get_workload("c2-64k")
    batch = c64[:16384]
    got = dev.analyze_batch(batch).to_python()
    assert len(got) == len(batch)
    sample = list(range(0, 16384, 4))
    bad = [batch[i] for i in sample if _norm(orc.analyze(batch[i])) != _norm(got[i])]
    assert not bad, (len(bad), bad[:3])
    if ref is not None:
        bad = [batch[i] for i in sample[:1024] if _norm(ref.analyze(batch[i])) != _norm(got[i])]
        assert not bad, (len(bad), bad[:3])


def test_position_step_and_general_kernel_agree_and_both_run(full, monkeypatch):
    """The search runs as the position-step kernel (k_pos_path) with the general kernel (k_best_path) behind it; KAMD_POS_PATH=0 leaves the general
    kernel alone.  Same analyses, same fp32 scores, from both -- on the benchmark corpus and on the mixed-length one -- and equal to the oracle."""
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    dev, orc, _, c2, c3 = full
    path, _, _ = get_workload("c2")
    monkeypatch.setenv("KAMD_POS_PATH", "0")
    gen = KiwiAmd(path)
    monkeypatch.delenv("KAMD_POS_PATH")
    texts = c2[:2048] + c3[:2048]
    a = dev.analyze_batch(texts).to_python()
    b = gen.analyze_batch(texts).to_python()
    gen.close()
    assert [_norm(x) for x in a] == [_norm(x) for x in b]
    bad = [s for s, y in zip(texts[::8], a[::8]) if _norm(orc.analyze(s)) != _norm(y)]
    assert not bad, (len(bad), bad[:3])


def _packed(dev, texts, **kw):
    r = dev.analyze_batch(texts, **kw)
    a = r.pack()
    r.close()
    return a


@pytest.mark.parametrize("workload,limit", [("c2-64k", 65536), ("c3", 65536), ("c4-cong", 32768)])
def test_position_step_kernel_equals_general_kernel_on_whole_corpora(monkeypatch, workload, limit):
    """Every sentence of the bench corpora -- 65 536 x 40 jamo, the mixed 5-200 jamo corpus, 32 768 of the CoNgram one -- through the position-step
    kernel and through the general kernel alone (KAMD_POS_PATH=0): the packed token records (tokens, positions, fp32 scores) are the same bytes.
    The general kernel is the one the oracle / reference comparisons above pin at sample size; this carries them over to the whole batch, to the
    many-chunk launch configuration (three waves per SIMD) and to whatever the long sentences of c3 reach (steps of more states than the LDS ring)."""
    import numpy as np
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, texts, _ = get_workload(workload)
    texts = texts[:limit]
    pos = KiwiAmd(path)
    a = _packed(pos, texts)
    pos.close()
    monkeypatch.setenv("KAMD_POS_PATH", "0")
    gen = KiwiAmd(path)
    b = _packed(gen, texts)
    gen.close()
    assert a.nbytes == b.nbytes and np.array_equal(a, b)


@pytest.mark.parametrize("workload,limit", [("c2-64k", 65536), ("c3", 32768), ("c4-cong", 32768)])
def test_all_lanes_lattice_kernel_equals_the_one_lane_replay_on_whole_corpora(monkeypatch, workload, limit):
    """k_lattice_wave (round 4: the ops of a chunk decided together by a fixpoint, every lane at work) against k_build_lattice (the reference's
    sequential replay on lane 0, the kernel the lattice dumps pin to the oracle and the real reference at sample size): every sentence of the bench
    corpora, the packed token records -- what the search makes of the lattices, node order and per-node facts included -- are the same bytes; and
    with LDS room for 3/4 match per text unit (KAMD_LATTICE_RATIO=12) the chunks that outgrow it are built by the wide launch, same bytes again."""
    import numpy as np
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, texts, _ = get_workload(workload)
    texts = texts[:limit]
    wave = KiwiAmd(path)
    a = _packed(wave, texts)
    wave.close()
    monkeypatch.setenv("KAMD_LATTICE_RATIO", "12")
    tight = KiwiAmd(path)
    c = _packed(tight, texts[:8192])
    a8 = None
    tight.close()
    monkeypatch.delenv("KAMD_LATTICE_RATIO")
    monkeypatch.setenv("KAMD_LATTICE_WAVE", "0")
    old = KiwiAmd(path)
    b = _packed(old, texts)
    b8 = _packed(old, texts[:8192])
    old.close()
    assert a.nbytes == b.nbytes and np.array_equal(a, b)
    assert c.nbytes == b8.nbytes and np.array_equal(c, b8)


@pytest.mark.parametrize("workload,limit", [("c2-64k", 65536), ("c3", 40000)])
def test_batch_in_parts_equals_the_batch_in_one_piece(monkeypatch, workload, limit):
    """Engine::analyzeBatch sends a large batch through in parts (the host prepares part k + 1 and assembles part k - 1 while part k is searched; launch /
    finish, copy stream, per-batch done events): the packed records of the whole batch are the same bytes as with KAMD_BATCH_PARTS=1, at the scale where
    the parts really are in flight together (the lane emulator, which runs a launch synchronously, checks the concatenation: tests/test_hipemu.py)."""
    import numpy as np
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, texts, _ = get_workload(workload)
    texts = texts[:limit]
    dev = KiwiAmd(path)
    a = _packed(dev, texts)                       # the engine's own choice: four parts
    monkeypatch.setenv("KAMD_BATCH_PARTS", "7")
    c = _packed(dev, texts)
    monkeypatch.setenv("KAMD_BATCH_PARTS", "1")
    b = _packed(dev, texts)
    dev.close()
    assert a.nbytes == b.nbytes and np.array_equal(a, b)
    assert c.nbytes == b.nbytes and np.array_equal(c, b)


def test_c3_sbg_2k_sentences_top3_vs_oracle_and_reference():
    """BASELINE config 3 on its own model: the 'full-sbg' synthetic model (Knlm + SkipBigram), 2048 mixed 5-200 jamo sentences of its
    lexicon (the c3 corpus), top-3 -- device vs the CPU oracle, analysis for analysis with fp32 scores, and vs the REAL reference where its
    library travelled: identical score lists for every text; identical analyses except where exactly tied analyses come out in the reference's
    hash-bucket order (tests/test_oracle_vs_ref.py::test_top_n_matches_reference_up_to_exact_ties states that rule).  The CPU sides run on a thread
    pool (ctypes releases the GIL): SkipBigram lattices on this model take tens of milliseconds per sentence on one core."""
    from concurrent.futures import ThreadPoolExecutor
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    from kiwi_amd.workloads import DATA
    _, texts, _ = get_workload("c3")      # (the same mixed 5-200 jamo sentence generator as c3-sbg over the same lexicon; its corpus file travels anyway)
    path, _, _ = get_workload("c3-sbg")      # (unpacks the model, or generates it: minutes)
    texts = texts[:2048]
    dev = KiwiAmd(path)
    got = dev.analyze_batch(texts, top_n=3).to_python()
    dev.close()
    workers = max(4, min(64, (os.cpu_count() or 8) // 2))

    def cpu_side(make):
        import threading
        local = threading.local()

        def one(s):
            if not hasattr(local, "k"):
                local.k = make()
            return local.k.analyze(s, top_n=3)
        with ThreadPoolExecutor(workers) as ex:
            return list(ex.map(one, texts))

    want = cpu_side(lambda: oraclelib.OracleKiwi(path))
    bad = [s for s, w, y in zip(texts, want, got) if _norm(w) != _norm(y)]
    assert not bad, (len(bad), bad[:2])
    if refbridge.available():
        ref = cpu_side(lambda: refbridge.RefKiwi(path))
        assert all([a[1] for a in r] == [a[1] for a in y] for r, y in zip(ref, got))
        same = sum(_norm(r) == _norm(y) for r, y in zip(ref, got))
        assert same >= 0.6 * len(texts), same


def test_c3_sbg_corpus_8192_sentences_and_the_heaviest_top3_vs_oracle():
    """BASELINE config 3 on the corpus its bench line times (c3-sbg: SkipBigram, top-3): the first 8192 sentences plus the corpus's 64 heaviest -- most
    lattice nodes with more than 512 incoming paths, where the large path container, the top-N key lists of the item table and the N-th-best pruning
    threshold run (tests/golden/c3_sbg_heaviest.json, tools/sbg_heaviest.py) -- device vs the CPU oracle, analysis for analysis with fp32 scores."""
    import json
    from concurrent.futures import ThreadPoolExecutor
    import threading
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, texts, _ = get_workload("c3-sbg")
    heavy = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_sbg_heaviest.json")))
    idx = list(range(8192)) + [h["index"] for h in heavy["heaviest"] if h["index"] >= 8192]
    assert sum(1 for h in heavy["heaviest"][:20] if h["nodesOver512"] > 0) == 20      # (the corpus does have such nodes)
    sample = [texts[i] for i in idx]
    dev = KiwiAmd(path)
    got = dev.analyze_batch(sample, top_n=3).to_python()
    dev.close()
    local = threading.local()

    def one(s):
        if not hasattr(local, "k"):
            local.k = oraclelib.OracleKiwi(path)
        return local.k.analyze(s, top_n=3)
    with ThreadPoolExecutor(max(4, min(32, (os.cpu_count() or 8) // 2))) as ex:
        want = list(ex.map(one, sample))
    bad = [i for i, w, y in zip(idx, want, got) if _norm(w) != _norm(y)]
    assert not bad, (len(bad), bad[:5])


def test_c4_cong_global_4k_sentences_and_the_longest_vs_oracle_and_reference():
    """The reference's largest model type on the model and corpus its bench line times (bench.py `c4-cong-global`: 'full-cong-global', window 7, the c4 corpus):
    the first 4096 sentences plus the corpus's 64 longest (hundreds of paths per node: the medium / large containers, the replay of the reference's container
    behaviour past 64 entries) -- device (kamd_open_mode lm_mode 4) vs the CPU oracle, top-1 and top-3 (first 512), analysis for analysis with fp32 scores; and vs
    the REAL src/CoNgramModel.cpp (SSE4.1 build, useDistantTokens) where the x86 reference library travelled (src/CoNgramModel.cpp:802-868, 1037, 1304, 1470-1490)."""
    from concurrent.futures import ThreadPoolExecutor
    import threading
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import get_workload
    path, texts, _ = get_workload("c4-cong-global")
    longest = sorted(range(len(texts)), key=lambda i: -len(texts[i]))[:64]
    idx = list(range(4096)) + [i for i in longest if i >= 4096]
    sample = [texts[i] for i in idx]
    dev = KiwiAmd(path, lm_mode=4)
    got1 = dev.analyze_batch(sample).to_python()
    got3 = dev.analyze_batch(sample[:512], top_n=3).to_python()
    dev.close()
    local = threading.local()

    def orc():
        if not hasattr(local, "k"):
            local.k = oraclelib.OracleKiwi(path)
            local.k.set_cong_global(True)
        return local.k
    with ThreadPoolExecutor(max(4, min(32, (os.cpu_count() or 8) // 2))) as ex:
        want1 = list(ex.map(lambda s: orc().analyze(s), sample))
        want3 = list(ex.map(lambda s: orc().analyze(s, top_n=3), sample[:512]))
    bad = [i for i, w, y in zip(idx, want1, got1) if _norm(w) != _norm(y)]
    assert not bad, (len(bad), bad[:5])
    bad = [i for i, w, y in zip(idx, want3, got3) if _norm(w) != _norm(y)]
    assert not bad, (len(bad), bad[:5])
    if refbridge.x86_available():
        ref = refbridge.RefKiwi(path, arch=3, model_dir_sbg="cong_global", x86=True)
        sub = list(range(0, 4096, 4)) + list(range(4096, len(sample)))
        bad = [idx[i] for i in sub if _norm(ref.analyze(sample[i])) != _norm(got1[i])]
        assert not bad, (len(bad), bad[:5])
