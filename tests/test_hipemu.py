"""CPU (not gpu): the product's HIP kernels EXECUTED on the host, lane by lane, by the test-only emulator of tests/hipemu
(the product's sources compiled as C++ against a stand-in <hip/hip_runtime.h>: lanes are fibers, cross-lane operations are
rendezvous of a convergence domain -- see tests/hipemu/include/hip/hip_runtime.h), called through the same C ABI and compared
with the CPU oracle exactly like the GPU parity suite does.

What this adds to `-m gpu`: the kernels' logic -- every lane-group instantiation, the fallback paths of the `smallcaps`
configuration, top-N, and the SkipBigram kernel that has not seen a GPU yet -- is checked wherever the CPU suite runs, and a
cross-lane operation in divergent control flow aborts with a lane-by-lane report instead of hanging a GPU.  What it cannot
show: anything about timing, occupancy, the compiler's gfx950 code, or memory-model races.  The emulated library is never
the product: only this file loads it (explicit lib_path); kiwi_amd/ has no reference to it."""
import os
import subprocess
from dataclasses import astuple

import pytest

from corpora import EDGE_TEXTS, dictionary_mix, force_lanes, fuzzed, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hipemu")


@pytest.fixture(scope="module")
def emu_libs():
    subprocess.check_call(["make", "-C", EMU, "-j8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", EMU, "smallcaps", "-j8"], stdout=subprocess.DEVNULL)
    return os.path.join(EMU, "_build", "libkiwi_hipemu.so"), os.path.join(EMU, "_build", "libkiwi_hipemu_smallcaps.so")


def _norm(res):
    return [([astuple(t) for t in a[0]], a[1]) for a in res]


def _check(dev, orc, texts, top_ns=(1,)):
    for top_n in top_ns:
        got = dev.analyze_batch(texts, top_n=top_n).to_python()
        for s, y in zip(texts, got):
            assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (top_n, s)


def test_the_product_library_is_not_the_emulator(emu_libs):
    from kiwi_amd import api
    assert "hipemu" not in api.LIB_PATH and os.path.basename(api.LIB_PATH) == "libkiwi_hip.so"
    for root, _, files in os.walk(os.path.join(os.path.dirname(HERE), "kiwi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                assert "hipemu" not in open(os.path.join(root, f), encoding="utf-8", errors="ignore").read(), f


@pytest.mark.parametrize("lanes,wps", [("pos", "2"), ("pos", "3"), ("pos8", "2"), ("pos8", "3"), ("16", "2"), ("16", "3"), ("8", "3"), ("8", "2"), ("4", "2"), ("32", "2"), ("64", "2")])
def test_emulated_knlm_kernels_match_oracle(emu_libs, oracle, small_model, monkeypatch, lanes, wps):
    """Dictionary scan, lattice build (LDS and HBM variants), candidate expansion, best-path search in every lane-group
    instantiation, end stage: tokens, positions and fp32 scores equal the oracle's, top-1 and top-2."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    force_lanes(monkeypatch, lanes)
    if lanes in ("8", "16", "pos", "pos8"):
        monkeypatch.setenv("KAMD_WPS", wps)
    n = 100 if lanes in ("16", "pos", "pos8") and wps == "2" else 40
    texts = synthetic(sm, n, 521, min_jamo=5, max_jamo=120) + dictionary_mix(sm, n // 2, 522) + (EDGE_TEXTS + fuzzed(sm, 150, 523) if n == 100 else [])      # (fuzzed: lone surrogates, pattern fragments, other scripts)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    _check(dev, oracle, texts, (1, 2))
    if n == 100:
        for s in texts[:60]:
            if s.strip():
                assert dev.split(s) == oracle.split(s), s
    dev.close()


@pytest.mark.parametrize("tiny", [False, True])
def test_emulated_batch_in_parts(emu_libs, oracle, small_model, monkeypatch, tiny):
    """Engine::analyzeBatch cuts a large batch into parts whose host stages overlap the kernels of their neighbours (Engine::launch / finish): the parts'
    result segments concatenate into the batch's -- same packed bytes as the batch in one piece, texts of several chunks (other start states: re-searched
    while later parts are in flight) and, with tiny arenas, the capacity ladder inside a part included."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    if tiny:
        monkeypatch.setenv("KAMD_TEST_TINY_ARENAS", "1")
    texts = synthetic(sm, 300, 1201, min_jamo=5, max_jamo=120) + EDGE_TEXTS + dictionary_mix(sm, 100, 1202)
    texts += [". ".join(texts[k:k + 6]) + '." \'' + texts[k + 7] + "'" for k in range(0, 120, 8)]      # several chunks per text, quotes carried across them
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    monkeypatch.setenv("KAMD_BATCH_PARTS", "1")
    whole = dev.analyze_batch(texts, top_n=2)
    one = whole.to_python(); packed_one = whole.pack()
    monkeypatch.setenv("KAMD_BATCH_PARTS", "3")
    parts = dev.analyze_batch(texts, top_n=2)
    assert bytes(parts.pack()) == bytes(packed_one)
    for s, y in zip(texts, parts.to_python()):
        assert _norm(oracle.analyze(s, top_n=2)) == _norm(y), s
    assert len(one) == len(texts)
    dev.close()


@pytest.mark.parametrize("match_extra", [0, 1 << 24, 1 << 17])
def test_emulated_fast_assembly_equals_the_general_assembly(emu_libs, oracle, small_model, monkeypatch, match_extra):
    """Result assembly of the common case (one chunk, top-1: kiwi_amd/csrc/post_fast.hpp, packed records straight from the device's token records through
    a per-morpheme template table) against the general assembly (post.cpp, the code the oracle runs): the same packed bytes on texts with brackets, quotes,
    line breaks, several sentences, other scripts and lone surrogates; with compatibility jamo; and with an affix-joining option (which the fast path must
    leave to the general one).  Oracle == device on the same texts."""
    from kiwi_amd.api import KiwiAmd, MATCH_ALL_WITH_NORMALIZING
    sm, path = small_model
    texts = synthetic(sm, 300, 1301, min_jamo=5, max_jamo=120) + EDGE_TEXTS + dictionary_mix(sm, 150, 1302) + fuzzed(sm, 150, 1303)
    texts += ["(" + texts[k] + ') "' + texts[k + 1] + '"\n\n' + texts[k + 2] + "\r\n[" + texts[k + 3] + "]" for k in range(0, 80, 4)]
    match = MATCH_ALL_WITH_NORMALIZING | match_extra
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    monkeypatch.setenv("KAMD_FAST_ASSEMBLY", "0")
    general = dev.analyze_batch(texts, top_n=1, match=match)
    packed_general = bytes(general.pack()); py_general = general.to_python()
    monkeypatch.delenv("KAMD_FAST_ASSEMBLY")
    fast = dev.analyze_batch(texts, top_n=1, match=match)
    assert bytes(fast.pack()) == packed_general
    if not match_extra:
        for s, y in zip(texts, py_general):
            assert _norm(oracle.analyze(s, top_n=1)) == _norm(y), s
    dev.close()


@pytest.mark.parametrize("lanes", ["pos", "16", "64"])
def test_emulated_order_4_knlm(emu_libs, small_order4_model, monkeypatch, lanes):
    """An order-4 Knlm (the reference's maximum): back-off chains one node longer than the pair a search state carries (ModelView::lmChain) -- the
    one-round-trip probe of the position-step kernel hands the tail to the general walk.  Oracle == real reference on the same model first."""
    import oraclelib
    import refbridge
    from kiwi_amd.api import KiwiAmd
    sm, path = small_order4_model
    force_lanes(monkeypatch, lanes)
    orc = oraclelib.OracleKiwi(path)
    texts = synthetic(sm, 120, 1111, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 50, 1112) + EDGE_TEXTS[:60] + fuzzed(sm, 60, 1113)
    if lanes == "pos" and refbridge.available():
        ref = refbridge.RefKiwi(path)
        for t in texts + synthetic(sm, 400, 1114, min_jamo=5, max_jamo=150):
            assert _norm(ref.analyze(t)) == _norm(orc.analyze(t)), t
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    _check(dev, orc, texts, (1, 3))
    dev.close()


@pytest.mark.parametrize("lanes", ["pos", "pos8", "16", "8", "64"])
def test_emulated_fallback_paths_with_small_capacities(emu_libs, small_model, monkeypatch, lanes):
    """The `smallcaps` configuration (LDS capacities of 4) with the container limits cut to 3 / 8 / 2 on both sides:
    medium / large containers, HBM work-item queue, HBM pruning path, far-back node lookup."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    force_lanes(monkeypatch, lanes)
    monkeypatch.setenv("KAMD_CONTAINER_LIMITS", "3,8,2")
    orc = oraclelib.OracleKiwi(path)
    orc.set_container_limits(3, 8, 2)
    dev = KiwiAmd(path, lib_path=emu_libs[1])
    _check(dev, orc, synthetic(sm, 50, 531, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 25, 532), (1, 2))
    dev.close()


def test_emulated_lattice_build_with_four_chunks_per_wavefront(emu_libs, oracle, small_model, monkeypatch):
    """k_build_lattice<16> (the engine's choice for batches of >= 32768 chunks; KAMD_LATTICE_GROUP=16 here): four chunks share a wavefront, each
    replayed by lane 0 of its 16-lane group.  Lattices (split) and analyses against the oracle, mixed lengths, the edge texts, and a batch whose
    size is not a multiple of four; KAMD_LATTICE_LDS=6000 leaves the longer chunks to the thread-per-chunk kernel in the same batch."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    texts = (synthetic(sm, 61, 543, min_jamo=5, max_jamo=150) + dictionary_mix(sm, 30, 544) + EDGE_TEXTS)[:-1]
    monkeypatch.setenv("KAMD_LATTICE_GROUP", "16")
    for budget in (None, "6000"):
        if budget:
            monkeypatch.setenv("KAMD_LATTICE_LDS", budget)
        dev = KiwiAmd(path, lib_path=emu_libs[0])
        _check(dev, oracle, texts)
        for t in texts[:40]:
            if t.strip():
                assert dev.split(t) == oracle.split(t), t
        dev.close()


def test_emulated_all_lanes_lattice_kernel(emu_libs, oracle, small_model, monkeypatch):
    """k_lattice_wave (lattice_wave.hip; the engine's default): the lattice dumps and the analyses of mixed-length, dictionary-mix, edge and fuzzed
    texts (surrogate pairs, emoji modifiers and pattern spans send a chunk through the one-lane character-type pass, everything else through the
    one-unit-per-lane pass) equal the oracle's; with LDS room for a quarter match per text unit (KAMD_LATTICE_RATIO=4) every chunk outgrows the first
    launch and is built by the wide one, with KAMD_LATTICE_LDS=3000 the longer chunks go to the thread-per-chunk replay; and the one-lane replay
    kernel (KAMD_LATTICE_WAVE=0) still answers the same."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    texts = synthetic(sm, 60, 551, min_jamo=5, max_jamo=200) + dictionary_mix(sm, 30, 552) + EDGE_TEXTS + fuzzed(sm, 120, 553)
    for env in ({}, {"KAMD_LATTICE_RATIO": "4"}, {"KAMD_LATTICE_LDS": "3000"}, {"KAMD_LATTICE_WAVE": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev = KiwiAmd(path, lib_path=emu_libs[0])
        _check(dev, oracle, texts[:120] if env else texts)
        for t in (texts if not env else texts[::3]):
            if t.strip():
                assert dev.split(t) == oracle.split(t), (env, t)
        dev.close()
        for k in env:
            monkeypatch.delenv(k)


def test_emulated_lattice_hbm_kernel_and_rerun_ladder(emu_libs, oracle, small_model, monkeypatch):
    """KAMD_LATTICE_LDS=0 sends every chunk to the thread-per-chunk lattice kernel; KAMD_TEST_TINY_ARENAS makes most chunks
    climb the capacity ladder (re-runs with 4x / 16x / 64x arenas)."""
    from kiwi_amd.api import KiwiAmd
    sm, path = small_model
    texts = synthetic(sm, 40, 541, min_jamo=5, max_jamo=120) + dictionary_mix(sm, 20, 542)
    monkeypatch.setenv("KAMD_LATTICE_LDS", "0")
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    _check(dev, oracle, texts)
    dev.close()
    monkeypatch.delenv("KAMD_LATTICE_LDS")
    monkeypatch.setenv("KAMD_TEST_TINY_ARENAS", "1")
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    _check(dev, oracle, texts)
    # the ladder is climbed inside run(), by all overflowing chunks of the batch together; kamd_batch_reruns reports how many there were
    b = dev.stage(texts)
    dev.run(b)
    assert 0 < dev.reruns(b)[0] <= b.info()["chunks"]
    b.close()
    dev.close()
    monkeypatch.delenv("KAMD_TEST_TINY_ARENAS")
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    b = dev.stage(texts)
    dev.run(b)
    assert dev.reruns(b) == (0, 0.0)
    b.close()
    dev.close()


@pytest.mark.parametrize("lanes,top_n", [("16", 1), ("16", 2), ("16", 3), ("64", 1), ("64", 2)])
def test_emulated_skipbigram_kernel_matches_oracle(emu_libs, small_sbg_model, monkeypatch, lanes, top_n):
    """The SkipBigram search kernel (viterbi_kernel_sbg.hip; gated on the device until it has passed tests/test_gpu_sbg.py on a
    GPU): history rings in the LM state, ring-aware container keys (whole ring for top-1, last four words and no root for top-N),
    glibc-exact exp / log -- against the oracle, which is pinned to the real reference's SkipBigram path.  Nodes of these lattices
    carry up to thousands of incoming paths, so the large-container / HBM-queue path does most of the work here."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_sbg_model
    monkeypatch.setenv("KAMD_EXPERIMENTAL_SBG", "1")
    force_lanes(monkeypatch, lanes)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    n = 40 if top_n == 1 else 60
    texts = synthetic(sm, n, 551, min_jamo=5, max_jamo=70 if top_n == 1 else 120) + dictionary_mix(sm, n // 2, 552) + EDGE_TEXTS
    _check(dev, orc, texts, (top_n,))
    dev.close()


@pytest.mark.parametrize("model,lanes,top_n,pool", [("sbg", "64", 1, "4096"), ("sbg", "16", 3, "4096"), ("sbg", "64", 2, "6"), ("knlm", "pos", 1, "2048"), ("knlm", "16", 2, "2048")])
def test_emulated_state_arenas_grow_into_the_pool(emu_libs, small_model, small_sbg_model, monkeypatch, model, lanes, top_n, pool):
    """Arenas of 1/64 of the worst case (KAMD_STATE_SCALE=1): most chunks fill theirs and carry on in arenas from the batch's pool (growArena: the states so far
    move, the node that met the full arena is evaluated again; the end stage grows the same way) -- same analyses.  Pool of 6/64 of the arenas: it runs out, and
    the chunks it could not serve go through the re-run ladder.  SkipBigram: the arenas are the search kernel's lane groups' (WorkView::slotCap; a group keeps
    what it grew into for its later chunks) and the kernel runs every chunk's end stage itself (finishPathsSolo)."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_sbg_model if model == "sbg" else small_model
    monkeypatch.setenv("KAMD_EXPERIMENTAL_SBG", "1")
    force_lanes(monkeypatch, lanes)
    monkeypatch.setenv("KAMD_STATE_SCALE", "1")
    monkeypatch.setenv("KAMD_STATE_POOL", pool)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    texts = [t for t in synthetic(sm, 50, 571, min_jamo=20, max_jamo=90) + dictionary_mix(sm, 20, 572) if t.strip()]
    b = dev.stage(texts)
    got = dev.fetch(b, top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), s
    p = b.pool()
    reruns, _ = dev.reruns(b)
    assert p["pool_states"] >= p["arena_states"] * int(pool) // 64 and p["pool_asked"] > p["arena_states"] // 2, p
    if pool == "6":
        assert p["pool_asked"] > p["pool_states"] and reruns > 0, (p, reruns)
    else:
        assert p["pool_asked"] <= p["pool_states"] and (reruns == 0 or model == "knlm"), (p, reruns)      # (the position-step kernel's own end stage does not grow)
    b.close()
    dev.close()


@pytest.mark.parametrize("lanes", ["16", "64"])
def test_emulated_skipbigram_fallback_paths(emu_libs, small_sbg_model, monkeypatch, lanes):
    """SkipBigram kernel in the `smallcaps` configuration: tiny LDS capacities, container limits 3 / 8 / 2 on both sides (the medium
    container's bucket hash chains the ring words), and a constant ring digest, so that every pair of items with equal packed
    keys reaches the exact ring comparison and the register path hands colliding batches over to the scanning path."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_sbg_model
    monkeypatch.setenv("KAMD_EXPERIMENTAL_SBG", "1")
    force_lanes(monkeypatch, lanes)
    monkeypatch.setenv("KAMD_CONTAINER_LIMITS", "3,8,2")
    orc = oraclelib.OracleKiwi(path)
    orc.set_container_limits(3, 8, 2)
    dev = KiwiAmd(path, lib_path=emu_libs[1])
    _check(dev, orc, synthetic(sm, 30, 561, min_jamo=5, max_jamo=70) + dictionary_mix(sm, 15, 562), (1, 2))
    dev.close()


def test_emulated_skipbigram_golden_sequence(emu_libs, small_sbg_model, monkeypatch):
    """Emulated SkipBigram kernel against the committed outputs of the REAL reference (tests/golden/small_sbg_model_sequence.json).
    The reference's large container hands equal-score paths on in a history-dependent order (top-N), so a tie may pick
    another analysis; the best score of every text must still be the reference's, bit for bit."""
    import json
    from kiwi_amd.api import KiwiAmd
    g = json.load(open(os.path.join(HERE, "golden", "small_sbg_model_sequence.json"), encoding="utf-8"))
    monkeypatch.setenv("KAMD_EXPERIMENTAL_SBG", "1")
    dev = KiwiAmd(small_sbg_model[1], lib_path=emu_libs[0])
    items = [it for it in g["items"] if len(it["text"]) <= 60][:60]
    got = dev.analyze_batch([it["text"] for it in items], top_n=g["top_n"]).to_python()
    exact = 0
    for it, y in zip(items, got):
        assert y[0][1] == it["analyses"][0]["score"], it["text"]
        toks = [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.sense_id, t.morph_id, t.score] for t in y[0][0]]
        exact += toks == it["analyses"][0]["tokens"]
    assert exact >= 0.9 * len(items), (exact, len(items))
    dev.close()


@pytest.mark.parametrize("devices", ["2"])      # (one device: the same suite runs on the real GPU, tests/test_gpu_capi.py; two emulated devices add the replica / split path)
def test_emulated_c_api_suite(emu_libs, devices):
    """The drop-in boundary on the CPU: tests/test_gpu_capi.py (Kiwi's own C API bound with ctypes, reader / receiver protocol,
    8 concurrent callers on one handle, a gcc-built C client) re-run against the emulated build of the same sources.  With two
    (emulated) devices visible kiwi_init builds an engine replica per device and kiwi_analyze_m / _mw spread every batch over both,
    one host thread each, delivering in input order -- the one-process-drives-all-GPUs path behind the boundary."""
    import sys
    env = dict(os.environ, KAMD_TEST_LIB=emu_libs[0], HIPEMU_DEVICES=devices)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_capi.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_emulated_pretokenized_spans_suite(emu_libs):
    """Pretokenized spans on the device path, on the CPU: tests/test_gpu_pretokenized.py (all 190 golden analyses of the real reference through the batch ABI and
    through kiwi_analyze_w with a kiwi_pretokenized_h, 480 fresh cases against the live reference and the oracle, byte offsets through kiwi_analyze, the error
    convention, SkipBigram / CoNgram / character-model variants) re-run against the emulated build of the same sources: the span entries behind a chunk's
    patterns, the forced node of the splitter's replay, the per-batch overlay of temporary forms / morphemes behind the model's tables."""
    import sys
    env = dict(os.environ, KAMD_TEST_LIB=emu_libs[0])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_pretokenized.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-p", "no:xdist"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("kind", ["knlm-tiny-arenas", "sbg", "typo", "chr", "sbgtypo", "congg"])
def test_emulated_kernels_under_address_and_ub_sanitizers(kind):
    """The emulated kernels compiled with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C tests/hipemu asan`): an access
    outside an HBM buffer or beyond the dynamic LDS a launch asked for -- which a GPU executes silently -- ends the run.  Knlm with
    KAMD_TEST_TINY_ARENAS (most chunks first hit a capacity limit: the code that must stop BEFORE writing out of bounds), and the
    SkipBigram kernel."""
    import sys
    subprocess.check_call(["make", "-C", EMU, "asan", "-j8"], stdout=subprocess.DEVNULL)
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no shared AddressSanitizer runtime next to the compiler")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    if kind == "sbg":
        env["KAMD_EXPERIMENTAL_SBG"] = "1"
    elif kind not in ("chr", "sbgtypo", "congg"):
        env["KAMD_TEST_TINY_ARENAS"] = "1"
    if kind == "typo":
        env["KAMD_EXPERIMENTAL_TYPO"] = "1"
    r = subprocess.run([sys.executable, os.path.join(EMU, "sanitizer_run.py"), os.path.join(EMU, "_build", "libkiwi_hipemu_asan.so"), kind if kind in ("sbg", "typo", "chr", "sbgtypo", "congg") else "knlm"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "sanitizer run complete" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr.replace("WARNING: ASan doesn't fully support makecontext/swapcontext", ""), r.stderr[-3000:]


def _tsan_run(kind, extra_env):
    import sys
    subprocess.check_call(["make", "-C", EMU, "tsan", "-j8"], stdout=subprocess.DEVNULL)
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no shared ThreadSanitizer runtime next to the compiler")
    env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(EMU, "sanitizer_run.py"), os.path.join(EMU, "_build", "libkiwi_hipemu_tsan.so"), kind],
                       env=env, capture_output=True, text=True)
    if "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot run in this address-space layout")
    return r


@pytest.mark.parametrize("kind", ["knlm", "sbg", "typo", "chr"])
def test_emulated_kernels_have_no_cross_lane_data_races(kind):
    """The emulated kernels under ThreadSanitizer with every lane a TSan fiber (tests/hipemu/emu.cpp): two lanes touching one LDS /
    HBM location, at least one writing non-atomically, with no rendezvous (ballot, shuffle, DPP, wave barrier, __syncthreads) in
    between is a reported race -- i.e. every cross-lane hand-over in the kernels is fenced by construction, not by the luck of
    lock-step execution."""
    r = _tsan_run(kind, {"KAMD_EXPERIMENTAL_SBG": "1"} if kind == "sbg" else {"KAMD_EXPERIMENTAL_TYPO": "1"} if kind == "typo" else {})
    assert r.returncode == 0 and "sanitizer run complete" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ThreadSanitizer: data race" not in r.stderr, r.stderr[:6000]


def test_race_detector_sees_a_dropped_wave_barrier():
    """Self-test of the above: with the wave barriers of one kernel turned into nothing (HIPEMU_TEST_DROP_WAVE_BARRIER) the same run
    must report races (what the kernel then computes, or whether it survives, does not matter)."""
    r = _tsan_run("knlm", {"HIPEMU_TEST_DROP_WAVE_BARRIER": "k_lattice_wave"})      # (the lattice kernel the engine runs by default: its phases exchange everything through LDS)
    assert "ThreadSanitizer: data race" in r.stderr, r.stdout[-1500:] + r.stderr[-1500:]


def _typo_lattices(dev, typo, text, threshold, dialect=0):
    import ctypes as C
    import numpy as np
    import oraclelib
    L = dev.lib
    L.kamd_typo_lattices.restype = C.c_size_t
    L.kamd_typo_lattices.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t]
    u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
    need = L.kamd_typo_lattices(dev.h, typo.h, threshold, dialect, u.ctypes.data, len(u), oraclelib.MATCH_ALL_WITH_NORMALIZING, None, 0)
    assert need, L.kamd_last_error()
    buf = np.zeros(need, np.uint8)
    L.kamd_typo_lattices(dev.h, typo.h, threshold, dialect, u.ctypes.data, len(u), oraclelib.MATCH_ALL_WITH_NORMALIZING, buf.ctypes.data, need)
    r = oraclelib._Reader(buf.tobytes())
    chunks = []
    for _ in range(r.get("I")):
        n, split_end = r.get("II")
        chunks.append((split_end, [r.get("IIIIiIIIf") for _ in range(n)]))
    return chunks


@pytest.mark.parametrize("rules,threshold", [("own", 2.5), ("own-continual", 2.5), ("own-continual", 1.2)])
def test_emulated_typo_lattice_kernel_matches_oracle(emu_libs, small_model, rules, threshold):
    """k_build_lattice_typo (typo graph from the product's host module, multi-state lattice build on the device) against the oracle's
    lattice over a typo graph -- which is pinned to the real reference by tests/test_typo_oracle.py -- on misspelt texts: nodes, links,
    positions, typo costs.  A building block: the analyze path does not use it yet."""
    import random
    import oraclelib
    import test_typo_product
    from kiwi_amd.api import KiwiAmd
    from typo_cases import COND, INF, RULES, misspell
    sm, path = small_model
    cont = 1.0 if "continual" in rules else INF
    test_typo_product.LIB = emu_libs[0]
    prod = test_typo_product.ProductTypo(cont, INF)
    orc_t = oraclelib.OracleTypo(cont, INF)
    for origs, errs, cost, cond, dia in RULES:
        for o in origs:
            for e in errs:
                assert prod.add(o, e, cost, COND[cond], dia) == 0
                orc_t.add(o, e, cost, COND[cond], dia)
    prod.prepare(True); orc_t.prepare(True)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    orc = oraclelib.OracleKiwi(path)
    rnd = random.Random(3)
    n_typo_nodes = 0
    for t in [misspell(t, rnd, True, "continual" in rules) for t in synthetic(sm, 60, 571, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 30, 572)] + EDGE_TEXTS:
        if not t.strip():
            continue
        want = orc.split_typo(orc_t, t, threshold)
        assert _typo_lattices(dev, prod, t, threshold) == want, t
        n_typo_nodes += sum(1 for c in want for nd in c[1] if nd[8] > 0)
    assert n_typo_nodes > 50
    dev.close(); prod.close()


def _analyze_typo(dev, typo, texts, threshold, top_n=1, dialect=0, match=None):
    import ctypes as C
    import numpy as np
    import oraclelib
    from kiwi_amd.api import Results
    L = dev.lib
    L.kamd_analyze_batch_typo.restype = C.c_void_p
    L.kamd_analyze_batch_typo.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
    enc = [np.frombuffer(t.encode("utf-16-le", errors="surrogatepass"), np.uint16) for t in texts]
    offs = np.zeros(len(enc) + 1, np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    flat = np.concatenate(enc) if enc else np.zeros(0, np.uint16)
    r = L.kamd_analyze_batch_typo(dev.h, typo.h, threshold, dialect, flat.ctypes.data, offs.ctypes.data, len(texts), top_n, oraclelib.MATCH_ALL_WITH_NORMALIZING if match is None else match, 0, 0)
    if not r:
        raise RuntimeError(L.kamd_last_error().decode())
    return Results(L, r).to_python()


def _typo_pair(lib, continual, lengthening=float("inf")):
    import oraclelib
    import test_typo_product
    from typo_cases import COND, RULES
    test_typo_product.LIB = lib
    prod = test_typo_product.ProductTypo(continual, lengthening)
    orc_t = oraclelib.OracleTypo(continual, lengthening)
    for origs, errs, cost, cond, dia in RULES:
        for o in origs:
            for e in errs:
                assert prod.add(o, e, cost, COND[cond], dia) == 0
                orc_t.add(o, e, cost, COND[cond], dia)
    prod.prepare(True); orc_t.prepare(True)
    return prod, orc_t


@pytest.mark.parametrize("continual,threshold,top_n,lanes,tiny,lengthening", [(float("inf"), 2.5, 1, "16", False, float("inf")), (1.0, 2.5, 1, "16", False, float("inf")),
                                                                               (1.0, 1.2, 3, "16", False, float("inf")), (1.0, 2.5, 1, "64", False, float("inf")),
                                                                               (1.0, 2.5, 2, "16", True, float("inf")), (1.0, 2.5, 1, "16", False, 0.25), (float("inf"), 4.0, 2, "64", False, 0.25),
                                                                               (1.0, 2.5, 1, "16", "smallcaps", 0.25), (1.0, 2.5, 1, "16", "budget", float("inf")),
                                                                               (1.0, 2.5, 1, "pos", False, 0.25), (float("inf"), 2.5, 1, "pos", False, float("inf")), (1.0, 2.5, 1, "pos", "smallcaps", 0.25)])
def test_emulated_typo_analyses_match_oracle(emu_libs, small_model, monkeypatch, continual, threshold, top_n, lanes, tiny, lengthening):
    """The whole typo-correcting analysis on the (emulated) device -- typo graphs from the host module, k_build_lattice_typo, the search kernel
    compiled with node typo costs (viterbi_kernel_typo.hip), end stage, host post-processing -- against the oracle (pinned to the real
    reference): tokens, positions, fp32 scores, per-token typo costs; continual and lengthening typos; also through the capacity ladder."""
    import random
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_model
    monkeypatch.setenv("KAMD_EXPERIMENTAL_TYPO", "1")
    force_lanes(monkeypatch, lanes)
    lib = emu_libs[0]
    if tiny is True:
        monkeypatch.setenv("KAMD_TEST_TINY_ARENAS", "1")
    elif tiny == "smallcaps":      # LDS node lists far too small: the wave-per-chunk lattice kernel hands most chunks to the thread-per-chunk one
        lib = emu_libs[1]
    elif tiny == "budget":         # an LDS budget only the shortest chunks fit
        monkeypatch.setenv("KAMD_LATTICE_LDS", "3000")
    prod, orc_t = _typo_pair(lib, continual, lengthening)
    dev = KiwiAmd(path, lib_path=lib)
    orc = oraclelib.OracleKiwi(path)
    rnd = random.Random(7)
    texts = [misspell(t, rnd, True, continual == 1.0, lengthening < 1e9) for t in synthetic(sm, 50, 581, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 25, 582)] + EDGE_TEXTS
    got = _analyze_typo(dev, prod, texts, threshold, top_n)
    corrected = 0
    for t, y in zip(texts, got):
        want = orc.analyze_typo(orc_t, t, threshold, 0, top_n=top_n)
        assert _norm(want) == _norm(y), t
        corrected += any(x.typo_cost > 0 for x in want[0][0])
    assert corrected >= 5
    dev.close(); prod.close()



@pytest.mark.parametrize("lanes,top_n", [("pos", 1), ("pos8", 1), ("16", 1), ("64", 1), ("16", 2), ("16", 3)])
def test_emulated_cong_kernels_match_oracle(emu_libs, small_cong_model, monkeypatch, lanes, top_n):
    """The search kernel compiled for CoNgram models (viterbi_kernel_cong.hip: context trie + int8 embedding dot product, candidates in the
    transposed evaluator's order, the reference kernel's rounding per node) against the oracle, whose CoNgram path is pinned to the REAL
    src/CoNgramModel.cpp (tests/test_cong_oracle.py): tokens, positions, fp32 scores."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_model
    force_lanes(monkeypatch, lanes)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    texts = synthetic(sm, 80, 911, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 40, 912) + EDGE_TEXTS
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (lanes, top_n, s)
    dev.close()


def test_emulated_cong_fallback_paths_with_small_capacities(emu_libs, small_cong_model, monkeypatch):
    """The small-capacity build (LDS queues of 4, container limits 3 / 8 / 2 on both sides): medium / large containers with the CoNgram
    state hash, HBM work-item queues carrying context ids."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_model
    monkeypatch.setenv("KAMD_CONTAINER_LIMITS", "3,8,2")
    orc = oraclelib.OracleKiwi(path)
    orc.set_container_limits(3, 8, 2)
    dev = KiwiAmd(path, lib_path=emu_libs[1])
    texts = synthetic(sm, 50, 913, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 25, 914)
    got = dev.analyze_batch(texts).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s)) == _norm(y), s
    dev.close()


@pytest.mark.parametrize("model,top_n", [("knlm", 1), ("knlm", 2), ("cong", 1), ("sbg", 1)])
def test_emulated_blocklist_matches_oracle(emu_libs, small_model, small_cong_model, small_sbg_model, model, top_n):
    """AnalyzeOption::blocklist on the (emulated) device -- kamd_morphset_add = Kiwi::findMorphemes, one bit per morpheme with Morpheme::hasMorpheme
    folded in, k_expand_cands dropping blocked candidates from a node's list (also in the transposed CoNgram order) -- against the oracle,
    whose blocklist path is pinned to the real reference (tests/test_oracle_vs_ref.py::test_blocklist_matches_reference)."""
    import oraclelib
    from corpora import pick_blocklist
    from kiwi_amd.api import KiwiAmd
    sm, path = {"knlm": small_model, "cong": small_cong_model, "sbg": small_sbg_model}[model]
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    texts = synthetic(sm, 60, 931, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 30, 932) + EDGE_TEXTS[:20]
    items = pick_blocklist(orc, texts, 20) + [("없는형태", 1)]
    ms, found = dev.morphset(items)
    assert found == orc.set_blocklist(items) and found[-1] == 0
    got = dev.analyze_batch_opt(texts, top_n=top_n, blocklist=ms).to_python()
    changed = 0
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (model, top_n, s)
    plain = dev.analyze_batch(texts, top_n=top_n).to_python()      # the list is per call: the next call without it is unconstrained again
    orc.set_blocklist([])
    for s, y, z in zip(texts, plain, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), s
        changed += _norm(y) != _norm(z)
    assert changed >= 20
    ms.close(); dev.close()


@pytest.mark.parametrize("lanes,top_n,continual,lengthening", [("pos", 1, 1.0, 0.25), ("pos8", 1, 1.0, 0.25), ("16", 1, 1.0, float("inf")), ("64", 1, 1.0, 0.25), ("16", 2, float("inf"), float("inf"))])
def test_emulated_typo_correction_with_a_cong_model(emu_libs, small_cong_model, monkeypatch, lanes, top_n, continual, lengthening):
    """Typo correction on a CoNgram model (the reference's default model type with its --typo configurations): the fifth compilation of the search
    kernel (viterbi_kernel_cong_typo.hip: CoNgram scoring + node typo costs) over the typo lattices, against the oracle -- pinned for this
    combination to the real reference's SSE4.1 build (tests/test_cong_oracle.py::test_typo_correction_with_a_cong_model_equals_reference)."""
    import random
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_cong_model
    force_lanes(monkeypatch, lanes)
    prod, orc_t = _typo_pair(emu_libs[0], continual, lengthening)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    orc = oraclelib.OracleKiwi(path)
    rnd = random.Random(17)
    texts = [misspell(t, rnd, True, continual == 1.0, lengthening < 1e9) for t in synthetic(sm, 50, 741, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 25, 742)] + EDGE_TEXTS[:20]
    got = _analyze_typo(dev, prod, texts, 2.5, top_n)
    corrected = 0
    for t, y in zip(texts, got):
        want = orc.analyze_typo(orc_t, t, 2.5, 0, top_n=top_n)
        assert _norm(want) == _norm(y), t
        corrected += any(x.typo_cost > 0 for x in want[0][0])
    assert corrected >= 5
    dev.close(); prod.close()


def test_emulated_kernels_on_a_quantised_knlm(emu_libs, small_quantised_model):
    """The device path on a model whose Knlm file is quantised / compressed (loader: kiwi_amd/csrc/model.cpp loadKnlm) against the oracle, which
    equals the real reference reading the same blob (tests/test_oracle_vs_ref.py::test_quantised_knlm_matches_reference)."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path, _ = small_quantised_model
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    _check(dev, orc, synthetic(sm, 60, 971, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 30, 972), top_ns=(1, 2))
    dev.close()


def _graph_bytes(dev, typo, text, dialect, norm_coda, use_device):
    import ctypes as C
    import numpy as np
    L = dev.lib
    L.kamd_typo_graph_device.restype = C.c_size_t
    L.kamd_typo_graph_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    u = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
    need = L.kamd_typo_graph_device(dev.h, typo.h, u.ctypes.data, len(u), dialect, int(norm_coda), int(use_device), None, 0)
    assert need, dev.last_error() if hasattr(dev, "last_error") else "kamd_typo_graph_device failed"
    buf = np.zeros(need, np.uint8)
    assert L.kamd_typo_graph_device(dev.h, typo.h, u.ctypes.data, len(u), dialect, int(norm_coda), int(use_device), buf.ctypes.data, need) == need
    return buf.tobytes()


def check_device_typo_graphs(lib, model_path, n_random=80):
    """Shared with tests/test_gpu_typo.py: the typo graphs of k_typo_graph (count pass + write pass; the graphs the analyze path uses) equal the
    host module's -- which tests/test_typo_product.py pins to the real reference byte for byte -- node for node, links, costs, continual
    indices, dialects, and the type / script of every node's last character, for this repo's own rule set (both directions, with and without
    continual typos, dialect masks) and every built-in set, on texts dense in the patterns, misspelt sentences, supplementary-plane
    characters and empty input."""
    import ctypes as C
    import random
    import test_typo_product
    from kiwi_amd.api import KiwiAmd
    from typo_cases import COND, INF, RULES, misspell, texts
    test_typo_product.LIB = lib
    dev = KiwiAmd(model_path, lib_path=lib)
    rnd = random.Random(11)
    corpus = texts(n_random, 77) + ["😀안돼 𠀀됬어 😀", "a😀", "\ud83d", "됬\ud83d어", " ", "가" * 70, "됬어요 " * 12]
    corpus += [misspell(t, rnd, True, True) for t in texts(30, 78)]
    n_nodes = n_pool = 0
    cases = []
    for inverse in (True, False):
        for cont in (INF, 1.0):
            prod = test_typo_product.ProductTypo(cont, 0.25 if cont != INF else INF)
            for origs, errs, cost, cond, dia in RULES:
                for o in origs:
                    for e in errs:
                        assert prod.add(o, e, cost, COND[cond], dia) == 0
            prod.prepare(inverse)
            cases.append((prod, (0, 8, 0xFFFF)))
    prod0 = test_typo_product.ProductTypo()
    prod0.lib.kamd_typo_default.restype = C.c_void_p
    prod0.lib.kamd_typo_default.argtypes = [C.c_int]
    for k in range(7):
        p = test_typo_product.ProductTypo(); p.close()
        p.h = prod0.lib.kamd_typo_default(k)
        assert p.h
        p.prepare(True)
        cases.append((p, (0xFFFF,) if k == 6 else (0,)))
    prod0.close()
    for prod, dialects in cases:
        for dia in dialects:
            for t in corpus:
                for nc in (True, False) if len(t) < 12 else (True,):
                    host = _graph_bytes(dev, prod, t, dia, nc, False)
                    assert _graph_bytes(dev, prod, t, dia, nc, True) == host, (dia, nc, t)
                    n_nodes += 1
        # ... and the hook's host side is kamd_typo_graph itself plus the last-character bytes
        t = corpus[1]
        base = prod.graph_bytes(t, dialects[0], True)
        assert _graph_bytes(dev, prod, t, dialects[0], True, False)[:len(base)] == base
        prod.close()
    assert n_nodes > 800
    dev.close()


@pytest.mark.parametrize("stride", ["64", "16"])
def test_emulated_typo_graph_kernel_matches_host_module(emu_libs, small_model, monkeypatch, stride):
    """stride 64 = one chunk per wave (the ordering and the output copy on all lanes), 16 = four chunks per wave (one lane each)."""
    monkeypatch.setenv("KAMD_TYPO_GRAPH_STRIDE", stride)
    check_device_typo_graphs(emu_libs[0], small_model[1], n_random=80 if stride == "64" else 30)


@pytest.mark.parametrize("lanes,top_n", [("16", 1), ("64", 2)])
def test_emulated_cong_kernel_on_a_file_as_the_reference_builder_writes_it(emu_libs, mid_cong_vl4_model, monkeypatch, lanes, top_n):
    """cong.mdl with variable-length 16-bit keys (keySize 3: LM ids >= 63488 take two trie steps in congStep), 4-bit grouped embeddings
    (requantised to int8 by the loader) and the global model's sections present: the emulated search kernel against the oracle, which
    tests/test_cong_oracle.py pins to the real reference on the same file."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = mid_cong_vl4_model
    force_lanes(monkeypatch, lanes)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    texts = synthetic(sm, 70, 915, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 30, 916) + EDGE_TEXTS[:20]
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    for s, y in zip(texts, got):
        assert _norm(orc.analyze(s, top_n=top_n)) == _norm(y), (lanes, top_n, s)
    dev.close()


@pytest.mark.parametrize("lanes,top_n,bias", [("16", 1, 0.0), ("64", 2, 2.5)])
def test_emulated_unknown_forms_scored_by_the_character_model(emu_libs, small_cong_chr_model, monkeypatch, lanes, top_n, bias):
    """Match::oovChrModel (row f4): k_unk_chr scores every node's unknown form with the character model (byte-keyed context trie, int8 dot
    product, output bias), the search reads those scores / the per-form table instead of the length rule -- against the oracle, which
    tests/test_chr_oracle.py pins to the real UnkFormScorer + CoNgramModel of the reference."""
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    sm, path = small_cong_chr_model
    force_lanes(monkeypatch, lanes)
    match = oraclelib.MATCH_ALL_WITH_NORMALIZING | (1 << 8)
    orc = oraclelib.OracleKiwi(path)
    orc.lib.korc_set_oov_chr_bias.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_float]
    orc.lib.korc_set_oov_chr_bias(orc.h, bias)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    dev.set_oov_chr_bias(bias)
    texts = synthetic(sm, 50, 917, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 25, 918) + EDGE_TEXTS[:40]
    got = dev.analyze_batch(texts, top_n=top_n, match=match).to_python()
    plain = dev.analyze_batch(texts, top_n=top_n).to_python()
    differ = 0
    for s, y, p in zip(texts, got, plain):
        assert _norm(orc.analyze(s, top_n=top_n, match=match)) == _norm(y), (lanes, top_n, s)
        differ += _norm(y) != _norm(p)
    assert differ > 10
    dev.close()


@pytest.mark.parametrize("lanes,top_n,mode,bias,params", [("16", 1, 2, 0.0, None), ("64", 2, 3, 2.5, (60.0, 1.5, 1.0))])
def test_emulated_unknown_forms_scored_with_substring_frequencies(emu_libs, small_cong_chr_model, monkeypatch, lanes, top_n, mode, bias, params):
    """Match::oovChrFreqModel / oovChrFreqBranchModel (row f4): k_unk_chr_freq counts the prefixes of every node's unknown form in the filtered text
    (chr_freq.hpp) and mixes them into the character model's score with tanhf / expf / logf restated from glibc -- against the oracle (its own
    substring table, libm), which tests/test_chr_oracle.py pins to the real UnkFormScorer + SubstringCounter of the reference.  Texts of several
    chunks count over the WHOLE text."""
    import ctypes as C
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from corpora import repeated_unknown_texts
    sm, path = small_cong_chr_model
    force_lanes(monkeypatch, lanes)
    match = oraclelib.MATCH_ALL_WITH_NORMALIZING | (mode << 8)
    orc = oraclelib.OracleKiwi(path)
    orc.lib.korc_set_oov_chr_bias.argtypes = [C.c_void_p, C.c_float]
    orc.lib.korc_set_oov_chr_bias(orc.h, bias)
    orc.lib.korc_set_oov_freq_params.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    dev.set_oov_chr_bias(bias)
    if params:
        orc.lib.korc_set_oov_freq_params(orc.h, *params); dev.set_oov_freq_params(*params)
    texts = repeated_unknown_texts(sm, 60, 931) + synthetic(sm, 15, 932, min_jamo=5, max_jamo=100) + dictionary_mix(sm, 15, 933) + EDGE_TEXTS[:40]
    texts += [". ".join(texts[:6]) + ".", "😀가나 😀가나 😀가나 ★다라★ ★다라★", "abc abc abc abcd abcd", "가" * 40 + " " + "가" * 40]
    got = dev.analyze_batch(texts, top_n=top_n, match=match).to_python()
    plain = dev.analyze_batch(texts, top_n=top_n, match=oraclelib.MATCH_ALL_WITH_NORMALIZING | (1 << 8)).to_python()
    differ = 0
    for s, y, p in zip(texts, got, plain):
        assert _norm(orc.analyze(s, top_n=top_n, match=match)) == _norm(y), (lanes, top_n, s)
        differ += _norm(y) != _norm(p)
    assert differ > 5
    dev.close()


def test_emulated_character_model_option_without_the_model_is_refused(emu_libs, small_cong_model):
    from kiwi_amd.api import KiwiAmd
    import oraclelib
    dev = KiwiAmd(small_cong_model[1], lib_path=emu_libs[0])
    with pytest.raises(Exception, match="character-level noun model is not loaded"):
        dev.analyze_batch(["가나다"], match=oraclelib.MATCH_ALL_WITH_NORMALIZING | (1 << 8))
    dev.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_emulated_typo_correction_with_the_character_model(emu_libs, small_cong_chr_model, mode):
    """Match::oovChrModel / oovChrFreqModel together with a typo transformer: k_unk_chr / k_unk_chr_freq over the nodes of the typo lattices, the typo + CoNgram search kernel reading its scores."""
    import random
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_cong_chr_model
    match = oraclelib.MATCH_ALL_WITH_NORMALIZING | (mode << 8)
    prod, orc_t = _typo_pair(emu_libs[0], 1.0)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    rnd = random.Random(19)
    texts = [misspell(t, rnd, True, True) for t in synthetic(sm, 40, 919, min_jamo=5, max_jamo=80) + dictionary_mix(sm, 20, 920)] + EDGE_TEXTS[:20]
    if mode == 2:
        from corpora import repeated_unknown_texts
        texts += [misspell(t, rnd, True, True) for t in repeated_unknown_texts(sm, 30, 923)]
    got = _analyze_typo(dev, prod, texts, 2.5, match=match)
    for t, y in zip(texts, got):
        assert _norm(orc.analyze_typo(orc_t, t, 2.5, 0, match=match)) == _norm(y), t
    dev.close(); prod.close()


@pytest.mark.parametrize("lanes", ["64", "16"])
def test_emulated_typo_correction_with_a_skipbigram_model(emu_libs, small_sbg_model, monkeypatch, lanes):
    """viterbi_kernel_sbg_typo.hip (history rings + node typo costs) against the oracle, which tests/test_typo_oracle.py compares with the real
    reference on this combination."""
    import random
    import oraclelib
    from kiwi_amd.api import KiwiAmd
    from typo_cases import misspell
    sm, path = small_sbg_model
    force_lanes(monkeypatch, lanes)
    prod, orc_t = _typo_pair(emu_libs[0], 1.0)
    orc = oraclelib.OracleKiwi(path)
    dev = KiwiAmd(path, lib_path=emu_libs[0])
    rnd = random.Random(23)
    texts = [misspell(t, rnd, True, True) for t in synthetic(sm, 24, 921, min_jamo=5, max_jamo=50) + dictionary_mix(sm, 12, 922)] + EDGE_TEXTS[:12]
    got = _analyze_typo(dev, prod, texts, 2.5)
    for t, y in zip(texts, got):
        assert _norm(orc.analyze_typo(orc_t, t, 2.5, 0)) == _norm(y), t
    dev.close(); prod.close()


def check_device_against_model_variant_goldens(lib, names=("htx-q8c", "cong", "cong-oov-chr"), limit=None):
    """Shared with tests/test_gpu_zz_fullmodel_typo.py: the device path against the committed outputs of the REAL reference for the round-2 model
    variants (tests/golden/model_variants_golden.json, tools/make_golden_models.py): tokens, positions, morpheme ids, fp32 scores."""
    import json
    from kiwi_amd.api import KiwiAmd
    from test_oracle_vs_ref import _golden_model
    sets = json.load(open(os.path.join(HERE, "golden", "model_variants_golden.json"), encoding="utf-8"))["sets"]
    for name in names:
        g = sets[name]
        items = g["items"][:limit] if limit else g["items"]
        dev = KiwiAmd(_golden_model(name), lib_path=lib)
        got = dev.analyze_batch([it["text"] for it in items], match=g["match"]).to_python()
        for it, y in zip(items, got):
            toks = [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.sense_id, t.morph_id] for t in y[0][0]]
            assert toks == it["tokens"], (name, it["text"])
            assert y[0][1] == it["score"], (name, it["text"])
        dev.close()


def test_emulated_device_matches_the_model_variant_goldens(emu_libs):
    check_device_against_model_variant_goldens(emu_libs[0], limit=70)
