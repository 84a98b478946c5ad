import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _locked_make():
    """Tests build the lane emulator / the oracle with `make` from inside fixtures; with several test processes (below) two of them would write the same object
    files.  One lock file serialises every `make` of the suite: the second caller finds the build up to date."""
    import fcntl
    import subprocess
    if getattr(subprocess, "_kamd_locked_make", False):
        return
    lock_path = os.path.join(ROOT, "_data", ".make.lock")
    os.makedirs(os.path.dirname(lock_path), exist_ok=True)

    def wrap(fn, retry_torchrun=False):
        def call(cmd, *a, **kw):
            if isinstance(cmd, (list, tuple)) and cmd and os.path.basename(str(cmd[0])) == "make":
                with open(lock_path, "w") as lk:
                    fcntl.flock(lk, fcntl.LOCK_EX)
                    return fn(cmd, *a, **kw)
            r = fn(cmd, *a, **kw)
            # (a multi-process gloo job next to seven busy test processes: a rank was seen to abort in its rendezvous once in three runs of the suite -- such a job
            # gets ONE more try; a real defect fails twice)
            if retry_torchrun and isinstance(cmd, (list, tuple)) and "torch.distributed.run" in [str(c) for c in cmd] and getattr(r, "returncode", 0) != 0:
                # (never silently: the first attempt's end goes to stderr -- pytest shows it with a failure, `-rA` / `-s` always -- and into _data/torchrun_retries.log)
                tail = (getattr(r, "stderr", None) or "")
                tail = tail[-1500:] if isinstance(tail, str) else tail[-1500:].decode("utf-8", "replace")
                note = f"[conftest] torch.distributed.run job failed (rc {r.returncode}) and is tried ONCE more: {' '.join(str(c) for c in cmd)[-300:]}\n{tail}\n"
                sys.stderr.write(note)
                try:
                    with open(os.path.join(ROOT, "_data", "torchrun_retries.log"), "a") as f:
                        f.write(note)
                except OSError:
                    pass
                r = fn(cmd, *a, **kw)
            return r
        return call
    subprocess.check_call = wrap(subprocess.check_call)
    subprocess.run = wrap(subprocess.run, retry_torchrun=True)
    subprocess._kamd_locked_make = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "xdist_group: tests of one file stay in one worker process (pytest-xdist's loadgroup; a no-op without it)")
    _locked_make()
    # The CPU suite (-m "not gpu": oracle vs reference vs goldens, the kernels on the lane emulator) is half an hour of single-core work; its files are independent, so
    # it runs them on several processes when pytest-xdist is there -- file by file (module-scoped fixtures, a file's torchrun tests stay in one process; the two long
    # emulator files are spread test by test, pytest_collection_modifyitems below).  KAMD_TEST_PROCS=1 turns it off; the GPU suite (-m gpu) always runs in one process: its tests share the device.
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return
    if (config.option.markexpr or "").strip() != "not gpu" or getattr(config.option, "numprocesses", None) or config.option.collectonly:
        return
    n = int(os.environ.get("KAMD_TEST_PROCS", "0")) or min(8, max(1, (os.cpu_count() or 2) - 1))
    if n > 1:
        config.option.numprocesses = n
        config.option.tx = ["popen"] * n
        config.option.dist = "loadgroup"


def pytest_collection_modifyitems(config, items):
    """(with the processes above) a file's tests stay together -- except the two long emulator files, whose tests are independent of each other (libraries built under
    the make lock, models written atomically, switches set through monkeypatch): they are spread over the processes one by one."""
    spread = {"test_hipemu.py", "test_cong_global.py"}
    for it in items:
        name = os.path.basename(str(it.fspath))
        if name not in spread:
            it.add_marker(pytest.mark.xdist_group(name=name))


@pytest.fixture(scope="session")
def small_model(tmp_path_factory):
    """Deterministic small synthetic model shared by all tests (kiwi_amd/synth.py, seed 'KIWI')."""
    from kiwi_amd.synth import SynthModel, SMALL_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "small.raw")
    sm = SynthModel(SMALL_SPEC)
    sm.raw.save(path)
    return sm, path


@pytest.fixture(scope="session")
def small_order4_model():
    """SMALL_SPEC's lexicon with an order-4 Knlm (kiwi_amd/synth.py SMALL_ORDER4_SPEC)."""
    from kiwi_amd.synth import SynthModel, SMALL_ORDER4_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "small-order4.raw")
    sm = SynthModel(SMALL_ORDER4_SPEC)
    sm.raw.save(path)
    return sm, path


@pytest.fixture(scope="session")
def small_sbg_model():
    """The small synthetic model plus SkipBigram tables (same lexicon and Knlm; kiwi_amd/synth.py SMALL_SBG_SPEC)."""
    from kiwi_amd.synth import SynthModel, SMALL_SBG_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "small-sbg.raw")
    sm = SynthModel(SMALL_SBG_SPEC)
    sm.raw.save(path)
    return sm, path


@pytest.fixture(scope="session", params=["q8c", "q5", "htx", "htx-q8c"])
def small_quantised_model(request):
    """The small synthetic model with its Knlm blob as the reference's builder writes it: 8-bit quantised with compressed node sizes (q8c), or
    5-bit quantised (q5: the generic fixed-length bit stream) -- kiwi_amd/synth.py quantize_knlm; htx: with a history transformer (tag
    histories: the oldest token of a trie path is stored as tag + vocab), what the reference's builder writes by default; htx-q8c: both."""
    from kiwi_amd.synth import SynthModel, SMALL_Q8_SPEC, SMALL_Q5_SPEC, SMALL_HTX_SPEC, SMALL_HTX_Q8_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    name, spec = {"q8c": ("small-q8", SMALL_Q8_SPEC), "q5": ("small-q5", SMALL_Q5_SPEC), "htx": ("small-htx", SMALL_HTX_SPEC), "htx-q8c": ("small-htx-q8", SMALL_HTX_Q8_SPEC)}[request.param]
    path = os.path.join(d, name + ".raw")
    sm = SynthModel(spec)
    sm.raw.save(path)
    return sm, path, name


@pytest.fixture(scope="session")
def small_cong_model():
    """The small synthetic model plus a local, 8-bit CoNgram model in the reference's cong.mdl layout (kiwi_amd/synth.py SMALL_CONG_SPEC)."""
    from kiwi_amd.synth import SynthModel, SMALL_CONG_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "small-cong.raw")
    sm = SynthModel(SMALL_CONG_SPEC)
    sm.raw.save(path)
    return sm, path


@pytest.fixture(scope="session")
def mid_cong_vl4_model():
    """A 66 000-word lexicon with a CoNgram-only model file as the reference's builder writes one for a large vocabulary (kiwi_amd/synth.py
    MID_CONG_VL4_SPEC): 4-bit grouped embeddings, variable-length 16-bit trie keys (LM ids beyond 63488), the global model's sections present."""
    from kiwi_amd.synth import SynthModel, MID_CONG_VL4_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "mid-cong-vl4.raw")
    sm = SynthModel(MID_CONG_VL4_SPEC)
    sm.raw.save(path)
    return sm, path


@pytest.fixture(scope="session")
def small_cong_chr_model():
    """SMALL_CONG_SPEC plus the character model of Match::oovChrModel (nounchr.mdl layout; kiwi_amd/synth.py SMALL_CONG_CHR_SPEC)."""
    from kiwi_amd.synth import SynthModel, SMALL_CONG_CHR_SPEC
    d = os.path.join(ROOT, "_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "small-cong-chr.raw")
    sm = SynthModel(SMALL_CONG_CHR_SPEC)
    sm.raw.save(path)
    return sm, path


@pytest.fixture(scope="session")
def oracle(small_model):
    import subprocess
    import oraclelib
    if not oraclelib.available():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return oraclelib.OracleKiwi(small_model[1])


@pytest.fixture(scope="session")
def reference(small_model):
    """The real reference translation units (oracle/_ref); skipped where the prebuilt library is absent."""
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref/libkiwi_ref.so not built (needs /root/reference)")
    return refbridge.RefKiwi(small_model[1])


@pytest.fixture(scope="session")
def engine(small_model):
    from kiwi_amd.api import KiwiAmd
    eng = KiwiAmd(small_model[1])
    yield eng
    eng.close()          # release streams / events / device blocks before the interpreter (and the HIP runtime) shut down
