import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
