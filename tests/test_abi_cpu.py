"""CPU: the product shared library loads, exports every symbol include/*.h declares, and refuses to run without
a GPU instead of falling back to any CPU path.  No compute calls here."""
import ctypes
import os

import pytest

from kiwi_amd import api


def _lib():
    if not os.path.exists(api.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(api.LIB_PATH)


@pytest.mark.parametrize("header", ["kiwi_amd.h", "kiwi_capi.h"])
def test_every_declared_symbol_is_exported(header):
    lib = _lib()
    names = api.declared_symbols(header)
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_open_fails_loudly_without_gpu(small_model):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError) as e:
        api.KiwiAmd(small_model[1])
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_product_sources_do_not_reference_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "kiwi_amd")):
        if "_obj" in dp or "__pycache__" in dp:
            continue
        for f in files:
            if f.endswith((".so", ".pyc", ".o")):
                continue
            text = open(os.path.join(dp, f), encoding="utf-8", errors="ignore").read()
            assert "oracle/" not in text.replace("oracle/ref_bridge.cpp", "").replace("oracle/_ref", "").replace("oracle's korc_split", "") or f in ("synth.py",), (dp, f)
