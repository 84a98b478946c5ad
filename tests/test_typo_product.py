"""CPU (not gpu): the product's host-side typo module (kiwi_amd/csrc/typo.cpp through kamd_typo_*: rule container, prepare(), typo graph, the
shipped built-in sets) against the oracle's -- and so, transitively and where oracle/_ref is present directly, against the real reference --
byte for byte.  The analysis with a transformer is tests/test_gpu_typo.py."""
import ctypes as C
import os

import numpy as np
import pytest

from typo_cases import COND, INF, RULES, texts

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip.so")


def _u16(s):
    return np.frombuffer(s.encode("utf-16-le", errors="surrogatepass"), np.uint16)


class ProductTypo:
    def __init__(self, continual=INF, lengthening=INF):
        L = self.lib = C.CDLL(LIB)
        L.kamd_typo_new.restype = C.c_void_p
        L.kamd_typo_new.argtypes = [C.c_float, C.c_float]
        L.kamd_typo_close.argtypes = [C.c_void_p]
        for f in (L.kamd_typo_add, L.kamd_typo_add_entry):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_int]
        L.kamd_typo_set_costs.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.kamd_typo_scale.argtypes = [C.c_void_p, C.c_float]
        L.kamd_typo_prepare.argtypes = [C.c_void_p, C.c_int]
        L.kamd_typo_graph.restype = C.c_size_t
        L.kamd_typo_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        self.h = L.kamd_typo_new(continual, lengthening)
        assert self.h

    def add(self, orig, error, cost=1.0, cond=0, dialect=0):
        o, e = _u16(orig), _u16(error)
        return self.lib.kamd_typo_add(self.h, o.ctypes.data, len(o), e.ctypes.data, len(e), cost, cond, dialect)

    def add_entry(self, orig, error, cost, cond, dialect):
        o, e = _u16(orig), _u16(error)
        assert self.lib.kamd_typo_add_entry(self.h, o.ctypes.data, len(o), e.ctypes.data, len(e), cost, cond, dialect) == 0

    def prepare(self, inverse=True):
        assert self.lib.kamd_typo_prepare(self.h, int(inverse)) == 0

    def graph_bytes(self, text, dialect=0, norm_coda=True):
        u = _u16(text)
        need = self.lib.kamd_typo_graph(self.h, u.ctypes.data, len(u), dialect, int(norm_coda), None, 0)
        assert need
        buf = np.zeros(need, np.uint8)
        self.lib.kamd_typo_graph(self.h, u.ctypes.data, len(u), dialect, int(norm_coda), buf.ctypes.data, need)
        return buf.tobytes()

    def close(self):
        self.lib.kamd_typo_close(self.h)


def _fill(t, by_value=True):
    for origs, errs, cost, cond, dia in RULES:
        for o in origs:
            for e in errs:
                assert t.add(o, e, cost, COND[cond], dia) in (0, None)


@pytest.mark.parametrize("inverse", [True, False])
def test_product_typo_graphs_equal_oracle(inverse):
    import oraclelib
    for cont, leng in ((INF, INF), (1.0, 0.25)):
        prod = ProductTypo(cont, leng); _fill(prod); prod.prepare(inverse)
        orc = oraclelib.OracleTypo(cont, leng); _fill(orc); orc.prepare(inverse)
        for dia in (0, 8, 0xFFFF):
            for t in texts(120, 41 + inverse):
                assert prod.graph_bytes(t, dia) == orc.graph_bytes(t, dia, True), (inverse, cont, dia, t)
        prod.close()


def test_product_rejects_malformed_rules():
    prod = ProductTypo()
    assert prod.add("ᄀ", "가") == -1          # onset vs syllable (TypoTransformer::addTypoNormalized throws)
    assert prod.add("ㅐ", "가") == -1          # vowel vs syllable
    assert prod.add("가", "나", 1.0, COND["vocalic"]) == -1      # left condition a rule cannot carry
    assert prod.lib.kamd_typo_scale(prod.h, -1.0) == -1
    prod.close()


@pytest.mark.parametrize("name", ["basic", "basic_with_continual_and_lengthening", "dialect"])
def test_product_builtin_sets_equal_reference(name):
    """The reference's built-in sets replayed entry by entry (update() order) into the product's container: same iteration order of
    the rule map, same replacement order, same graphs as the REAL reference."""
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    from test_typo_oracle import _ref_bytes
    ents, cont, leng = refbridge.default_typo_entries(name)
    ref = refbridge.RefTypo(); ref.update_default(name); ref.prepare(True)
    prod = ProductTypo()
    for o, e, cost, cond, dia in ents:
        prod.add_entry(o, e, cost, cond, dia)
    prod.lib.kamd_typo_set_costs(prod.h, cont, leng)
    prod.prepare(True)
    dia = 0xFFFF if name == "dialect" else 0
    for t in texts(100, 53):
        assert _ref_bytes(ref, t, dia) == prod.graph_bytes(t, dia), (name, t)
    prod.close()


@pytest.mark.parametrize("name", ["without", "basic", "continual", "basic_with_continual", "lengthening", "basic_with_continual_and_lengthening", "dialect"])
def test_shipped_builtin_sets_equal_reference(name):
    """kamd_typo_default / kiwi_typo_get_default: the sets assembled by the product from its own copy of the rule tables
    (kiwi_amd/csrc/typo_sets.inc + typo.cpp defaultTypoSet) give the graphs of the reference's getDefaultTypoSet objects themselves."""
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref not built")
    from test_typo_oracle import _ref_bytes
    ref = refbridge.RefTypo.from_default(name); ref.prepare(True)
    prod = ProductTypo()
    prod.close()
    prod.lib.kamd_typo_default.restype = C.c_void_p
    prod.lib.kamd_typo_default.argtypes = [C.c_int]
    prod.h = prod.lib.kamd_typo_default(refbridge.DEFAULT_TYPO_SETS[name])
    assert prod.h
    prod.prepare(True)
    for dia in ((0, 0xFFFF, 8) if name == "dialect" else (0,)):
        for t in texts(100, 57):
            assert _ref_bytes(ref, t, dia) == prod.graph_bytes(t, dia), (name, dia, t)
    prod.close()
    assert not prod.lib.kamd_typo_default(7)
