"""CPU: the pattern recognisers of the host text preparation (kiwi_amd/csrc/textprep.cpp: URL, e-mail, mention, hashtag, number, serial, abbreviation,
emoji -- the languages of the reference's src/PatternMatcher.cpp) against the REAL reference: committed golden vectors (always), and a larger seeded
fuzz run against the compiled reference where oracle/_ref is built."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from pattern_cases import pattern_cases

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "kiwi_amd", "libkiwi_hip.so")


@pytest.fixture(scope="module")
def probe():
    lib = C.CDLL(LIB)
    lib.kamd_debug_match_pattern.restype = C.c_uint64
    lib.kamd_debug_match_pattern.argtypes = [C.c_uint16, C.c_void_p, C.c_uint32, C.c_uint64]

    def f(left, units, match):
        u = np.asarray(units, np.uint16)
        r = int(lib.kamd_debug_match_pattern(left, u.ctypes.data, len(u), match))
        return r & 0xFFFFFFFF, r >> 32
    return f


def test_recognisers_match_golden_vectors_of_the_reference(probe):
    g = json.load(open(os.path.join(HERE, "golden", "pattern_golden.json"), encoding="utf-8"))
    assert len(g["items"]) >= 3000 and sum(1 for it in g["items"] if it["len"]) > 1000
    for it in g["items"]:
        assert probe(it["left"], it["units"], it["match"]) == (it["len"], it["tag"]), it


def test_recognisers_match_the_compiled_reference_on_seeded_fuzz(probe):
    import refbridge
    if not refbridge.available():
        pytest.skip("oracle/_ref/libkiwi_ref.so not built")
    hits = 0
    for left, text, match in pattern_cases(40000, 1234):
        units = np.frombuffer(text.encode("utf-16-le", errors="surrogatepass"), np.uint16)
        for k in range(min(3, len(units))):      # (also from inside the text: the unit before is then a real left context)
            lc = ord(left) if k == 0 else int(units[k - 1])
            want = refbridge.match_pattern(chr(lc) if lc < 0xD800 or lc > 0xDFFF else left, text[:0] + units[k:].tobytes().decode("utf-16-le", errors="surrogatepass"), match)
            if lc >= 0xD800 and lc <= 0xDFFF:
                lc = ord(left)
            got = probe(lc, units[k:], match)
            assert got == want, (left, text, k, match)
            hits += got[0] > 0
    assert hits > 10000
