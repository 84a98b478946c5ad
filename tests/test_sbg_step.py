"""csrc/sbg_eval.hpp: the SkipBigram LM step the search kernel of SkipBigram models uses (shared host/device source).
CPU: the host build of that code (kamd_debug_sbg_next, no device) against the CPU restatement -- which is itself pinned
to the real reference's SbgState::next by tests/test_oracle_vs_ref.py -- bit for bit over random word sequences."""
import ctypes as C
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_host_sbg_step_equals_restatement(small_sbg_model):
    import oraclelib
    sm, path = small_sbg_model
    lib = C.CDLL(os.path.join(ROOT, "kiwi_amd", "libkiwi_hip.so"))
    lib.kamd_debug_sbg_next.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
    lib.kamd_last_error.restype = C.c_char_p
    orc = oraclelib.OracleKiwi(path)
    rng = np.random.default_rng(11)
    vocab = sm.raw.vocab_size
    steps = scored = 0
    for _ in range(150):
        node, pos, hist = 0, 0, [0] * 8          # restatement's state
        ring = np.zeros(8, np.uint32)            # product's state (the Knlm node is the restatement's: that step is not under test)
        ppos = C.c_uint32(0)
        for _ in range(40):
            w = int(rng.integers(0, vocab)) if rng.random() < 0.7 else int(rng.integers(3, 60))
            knlm_ll, _ = orc.lm_progress(node, w)
            want_ll, node, pos, hist = orc.lm_next(node, pos, hist, w)
            got = C.c_float(0)
            rc = lib.kamd_debug_sbg_next(path.encode(), ring.ctypes.data, C.byref(ppos), w, knlm_ll, C.byref(got))
            assert rc == 0, lib.kamd_last_error()
            assert struct.pack("f", got.value) == struct.pack("f", want_ll), (w, knlm_ll, got.value, want_ll)
            assert list(map(int, ring)) == hist and ppos.value == pos
            steps += 1
            scored += struct.pack("f", want_ll) != struct.pack("f", knlm_ll)
    assert steps == 6000 and scored > 500      # the mixture actually changed the Knlm score in a good share of the steps
