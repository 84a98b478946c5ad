#!/usr/bin/env python3
"""Developer aid: what is in a Kiwi model directory, and which of it kiwi_init of this library reads.

    python tools/check_model_dir.py /path/to/models/cong/base

Prints the header fields of sj.knlm / skipbigram.mdl / cong.mdl / nounchr.mdl (layouts: include/kiwi/Knlm.h:9-15, src/SkipBigramModel.hpp:40-105,
include/kiwi/CoNgramModel.h:18-34) and says for each file whether kiwi_amd/csrc/model.cpp loads that variant.  The text files of a real model
directory (combiningRule.txt, default.dict, multi.dict, typo.dict, dialect.dict) are what the reference's KiwiBuilder consumes at build time
(src/KiwiBuilder.cpp:1035-1092, 2385-2640: rule-combined morphemes, dictionary entries); this library does not read them -- kiwi_init refuses a
directory that holds combiningRule.txt unless KAMD_ALLOW_UNEXPANDED_MODEL=1, because its analyses would silently differ from the reference's."""
import os
import struct
import sys


def main(d):
    def blob(name):
        p = os.path.join(d, name)
        return open(p, "rb").read() if os.path.exists(p) else None
    m = blob("sj.morph")
    print("sj.morph      :", "missing (required)" if m is None else f"{len(m)} bytes, key {m[:4]!r} ({'ok' if m[:4] == b'KIWI' else 'NOT the serializer key KIWI'})")
    k = blob("sj.knlm")
    if k is not None:
        (num_nodes, node_off, key_off, ll_off, gamma_off, qtable_off, htx_off, unk_id, bos_id, eos_id, vocab, order, key_size, diff_size, quantized, extra) = struct.unpack_from("<11Q4BI", k, 0)
        ok = key_size in (2, 4) and (quantized & 0x1F) <= 16 and not ((quantized & 0x80) and key_size != 2)
        print(f"sj.knlm       : {len(k)} bytes, nodes {num_nodes}, vocab {vocab}, order {order}, key bytes {key_size}, quantised bits {quantized & 0x1F}, "
              f"compressed node sizes {bool(quantized & 0x80)}, history transformer {bool(htx_off)} -> {'loaded' if ok else 'NOT supported'}")
    s = blob("skipbigram.mdl")
    if s is not None:
        vocab, key_size, window, compressed, quantize = struct.unpack_from("<Q4B", s, 0)
        ok = not compressed and not quantize and key_size in (2, 4) and window == 8
        print(f"skipbigram.mdl: {len(s)} bytes, vocab {vocab}, key bytes {key_size}, window {window}, compressed {compressed}, quantised {quantize} -> {'loaded' if ok else 'NOT supported (only the uncompressed, unquantised layout is)'}")
    for name in ("cong.mdl", "nounchr.mdl"):
        c = blob(name)
        if c is None:
            continue
        vocab, nctx, dim, flags, key_size, window, qbit, qgroup, num_nodes = struct.unpack_from("<QQHHBBBBQ", c, 0)
        if name == "cong.mdl":
            ok = flags == 0 and key_size in (2, 3, 4) and (qbit == 8 or (qbit == 4 and qgroup in (4, 8, 16) and dim % 16 == 0)) and dim % 4 == 0
            note = "scored locally (ModelType::cong); the global variant (window 7) is not built" if window else "local model"
        else:
            ok = key_size == 1 and qbit == 8 and window == 0
            note = "used by Match::oovChrModel next to cong.mdl"
        print(f"{name:14}: {len(c)} bytes, vocab {vocab}, contexts {nctx}, dim {dim}, flags {flags}, keySize {key_size}, window {window}, qbit {qbit}, qgroup {qgroup}, "
              f"trie nodes {num_nodes} -> {'loaded' if ok else 'NOT supported'} ({note})")
    texts = [f for f in ("combiningRule.txt", "default.dict", "multi.dict", "typo.dict", "dialect.dict", "extract.mdl") if os.path.exists(os.path.join(d, f))]
    if texts:
        print("build-time inputs of the reference's KiwiBuilder that this library does NOT consume:", ", ".join(texts))
        if "combiningRule.txt" in texts:
            print("  -> kiwi_init refuses this directory: export it with the reference's builder (tools/export_built.cpp, built against libkiwi) and load the "
                  "container that writes;")
            print("     (or set KAMD_ALLOW_UNEXPANDED_MODEL=1 to analyse with sj.morph + the language model alone: "
                  "rule-combined morphemes and dictionary entries will be missing, results differ from the reference's)")


if __name__ == "__main__":
    if len(sys.argv) != 2:
        sys.exit(__doc__)
    main(sys.argv[1])
