#!/usr/bin/env python3
"""Writes the fixtures of tests/test_built_model.py: a model built by the REAL KiwiBuilder (src/KiwiBuilder.cpp, compiled unmodified into
oracle/_ref/libkiwi_ref_x86.so) from a directory as Kiwi ships it --

    sj.morph + sj.knlm   the small synthetic model + the gold lexicon of eval_data + the five morphemes the dictionary files refer to
                         (kiwi_amd.workloads.eval_model(for_builder=True); the shipped binaries are git-LFS pointers here)
    extract.mdl          empty tables, written through the reference's serializer (the word detector is not on the analysis path)
    combiningRule.txt, default.dict, typo.dict     the REAL files of /root/reference/models/cong/base (113 k dictionary entries, the combining rules)

-- exported after the builder's own buildCombinedMorphemes step as a raw-model container (tools/export_built.cpp, the maintainer's exporter) into
tests/golden/eval_built_model.raw.xz, and what the built Kiwi itself (KiwiBuilder::build) answers on column 1 of the eval_data files into
tests/golden/eval_built_<file>.json (typo files with the built-in set basicTypoSetWithContinual).  Run in the build container."""
import ctypes as C, json, lzma, os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
SHIPPED = "/root/reference/models/cong/base"
EVAL = "/root/reference/eval_data"
FILES = ("web", "written", "web_with_typos", "web_with_cont_typos")
GOLD = os.path.join(ROOT, "tests", "golden")
OPTIONS = 1 | 2 | 4          # BuildOption: integrateAllomorph | loadDefaultDict | loadTypoDict (the checkout has no multi.dict)
MODEL_TYPE = 2               # ModelType::knlm


def shipped_dir(raw_path):
    """A model directory as Kiwi ships it, around the synthetic sj.morph / sj.knlm of `raw_path`."""
    import refbridge
    lib = C.CDLL(refbridge.LIB_X86_PATH)
    d = tempfile.mkdtemp()
    lib.kref_write_model_dir.argtypes = [C.c_char_p, C.c_char_p]
    lib.kref_write_empty_extract.argtypes = [C.c_char_p]
    assert lib.kref_write_model_dir(raw_path.encode(), d.encode()) == 0
    assert lib.kref_write_empty_extract(d.encode()) == 0
    for f in ("combiningRule.txt", "default.dict", "typo.dict"):
        shutil.copy(os.path.join(SHIPPED, f), d)
    return lib, d


def build_exporter():
    """tools/export_built.cpp -- the exporter a Kiwi maintainer builds against libkiwi -- linked here against the reference objects of oracle/_ref
    (every object but the bridge's).  Returns the path of the program."""
    import glob, subprocess
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "oracle", "_ref", "obj_x86", "*.o"))) if not o.endswith("ref_bridge.o")]
    assert objs, "run `make -C oracle refx86` first"
    exe = os.path.join(ROOT, "tools", "_build", "export_built")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    src = os.path.join(ROOT, "tools", "export_built.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(p) for p in [src] + objs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-w", "-DKIWI_ARCH_X86_64", "-I" + os.path.join(ROOT, "oracle", "standin"), "-I/root/reference/include",
                               "-I/root/reference/src", "-I" + ROOT, src] + objs + ["-pthread", "-o", exe])
    return exe


def export(d, out):
    """export_built <directory> <out> <model type> <options> -> the program's report line."""
    import subprocess
    return subprocess.run([build_exporter(), d, out, str(MODEL_TYPE), str(OPTIONS)], check=True, capture_output=True, text=True).stdout.strip()


def main():
    import refbridge
    from kiwi_amd.workloads import eval_model
    raw, _ = eval_model(for_builder=True)
    lib, d = shipped_dir(raw)
    try:
        out = os.path.join(tempfile.gettempdir(), "eval_built_model.raw")
        print(export(d, out))
        with open(out, "rb") as f, lzma.open(os.path.join(GOLD, "eval_built_model.raw.xz"), "wb", preset=9) as g:
            g.write(f.read())
        print(os.path.getsize(out), "bytes ->", os.path.getsize(os.path.join(GOLD, "eval_built_model.raw.xz")), "compressed")
        built = refbridge.RefKiwi.built(d, MODEL_TYPE, OPTIONS)
        typo = refbridge.RefTypo.from_default("basic_with_continual")
        typo.prepare(True)
        for fn in FILES:
            with_typo = "typos" in fn
            items = []
            for line in open(os.path.join(EVAL, fn + ".txt"), encoding="utf-8"):
                text = line.rstrip("\n").split("\t")[0]
                if not text:
                    continue
                res = built.analyze_typo(typo, text, 2.5, 0) if with_typo else built.analyze(text)
                toks, score = res[0]
                items.append({"text": text, "score": score,
                              "tokens": [[t.form, t.tag, t.position, t.length, t.word_position, t.sent_position, t.line_number, t.score, t.typo_cost] for t in toks]})
            json.dump({"source": f"Kiwi built by the real KiwiBuilder (oracle/_ref x86) on eval_data/{fn}.txt column 1" + (", typo set basicTypoSetWithContinual, threshold 2.5" if with_typo else ""),
                       "typo": with_typo, "items": items}, open(os.path.join(GOLD, f"eval_built_{fn}.json"), "w", encoding="utf-8"), ensure_ascii=False, separators=(",", ":"))
            print(fn, len(items), "lines,", sum(len(i["tokens"]) for i in items), "tokens")
    finally:
        shutil.rmtree(d)


if __name__ == "__main__":
    main()
