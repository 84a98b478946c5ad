#!/usr/bin/env python3
"""Developer aid: are the gfx950 kernels of two builds of libkiwi_hip.so the same machine code?

    python tools/isa_diff.py OLD.so NEW.so [--show NAME]

Extracts the gfx950 code object from each library's offload bundle, disassembles it (llvm-objdump) and compares every
function instruction by instruction.  pc-relative address literals (the s_add_u32 / s_addc_u32 after s_getpc_b64) and the
padding after a function's last instruction are ignored: they move with the layout, not with the code.

Why: the search kernel is sensitive enough to register allocation that "a harmless refactor" can change its speed or, with
this compiler, its behaviour (findings); there is not always a GPU at hand to re-measure.  A change that is meant
to leave a measured kernel alone can be PROVEN to do so here on the build box -- this is how the SkipBigram kernel was added
as a second translation unit without touching the Knlm kernels (k_best_path<*, *> identical before and after).
"""
import collections
import difflib
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        o = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + 1)
    return out


def functions(path):
    funcs = collections.OrderedDict()
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            text = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line.strip())
            if m:
                cur = m.group(1)
                funcs[cur] = []
                continue
            m = re.match(r"^\s*([a-z_0-9]+.*?)\s*//\s*[0-9A-F]+:", line)
            if cur is not None and m:
                ins = re.sub(r"\s+", " ", m.group(1))
                ins = re.sub(r"^(s_addc?_u32 s\d+, s\d+, )0x[0-9a-f]{5,}$", r"\1<pcrel>", ins)
                funcs[cur].append(ins)
    for k, v in funcs.items():       # drop what follows the last return / end of program (alignment padding decoded as code)
        last = max((i for i, ins in enumerate(v) if ins.startswith(("s_endpgm", "s_setpc_b64", "s_swappc_b64"))), default=len(v) - 1)
        funcs[k] = v[:last + 1]
    return funcs


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    show = sys.argv[sys.argv.index("--show") + 1] if "--show" in sys.argv else None
    names = demangle(list(a) + [k for k in b if k not in a])
    changed = 0
    for k in a:
        short = re.sub(r"\(.*", "", names[k])[:100]
        if k not in b:
            print(f"{'only in OLD':28s} {short}")
            changed += 1
        elif a[k] == b[k]:
            print(f"{'identical':28s} {short}  ({len(a[k])} instructions)")
        else:
            r = difflib.SequenceMatcher(None, a[k], b[k], autojunk=False).ratio()
            print(f"{'DIFFERENT (%.3f)' % r:28s} {short}  ({len(a[k])} -> {len(b[k])} instructions)")
            changed += 1
            if show and show in names[k]:
                for line in list(difflib.unified_diff(a[k], b[k], lineterm="", n=1))[:200]:
                    print("    " + line)
    for k in b:
        if k not in a:
            print(f"{'only in NEW':28s} {re.sub(r'[(].*', '', names[k])[:100]}  ({len(b[k])} instructions)")
    sys.exit(1 if changed else 0)


if __name__ == "__main__":
    main()
