#!/usr/bin/env python3
"""Developer aid: register / scratch / LDS figures and instruction counts of the gfx950 kernels in a build of libkiwi_hip.so
(from the code object's metadata notes and disassembly).   python tools/kernel_regs.py [LIB] [name filter]"""
import re, subprocess, sys, tempfile
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_diff import code_objects, functions
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
lib = sys.argv[1] if len(sys.argv) > 1 else "kiwi_amd/libkiwi_hip.so"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = functions(lib)
for co in code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    for blk in text.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dem: continue
        ins = funcs.get(name, [])
        n_s = sum(1 for i in ins if i.startswith("s_")); n_v = sum(1 for i in ins if i.startswith("v_")); n_m = len(ins) - n_s - n_v
        print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>5} static-insts {len(ins)} (s {n_s} v {n_v} mem/other {n_m})")
