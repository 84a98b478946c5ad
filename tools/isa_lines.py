#!/usr/bin/env python3
"""Developer aid: static instruction counts per source line of one kernel in the line-table build (make -C kiwi_amd/csrc lines):
where the machine code of a kernel comes from.   python tools/isa_lines.py kiwi_amd/libkiwi_hip_lines.so "k_pos_path<16, 3>" [file filter]
Prints, per source file, the lines ordered by line number with VALU / SALU / other counts, and sums per `// ----` section header if present."""
import collections, re, subprocess, sys, tempfile, os
sys.path.insert(0, os.path.dirname(__file__))
from isa_diff import code_objects
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
lib, kern = sys.argv[1], sys.argv[2]
for co in code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        text = subprocess.run([OBJDUMP, "-d", "-l", "--no-show-raw-insn", "-C", f.name], capture_output=True, text=True).stdout
    cur_fn, cur_line = None, None
    cnt = collections.defaultdict(lambda: [0, 0, 0])
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m: cur_fn = m.group(1); continue
        if cur_fn is None or kern not in cur_fn or cur_fn.endswith(".kd"): continue
        m = re.match(r"^; (\S+):(\d+)$", ln)
        if m: cur_line = (os.path.basename(m.group(1)), int(m.group(2))); continue
        ins = ln.strip().split()
        if not ins or ins[0].startswith(";"): continue
        op = ins[0]
        k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2
        cnt[cur_line][k] += 1
    if not cnt: continue
    tot = [sum(v[i] for v in cnt.values()) for i in range(3)]
    print("total valu %d salu %d other %d" % tuple(tot))
    for key in sorted(cnt, key=lambda k: (k is None, k)):
        v = cnt[key]
        print("%s:%s\t%d\t%d\t%d" % (key[0] if key else "?", key[1] if key else 0, v[0], v[1], v[2]))
