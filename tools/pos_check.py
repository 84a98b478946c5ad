"""Developer check: the first N sentences of a workload through the HIP path (library from $KAMD_LIB) vs the CPU oracle; KAMD_POS_STATS=1 prints what
the position-step kernel handed over to the general one."""
import os, sys, time
from dataclasses import astuple
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kiwi_amd.workloads import get_workload, workload_top_n
from kiwi_amd.api import KiwiAmd
import oraclelib
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
path, texts, desc = get_workload(name)
texts = texts[:n]
o = oraclelib.OracleKiwi(path); k = KiwiAmd(path)
t = time.time()
res = k.analyze_batch(texts, top_n=workload_top_n(name)).to_python()
print("device side %.2f s" % (time.time() - t), flush=True)
bad = 0
for s, y in zip(texts, res):
    x = o.analyze(s, top_n=workload_top_n(name))
    if [([astuple(t) for t in a[0]], a[1]) for a in x] != [([astuple(t) for t in a[0]], a[1]) for a in y]: bad += 1
print("RESULT", name, "bad", bad, "/", len(texts), flush=True)
