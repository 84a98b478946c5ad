import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
from dataclasses import astuple
import oraclelib
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
def _norm(res): return [([astuple(t) for t in a[0]], a[1]) for a in res]
path, texts, _ = get_workload("c3-sbg")
orc = oraclelib.OracleKiwi(path)
dev = KiwiAmd(path, lib_path=sys.argv[1])
for i in (1505, 1188, 5, 100, 200, 7):
    t0 = time.perf_counter()
    got = dev.analyze_batch([texts[i]], top_n=3).to_python()[0]
    dt = time.perf_counter() - t0
    ok = _norm(orc.analyze(texts[i], top_n=3)) == _norm(got)
    print(i, "emulated %.2f s" % dt, "== oracle" if ok else "DIFFERS")
dev.close()
