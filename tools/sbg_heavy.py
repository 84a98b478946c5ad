import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import oraclelib
from kiwi_amd.workloads import get_workload
path, texts, _ = get_workload("c3-sbg")
orc = oraclelib.OracleKiwi(path)
for i in (1505, 1188, 5, 100, 200):
    orc.counters(reset=True)
    a = time.perf_counter(); orc.analyze(texts[i], top_n=3); dt = time.perf_counter() - a
    c = orc.counters()
    print(i, "chars", len(texts[i]), "sec %.3f" % dt, {k: v for k, v in c.items() if v and k not in ("inputUnits",)})
