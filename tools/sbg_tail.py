import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
import oraclelib
from kiwi_amd.workloads import get_workload
path, texts, _ = get_workload("c3-sbg")
orc = oraclelib.OracleKiwi(path)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rows = []
t0 = time.time()
for t in texts[:n]:
    orc.counters(reset=True)
    a = time.perf_counter()
    orc.analyze(t, top_n=3)
    dt = time.perf_counter() - a
    c = orc.counters()
    rows.append((len(t), dt, oraclelib.alg_bytes(c)["search"]))
print("elapsed", time.time() - t0)
r = np.array(rows)
for name, col in (("oracle seconds", 1), ("algorithmic search bytes", 2)):
    v = np.sort(r[:, col])[::-1]
    tot = v.sum()
    print(name, "total %.3g  mean %.3g  median %.3g  p99 %.3g  max %.3g  max/mean %.1f  top 1%% share %.2f  top 0.1%% share %.2f" % (tot, v.mean(), np.median(v), v[len(v) // 100], v[0], v[0] / v.mean(), v[:max(1, len(v) // 100)].sum() / tot, v[:max(1, len(v) // 1000)].sum() / tot))
heavy = np.argsort(r[:, 2])[::-1][:5]
for i in heavy: print("heavy", i, "chars", int(r[i, 0]), "seconds %.3f" % r[i, 1], "bytes %.3g" % r[i, 2])
