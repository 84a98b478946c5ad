"""Developer aid (round 5): per-chunk timeline of the SkipBigram search on the first N sentences of c3-sbg (top-3) -- the slowest chunks phase by phase.
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_timeline.so python tools/sbg_timeline.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
p, t, d = get_workload("c3-sbg")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
e = KiwiAmd(p)
e.analyze_batch(t[:2048], top_n=3).close()
b = e.stage(t[:n])
e.fetch(b, 3).close()
os.environ["KAMD_TIMELINE_PRINT"] = "1"
t0 = time.perf_counter()
print(e.run(b), "wall %.1f ms" % (1e3 * (time.perf_counter() - t0)))
