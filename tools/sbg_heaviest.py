"""Which sentences of the c3-sbg corpus are the heaviest for the SkipBigram search -- most lattice nodes with more than 512 incoming paths (the oracle's
event counters), then most incoming paths of one node.  Writes tests/golden/c3_sbg_heaviest.json (indices into the corpus, read by tests/test_gpu_fullmodel.py).
  python tools/sbg_heaviest.py [threads]"""
import json, os, sys, threading, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oraclelib
from kiwi_amd.workloads import get_workload
path, texts, _ = get_workload("c3-sbg")
local = threading.local()
def one(i):
    if not hasattr(local, "k"):
        local.k = oraclelib.OracleKiwi(path)
    o = local.k
    o.counters(reset=True)
    o.analyze(texts[i], top_n=3)
    c = o.counters()
    return (i, int(c["nodesOver512"]), int(c["maxPrevPaths"]), int(c["transitions"]))
t0 = time.time()
with ThreadPoolExecutor(int(sys.argv[1]) if len(sys.argv) > 1 else 8) as ex:
    rows = list(ex.map(one, range(len(texts))))
rows.sort(key=lambda r: (-r[1], -r[2], r[0]))
out = {"corpus": "c3-sbg", "sentences": len(texts), "ranked_by": "nodesOver512, then maxPrevPaths (oracle event counters, top-3)", "heaviest": [{"index": r[0], "nodesOver512": r[1], "maxPrevPaths": r[2], "transitions": r[3]} for r in rows[:64]]}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "c3_sbg_heaviest.json"), "w"), indent=0)
print("elapsed %.0f s" % (time.time() - t0), rows[:5])
