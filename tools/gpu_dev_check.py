"""Developer check run on the GPU box: HIP path vs the CPU oracle on a synthetic model."""
import os
import sys
import time
from dataclasses import astuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kiwi_amd.synth import SynthModel, SMALL_SPEC  # noqa: E402
from kiwi_amd.api import KiwiAmd  # noqa: E402
import oraclelib  # noqa: E402

os.makedirs(os.path.join(ROOT, "_data"), exist_ok=True)
path = os.path.join(ROOT, "_data", "small.raw")
sm = SynthModel(SMALL_SPEC)
sm.raw.save(path)
o = oraclelib.OracleKiwi(path)
k = KiwiAmd(path)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
corpus = sm.make_corpus(n, 77, min_jamo=5, max_jamo=120)
bad = 0
for s in corpus[:100]:
    x = o.split(s)
    y = k.split(s)
    if x != y:
        bad += 1
        if bad < 3:
            print("SPLIT DIFF", s)
            for cx, cy in zip(x, y):
                print(cx[0], cy[0], len(cx[1]), len(cy[1]))
                for i, (a, b) in enumerate(zip(cx[1], cy[1])):
                    if a != b:
                        print(i, a, b)
                        break
print("split bad", bad, "/ 100", flush=True)
t0 = time.time()
res = k.analyze_batch(corpus).to_python()
print("gpu batch sec", time.time() - t0, flush=True)
bada = 0
for s, y in zip(corpus, res):
    x = o.analyze(s)
    xs = [([astuple(t) for t in a[0]], a[1]) for a in x]
    ys = [([astuple(t) for t in a[0]], a[1]) for a in y]
    if xs != ys:
        bada += 1
        if bada < 4:
            print("ANALYZE DIFF", s)
            print(x[0][1], y[0][1], len(x[0][0]), len(y[0][0]))
            for tx, ty in zip(x[0][0], y[0][0]):
                if astuple(tx) != astuple(ty):
                    print(tx, "\n", ty)
                    break
print("analyze bad", bada, "/", len(corpus), flush=True)
b = k.stage(corpus)
print(b.info())
for _ in range(3):
    print(k.run(b))
