"""EXPERIMENT (developer aid): where the lattice build spends its time -- the kernel is told to leave after a phase through KAMD_LATTICE_STOP;
results of such runs are void, only lattice_ms is read.  k_lattice_wave (default): 1 staging + match digest, 2 + character-type pass, 3 + ops and groups,
4 + fixpoint, 5 + ranks / successor masks, 6 + connectivity sweep and prefix, 0 everything.  KAMD_LATTICE_WAVE=0 (k_build_lattice): 1 staging, 2 + replay, 3 + reachability."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys
sys.path.insert(0, %r)
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
model_path, texts, desc = get_workload(sys.argv[1])[:3]
eng = KiwiAmd(model_path, 0)
b = eng.stage(texts)
for _ in range(2): eng.run(b)
t = [eng.run(b) for _ in range(5)]
print(sys.argv[1], "stop", sys.argv[2], "lattice_ms %%.3f scan_ms %%.3f" %% (sum(x["lattice_ms"] for x in t) / 5, sum(x["scan_ms"] for x in t) / 5))
''' % ROOT
for wl in sys.argv[1:] or ["c2", "c3"]:
    for stop in (("1", "2", "3", "0") if os.environ.get("KAMD_LATTICE_WAVE") == "0" else ("1", "2", "3", "4", "5", "6", "0")):
        subprocess.run([sys.executable, "-c", CODE, wl, stop], env=dict(os.environ, KAMD_LATTICE_STOP=stop))
