import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
p, t, d = get_workload(sys.argv[1] if len(sys.argv) > 1 else "c2")
print('workload ready', flush=True)
e = KiwiAmd(p)
print('engine ready', flush=True)
e.lib.kamd_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
b = e.stage(t)
print('staged', flush=True)
print(e.run(b), flush=True)
a = np.zeros(16, np.uint64)
e.lib.kamd_debug_profile(a.ctypes.data, 1)
for _ in range(3):
    r = e.run(b)
e.lib.kamd_debug_profile(a.ctypes.data, 1)
names = ["node setup", "gather/other in evaluate", "scoring", "emission", "prune", "node bookkeeping", "finish", "chunk fetch"]
tot = float(a[:8].sum())
print(r)
for n, v in zip(names, a[:8]):
    print(f"{n:28s} {int(v):14d} {100.0 * float(v) / tot:6.1f}%")
