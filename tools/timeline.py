"""Developer aid: per-chunk / per-phase timeline of the search kernel.  Build `make -C kiwi_amd/csrc timeline`, run with
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_timeline.so python tools/timeline.py [workload]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
p, t, d = get_workload(sys.argv[1] if len(sys.argv) > 1 else "c2")
e = KiwiAmd(p); b = e.stage(t)
e.run(b)
os.environ["KAMD_TIMELINE_PRINT"] = "1"
print(e.run(b))
