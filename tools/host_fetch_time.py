"""Developer aid: host time of kamd_fetch (D2H + result assembly) per sentence on ONE core, measured on a batch whose kernels ran once (lane emulator: LIB=tests/hipemu/_build/libkiwi_hipemu.so,
or the product library on a GPU box).  KAMD_HOST_THREADS=1 python tools/host_fetch_time.py <sentences> <repetitions>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
p,t,d = get_workload('c2-64k')
n = int(sys.argv[1]); reps = int(sys.argv[2])
dev = KiwiAmd(p, lib_path=os.environ.get("LIB", '/root/repo/tests/hipemu/_build/libkiwi_hipemu.so'))
b = dev.stage(t[:n])
dev.fetch(b, 1).close()
t0=time.perf_counter()
for _ in range(reps): dev.fetch(b, 1).close()
el=time.perf_counter()-t0
print("fetch+close per sentence: %.2f us (%d sentences x %d)" % (el/reps/n*1e6, n, reps))
