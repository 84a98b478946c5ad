"""Developer aid: end-to-end timing of one batch through the C ABI -- stage (host text preparation + H2D), run (kernels),
fetch (D2H + host post-processing into token records) -- i.e. the PCIe- and host-inclusive rate next to bench.py's value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import get_workload
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
p, t, d = get_workload(name)
e = KiwiAmd(p)
b = e.stage(t); e.run(b); e.fetch(b)          # warm
for threads in (0, 1):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); b = e.stage(t, host_threads=threads)
        t1 = time.perf_counter(); e.run(b)
        t2 = time.perf_counter(); r = e.fetch(b)
        t3 = time.perf_counter(); ts.append((t1 - t0, t2 - t1, t3 - t2))
    s, r_, f = min(ts, key=sum)
    print(f"{name} host_threads={threads}: stage {s*1e3:.1f} ms, run {r_*1e3:.1f} ms, fetch {f*1e3:.1f} ms -> {len(t)/(s+r_+f):.0f} sentences/s end to end")
