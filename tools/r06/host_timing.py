"""Developer aid: where one end-to-end batch spends its host time on the GPU box (KAMD_HOST_TIMING laps of stage / fetch / release) for several
part counts.   python tools/r06/host_timing.py [workload]"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kiwi_amd.api import KiwiAmd, pack_texts
from kiwi_amd.workloads import get_workload
name = sys.argv[1] if len(sys.argv) > 1 else "c2-64k"
p, t, d = get_workload(name)
e = KiwiAmd(p)
flat, offs = pack_texts(t)
for _ in range(3): e.analyze_packed(flat, offs, 1).close()
for parts in os.environ.get("PARTS", "1,2,4,8").split(","):
    os.environ["KAMD_BATCH_PARTS"] = parts
    e.analyze_packed(flat, offs, 1).close()
    ts = []
    for k in range(8):
        if k == 7: os.environ["KAMD_HOST_TIMING"] = "1"; sys.stderr.write(f"---- parts {parts}\n"); sys.stderr.flush()
        t0 = time.perf_counter(); r = e.analyze_packed(flat, offs, 1); t1 = time.perf_counter(); r.close(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    os.environ.pop("KAMD_HOST_TIMING", None)
    a = sorted(x + y for x, y in ts)[len(ts) // 2]
    # CPU time of the whole process per batch (what a CFS quota meters) and throttling over a run of batches
    def throttled():
        try: return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat")).get("nr_throttled", "?")
        except Exception: return "?"
    th0 = throttled(); c0 = time.process_time(); w0 = time.perf_counter(); r0 = resource.getrusage(resource.RUSAGE_SELF)
    for _ in range(20): e.analyze_packed(flat, offs, 1).close()
    c1 = time.process_time(); w1 = time.perf_counter(); r1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"{name} parts={parts}: per batch: user {(r1.ru_utime-r0.ru_utime)/20*1e3:.1f} ms, system {(r1.ru_stime-r0.ru_stime)/20*1e3:.1f} ms, minor faults {(r1.ru_minflt-r0.ru_minflt)/20:.0f}, voluntary switches {(r1.ru_nvcsw-r0.ru_nvcsw)/20:.0f}, involuntary {(r1.ru_nivcsw-r0.ru_nivcsw)/20:.0f}", flush=True)
    print(f"{name} parts={parts}: 20 batches back to back: {(w1-w0)/20*1e3:.2f} ms wall, {(c1-c0)/20*1e3:.1f} ms CPU per batch ({(c1-c0)/(w1-w0):.1f} cores busy), throttled periods {th0} -> {throttled()}", flush=True)
    print(f"{name} parts={parts}: median {a*1e3:.2f} ms per batch ({len(t)/a/1e6:.2f} M sentences/s), close {sorted(y for x, y in ts)[len(ts)//2]*1e3:.2f} ms", flush=True)
