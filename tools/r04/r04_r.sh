#!/bin/bash
# round 4: host side of the end-to-end batch -- worker count against the container's CPU quota, kiwi_analyze_m batch size
mkdir -p gpurun_out/r04_r; O=$PWD/gpurun_out/r04_r
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
python - <<'PY'
import sys; sys.path.insert(0, '.')
from kiwi_amd.workloads import get_workload
import bench
p, t, d = get_workload("c2-64k")      # model + corpus file for the C client
print(p)
PY
for HT in 0 64 32 24 16; do
  echo "== KAMD_HOST_THREADS=$HT"
  ( [ $HT = 0 ] || export KAMD_HOST_THREADS=$HT; KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c2-64k --no-cpu-baseline --no-side-models --steps 30 > $O/bench_ht$HT.json 2> $O/bench_ht$HT.err )
  python - $O/bench_ht$HT.json $O/bench_ht$HT.err <<'PY'
import json, sys, re, collections
d = json.load(open(sys.argv[1]))
print(" e2e %.0f sent/s %.1f ms/batch | capi %.0f sent/s %.1f ms/pass" % (d["e2e"]["value"], d["e2e"]["ms_per_batch"], d["capi"]["value"], d["capi"]["ms_per_pass"]))
acc = collections.defaultdict(list)
for l in open(sys.argv[2]):
    m = re.match(r"\[host\] (.*) ([0-9.]+) ms", l)
    if m: acc[m.group(1)].append(float(m.group(2)))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print("   %-55s median %.2f  max %.2f  (n %d)" % (k, sorted(v)[len(v) // 2], max(v), len(v)))
PY
done 2>&1 | tee $O/host_threads.txt
M=_data/full.raw; C=_data/c2-64k.corpus.txt
for CB in 65536 32768 16384 8192; do for HT in 0 32; do
  echo "== KAMD_CAPI_BATCH=$CB KAMD_HOST_THREADS=$HT"
  ( [ $HT = 0 ] || export KAMD_HOST_THREADS=$HT; KAMD_CAPI_BATCH=$CB KAMD_CAPI_TIMING=1 timeout 120 tools/_build/capi_bench $M $C 8 1 2>&1 | tail -3 | cut -c1-220 )
done; done 2>&1 | tee $O/capi_batch.txt
