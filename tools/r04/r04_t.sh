#!/bin/bash
# round 4: what stalls a host stage for 20 - 40 ms now and then -- CFS throttling (cpu.stat) or the allocator (mmap / munmap of the per-batch vectors)?
mkdir -p gpurun_out/r04_t; O=$PWD/gpurun_out/r04_t
run() {
  echo "== $*"
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  ( export "$@"; KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c2-64k --no-cpu-baseline --no-side-models --steps 30 > $O/b.json 2> $O/b.err )
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  echo " cpu.stat before: $a | after: $b"
  python - $O/b.json $O/b.err <<'PY'
import json, sys, re, collections
d = json.load(open(sys.argv[1]))
print(" e2e %.0f sent/s %.1f ms/batch | capi %.0f sent/s %.1f ms/pass" % (d["e2e"]["value"], d["e2e"]["ms_per_batch"], d["capi"]["value"], d["capi"]["ms_per_pass"]))
acc = collections.defaultdict(list)
for l in open(sys.argv[2]):
    m = re.match(r"\[host\] (.*) ([0-9.]+) ms", l)
    if m: acc[m.group(1)].append(float(m.group(2)))
for k, v in acc.items():
    if "stage" in k or "fetch" in k:
        v = v[len(v) // 2:]
        print("   %-50s median %.2f mean %.2f max %.2f (n %d)" % (k, sorted(v)[len(v) // 2], sum(v) / len(v), max(v), len(v)))
PY
}
run X=1
run KAMD_BATCH_PARTS=1
run MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456
run MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184 MALLOC_TOP_PAD_=268435456 KAMD_BATCH_PARTS=1
run MALLOC_ARENA_MAX=4 MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184
run KAMD_HOST_THREADS=32
run KAMD_HOST_THREADS=32 MALLOC_MMAP_THRESHOLD_=4294967296 MALLOC_TRIM_THRESHOLD_=17179869184
