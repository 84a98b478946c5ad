#!/bin/bash
# round 4: a batch in parts (host stages under the neighbours' kernels), worker pool sized by the CPU quota: parity with forced parts, then e2e / capi
mkdir -p gpurun_out/r04_s; O=$PWD/gpurun_out/r04_s
KAMD_BATCH_PARTS=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py tests/test_gpu_capi.py tests/test_gpu_typo.py tests/test_gpu_cong.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_parts3.txt
for P in 0 1 2 4 8; do
  echo "== KAMD_BATCH_PARTS=$P (0: automatic)"
  ( [ $P = 0 ] || export KAMD_BATCH_PARTS=$P; KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c2-64k --no-cpu-baseline --no-side-models --steps 30 > $O/bench_p$P.json 2> $O/bench_p$P.err )
  python - $O/bench_p$P.json $O/bench_p$P.err <<'PY'
import json, sys, re, collections
d = json.load(open(sys.argv[1]))
print(" kernels %.0f sent/s | e2e %.0f sent/s %.1f ms/batch | capi %.0f sent/s %.1f ms/pass" % (d["value"], d["e2e"]["value"], d["e2e"]["ms_per_batch"], d["capi"]["value"], d["capi"]["ms_per_pass"]))
acc = collections.defaultdict(list)
for l in open(sys.argv[2]):
    m = re.match(r"\[host\] (.*) ([0-9.]+) ms", l)
    if m: acc[m.group(1)].append(float(m.group(2)))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print("   %-55s median %.2f  max %.2f  (n %d)" % (k, sorted(v)[len(v) // 2], max(v), len(v)))
PY
done 2>&1 | tee $O/parts.txt
