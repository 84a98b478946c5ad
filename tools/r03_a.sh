#!/bin/bash
# round 3, call a: first contact of the position-step kernel -- parity (the two suites that cover the Knlm path), c2 / c2-64k with and without it
mkdir -p gpurun_out/r03_a; O=gpurun_out/r03_a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
for w in c2 c2-64k; do
  KAMD_POS_STATS=1 timeout 300 python bench.py --workload $w --steps 20 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  KAMD_POS_PATH=0 timeout 300 python bench.py --workload $w --steps 20 --no-cpu-baseline > $O/bench_${w}_general.json 2> $O/bench_${w}_general.err
  KAMD_WPS=2 timeout 300 python bench.py --workload $w --steps 20 --no-cpu-baseline > $O/bench_${w}_wps2.json 2> $O/bench_${w}_wps2.err
done
grep -h "pos\]" $O/*.err | sort | uniq -c | head
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_a/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "value %.3g"%j["value"], j["config"]["kernel_ms"], "e2e %.3g"%j["e2e"]["value"])
    except Exception as e: print(f, "ERR", e)
PY
