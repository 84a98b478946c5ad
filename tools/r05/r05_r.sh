#!/bin/bash
# round 5, call r (third run): the items of a node formed over its LIVE incoming paths (two thirds of a SkipBigram top-3 node's paths are pruned ones) -- suites, c3-sbg, c4-cong-global
mkdir -p gpurun_out/r05_r; O=$PWD/gpurun_out/r05_r
timeout 1500 python -m pytest tests/test_gpu_sbg.py tests/test_gpu_cong_global.py tests/test_gpu_fullmodel.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_sbg_congg_fullmodel.txt
for w in c3-sbg c4-cong-global; do
timeout 900 python - $w > $O/side_$w.json 2> $O/side_$w.err <<'PY'
import json, sys, bench
print(json.dumps(bench.side_measurement(None, sys.argv[1], steps=3)))
PY
python - $O/side_$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print({k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks", "roofline_frac") if k in d})
PY
done
