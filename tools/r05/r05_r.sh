#!/bin/bash
# round 5, call r (second run; the first measured a 64-lane register de-duplication path -- no gain, removed): right halves of split stems find their left half through one pass over the paths instead of a scan per item -- SkipBigram / global CoNgram suites, c3-sbg and c4-cong-global
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
