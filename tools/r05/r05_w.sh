#!/bin/bash
# round 5, call w: live-path compaction for every group width (threshold: more incoming paths than the group has lanes) -- SkipBigram / global CoNgram / full-model / typo / parity suites,
# then c5 (typo lattices take the 16-lane general kernel), c4-cong and c3-sbg
mkdir -p gpurun_out/r05_w; O=$PWD/gpurun_out/r05_w
timeout 1700 python -m pytest tests/test_gpu_sbg.py tests/test_gpu_cong_global.py tests/test_gpu_fullmodel.py tests/test_gpu_typo.py tests/test_gpu_parity.py tests/test_gpu_cong.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest_suites.txt
for w in c5 c4-cong c3-sbg; do
timeout 900 python - $w > $O/side_$w.json 2> $O/side_$w.err <<'PY'
import json, sys, bench
print(json.dumps(bench.side_measurement(None, sys.argv[1], steps=3, min_seconds=0.5 if sys.argv[1] == "c5" else 0.0)))
PY
python - $O/side_$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print({k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks", "roofline_frac") if k in d})
PY
done
