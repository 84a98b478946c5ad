#!/bin/bash
# round 5, the last GPU seconds: live-path compaction from sixteen incoming paths on (was: more than the group's lanes) -- c3-sbg whole, then the SkipBigram suite
mkdir -p gpurun_out/r05_zy; O=$PWD/gpurun_out/r05_zy
timeout 70 python - > $O/side_c3-sbg.json 2> $O/side.err <<'PY'
import json, bench
d = bench.side_measurement(None, "c3-sbg", steps=3)
print(json.dumps({k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks", "roofline_frac")}))
PY
cat $O/side_c3-sbg.json; tail -2 $O/side.err
timeout 40 python -m pytest tests/test_gpu_sbg.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_sbg.txt
