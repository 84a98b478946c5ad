#!/bin/bash
# round 5, call m: state arenas -- per-chunk arenas sized for the typical chunk + the append-only pool (growArena); for the history models (SkipBigram, global CoNgram) arenas of the lane groups + the end stage inside the search kernel -- on the MI355X -- SkipBigram / global CoNgram / parity suites, then
# c3-sbg and c4-cong-global with the arena statistics (device bytes, pool use, re-runs)
mkdir -p gpurun_out/r05_m; O=$PWD/gpurun_out/r05_m
timeout 1500 python -m pytest tests/test_gpu_sbg.py tests/test_gpu_cong_global.py tests/test_gpu_fullmodel.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_sbg_congg_parity.txt
for w in c3-sbg c4-cong-global; do
timeout 900 python - $w > $O/side_$w.json 2> $O/side_$w.err <<'PY'
import json, sys, bench
print(json.dumps(bench.side_measurement(None, sys.argv[1], steps=3)))
PY
python - $O/side_$w.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print({k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks", "roofline_frac") if k in d})
PY
grep "state arenas" $O/side_$w.err | cut -c1-900
done
