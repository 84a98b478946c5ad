#!/bin/bash
# round 5, call q: EXPERIMENT -- lane-group width of the SkipBigram search in the throughput regime: 16-lane groups (four chunks per wavefront) against one chunk per wavefront
mkdir -p gpurun_out/r05_q; O=$PWD/gpurun_out/r05_q; rm -f $O/g.txt
for g in 64 16; do
KAMD_GROUP_LANES=$g timeout 900 python - "lanes $g" >> $O/g.txt 2>> $O/g.err <<'PY'
import json, sys, bench
d = bench.side_measurement(None, "c3-sbg", steps=3, limit=16384)
print(sys.argv[1], {k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks")})
PY
done
cat $O/g.txt; tail -3 $O/g.err
