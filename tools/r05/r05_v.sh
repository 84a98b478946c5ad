#!/bin/bash
# round 5, call v: how much of their arenas the chunks of c2-64k and c4-cong use (the histogram behind stateScale64), and the device bytes of both
mkdir -p gpurun_out/r05_v; O=$PWD/gpurun_out/r05_v
for w in c2-64k c4-cong; do
KAMD_LATTICE_STATS=1 timeout 600 python - $w > $O/side_$w.txt 2> $O/side_$w.err <<'PY'
import json, sys, bench
d = bench.side_measurement(None, sys.argv[1], steps=5)
print(sys.argv[1], {k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks")})
PY
cat $O/side_$w.txt; grep "state arenas" $O/side_$w.err | cut -c1-700
done
