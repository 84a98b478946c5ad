#!/bin/bash
# round 5, call n: EXPERIMENT -- the SkipBigram search kernel built for three / four waves per SIMD (make EXTRA=-DKAMD_HIST_WPS=3|4: smaller LDS caches, 168 / 128 VGPRs)
# against the two-waves build, c3-sbg first 16384 sentences
mkdir -p gpurun_out/r05_n; O=$PWD/gpurun_out/r05_n
for cfg in "libkiwi_hip.so 8" "libkiwi_hip_wps3.so 12"; do
set -- $cfg
KAMD_LIB=$PWD/kiwi_amd/$1 KAMD_HIST_BLOCKS=$2 timeout 600 python - "$1 blocks/CU $2" >> $O/wps.txt 2>> $O/wps.err <<'PY'
import json, sys, bench
d = bench.side_measurement(None, "c3-sbg", steps=3, limit=16384)
print(sys.argv[1], {k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks")})
PY
done
cat $O/wps.txt; tail -3 $O/wps.err
