#!/bin/bash
# round 5, call p: the history compilations (SkipBigram, global CoNgram) built for three waves per SIMD (-DKAMD_HIST_WPS=3, 12 blocks per CU) against two
mkdir -p gpurun_out/r05_p; O=$PWD/gpurun_out/r05_p; rm -f $O/wps.txt
for cfg in "libkiwi_hip.so 8 c4-cong-global 32768" "libkiwi_hip_wps3.so 12 c4-cong-global 32768" "libkiwi_hip.so 8 c3-sbg 16384" "libkiwi_hip_wps3.so 12 c3-sbg 16384"; do
set -- $cfg
KAMD_LIB=$PWD/kiwi_amd/$1 KAMD_HIST_BLOCKS=$2 timeout 600 python - "$1 blocks/CU $2" $3 $4 >> $O/wps.txt 2>> $O/wps.err <<'PY'
import json, sys, bench
d = bench.side_measurement(None, sys.argv[2], steps=3, limit=int(sys.argv[3]))
print(sys.argv[1], sys.argv[2], {k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks")})
PY
done
cat $O/wps.txt; tail -3 $O/wps.err
