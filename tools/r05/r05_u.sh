#!/bin/bash
# round 5, call u: EXPERIMENT -- the SkipBigram mixture as a real function call (-DKAMD_SBG_CALL: registers saved / restored around it in bursts) against the inlined one
mkdir -p gpurun_out/r05_u; O=$PWD/gpurun_out/r05_u; rm -f $O/call.txt
for lib in libkiwi_hip.so libkiwi_hip_call.so; do
KAMD_LIB=$PWD/kiwi_amd/$lib timeout 600 python - "$lib" >> $O/call.txt 2>> $O/call.err <<'PY'
import json, sys, bench
d = bench.side_measurement(None, "c3-sbg", steps=3, limit=16384)
print(sys.argv[1], {k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks")})
PY
done
cat $O/call.txt; tail -3 $O/call.err
