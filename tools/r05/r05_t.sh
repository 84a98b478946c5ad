#!/bin/bash
# round 5, call t: the general search's LM step through the chain probe (ModelView::lmChain: the whole back-off walk of an order-4 model in one round trip) against the plain walk
# (-DKAMD_LM_PLAIN_WALK), c3-sbg first 16384 sentences; the SkipBigram suite on the new build
mkdir -p gpurun_out/r05_t; O=$PWD/gpurun_out/r05_t; rm -f $O/lm.txt
for lib in libkiwi_hip_plainwalk.so libkiwi_hip.so; do
KAMD_LIB=$PWD/kiwi_amd/$lib timeout 600 python - "$lib" >> $O/lm.txt 2>> $O/lm.err <<'PY'
import json, sys, bench
d = bench.side_measurement(None, "c3-sbg", steps=3, limit=16384)
print(sys.argv[1], {k: d[k] for k in ("value", "steps", "ms_per_step", "kernel_ms", "device_bytes", "rerun_chunks")})
PY
done
cat $O/lm.txt; tail -3 $O/lm.err
timeout 900 python -m pytest tests/test_gpu_sbg.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_sbg.txt
