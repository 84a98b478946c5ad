#!/bin/bash
# round 5, call a: the top-N key lists of the SkipBigram search (tools/r04/sbg_topn_key_lists.patch applied) on hardware -- its GPU tests, c3-sbg whole corpus,
# and the per-chunk timeline of the first 8192 sentences (slowest chunks phase by phase)
mkdir -p gpurun_out/r05_a; O=$PWD/gpurun_out/r05_a
timeout 400 python -m pytest tests/test_gpu_sbg.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest_sbg.txt
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_timeline.so timeout 300 python tools/r05/sbg_timeline.py 8192 > $O/timeline_c3_sbg_8k.txt 2>&1; grep -c . $O/timeline_c3_sbg_8k.txt; grep "slow chunk\|wall\|first chunk" $O/timeline_c3_sbg_8k.txt | head -24
timeout 600 python bench.py --workload c3-sbg --kernels-only --steps 2 --warmup 1 > $O/bench_c3_sbg.json 2> $O/bench_c3_sbg.err; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"device_bytes": [0-9]*' $O/bench_c3_sbg.json | head -3; tail -3 $O/bench_c3_sbg.err
