#!/bin/bash
# round 5, call c: the top-N pruning threshold (N-th best score per root instead of every path counting the paths above it) on hardware: the SkipBigram GPU tests,
# c3-sbg whole corpus, timeline of the first 8192 sentences
mkdir -p gpurun_out/r05_c; O=$PWD/gpurun_out/r05_c
timeout 400 python -m pytest tests/test_gpu_sbg.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest_sbg.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "top_n or topn or top" 2>&1 | tail -2 | tee $O/pytest_topn.txt
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_tl5.so timeout 300 python tools/r05/sbg_timeline.py 8192 > $O/timeline_c3_sbg_8k.txt 2>&1; grep "slow chunk\|wall\|first chunk" $O/timeline_c3_sbg_8k.txt | head -8 | cut -c1-300
timeout 600 python bench.py --workload c3-sbg --kernels-only --steps 3 --warmup 1 > $O/bench_c3_sbg.json 2> $O/bench_c3_sbg.err; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"device_bytes": [0-9]*' $O/bench_c3_sbg.json | head -3; tail -3 $O/bench_c3_sbg.err
