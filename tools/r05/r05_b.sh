#!/bin/bash
# round 5, call b: per-chunk timeline of the SkipBigram search (first 8192 sentences of c3-sbg, top-3): the slowest chunks phase by phase
mkdir -p gpurun_out/r05_b; O=$PWD/gpurun_out/r05_b
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_tl5.so timeout 300 python tools/r05/sbg_timeline.py 8192 > $O/timeline_c3_sbg_8k.txt 2>&1; grep -c . $O/timeline_c3_sbg_8k.txt; grep "slow chunk\|wall\|first chunk" $O/timeline_c3_sbg_8k.txt | head -24 | cut -c1-300
