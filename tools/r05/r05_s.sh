#!/bin/bash
# round 5, call s: per-chunk timeline of the SkipBigram search after the round's changes (three waves per SIMD, slot arenas), scoring split into before-LM / Knlm / mixture+rest,
# emission into key-table build / rest (first 8192 sentences of c3-sbg, top-3)
mkdir -p gpurun_out/r05_s; O=$PWD/gpurun_out/r05_s
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_tl5.so timeout 300 python tools/r05/sbg_timeline.py 8192 > $O/timeline_c3_sbg_8k.txt 2>&1; grep -c . $O/timeline_c3_sbg_8k.txt; grep "timeline\|wall" $O/timeline_c3_sbg_8k.txt | head -30 | cut -c1-260
