#!/bin/bash
# round 5, call i: waves per SIMD x lane-group width x signature filter on c2-64k (k_pos_path)
mkdir -p gpurun_out/r05_i; O=$PWD/gpurun_out/r05_i
for v in hip hip_w3; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_$v.so timeout 300 python tools/bench_multi.py c2-64k "g16w3-$v:KAMD_POS_G=16,KAMD_WPS=3;g16w2-$v:KAMD_POS_G=16,KAMD_WPS=2;g8w3-$v:KAMD_POS_G=8,KAMD_WPS=3;g8w2-$v:KAMD_POS_G=8,KAMD_WPS=2" 20 2>&1 | tee -a $O/bench_multi.txt | sed 's/"env.*"kernel_ms"/"kernel_ms"/' | cut -c1-200
done
