#!/bin/bash
# round 5, call f: timing experiments (results knowingly wrong) -- is k_pos_path bound by the number of vector-memory accesses?  w1: one slot of every Knlm
# bucket loaded instead of four; w2: additionally no back-off weight loads
mkdir -p gpurun_out/r05_f; O=$PWD/gpurun_out/r05_f
for v in hip hip_w1 hip_w2; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_$v.so timeout 300 python tools/bench_multi.py c2-64k,c2 "g16-$v:KAMD_POS_G=16;g8-$v:KAMD_POS_G=8" 20 2>&1 | tee -a $O/bench_multi.txt | cut -c1-330
done
