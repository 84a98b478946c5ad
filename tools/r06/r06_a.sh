#!/bin/bash
# round 6, call a: k_pos_path with a new state's back-off chain fetched at the head of the next step (v2) against the round-5 order (v0)
mkdir -p gpurun_out/r06_a; O=$PWD/gpurun_out/r06_a
for v in v0 v2 v0 v2; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_$v.so timeout 300 python tools/bench_multi.py c2-64k,c2 "$v:" 30 2>&1 | tee -a $O/bench_multi.txt | sed 's/"env.*"kernel_ms"/"kernel_ms"/' | cut -c1-220
done
