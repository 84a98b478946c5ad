#!/bin/bash
# round 6, call g: k_build_lattice_typo_lds built for more waves per SIMD (launch bounds 5 / 6 / 8; default 4 = 121 VGPRs) on c5
mkdir -p gpurun_out/r06_g; O=$PWD/gpurun_out/r06_g
for rep in 1 2; do for v in hip hip_tl5 hip_tl6 hip_tl8; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_$v.so timeout 400 python tools/bench_multi.py c5 "$v:" 20 2>&1 | tee -a $O/bench_multi.txt | sed 's/"env.*"kernel_ms"/"kernel_ms"/' | cut -c1-230
done; done
