#!/bin/bash
# round 4: cycles per phase of k_lattice_wave (LW_PROFILE build) and rounds per chunk
mkdir -p gpurun_out/r04_d; O=$PWD/gpurun_out/r04_d
for WL in c2-64k c4-cong; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_lwprof.so KAMD_LATTICE_PROFILE=1 KAMD_LATTICE_STATS=1 timeout 300 python bench.py --workload $WL --steps 2 --warmup 1 --kernels-only 2>&1 | grep "lattice profile\|lattice wave" | tail -2 | tee -a $O/profile.txt
done
