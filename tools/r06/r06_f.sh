#!/bin/bash
# round 6, call f: scheduler strategies of the AMDGPU backend applied at the device link (-Xoffload-linker -mllvm=...), whole library: max-ilp, max-memory-clause,
# the AMDGPU register-pressure trackers, relaxed occupancy -- against the default build, same process order twice
mkdir -p gpurun_out/r06_f; O=$PWD/gpurun_out/r06_f
for rep in 1 2; do for v in hip hip_f_ilp hip_f_mem hip_f_trk hip_f_relax; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_$v.so timeout 400 python tools/bench_multi.py c2-64k,c4-cong "$v:" 20 2>&1 | tee -a $O/bench_multi.txt | sed 's/"env.*"kernel_ms"/"kernel_ms"/' | cut -c1-230
done; done
