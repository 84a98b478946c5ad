#!/bin/bash
# round 4: where a position step's cycles go now (KAMD_POS_DEBUG build: group-lane-0 clock per phase), c2-64k and c2
mkdir -p gpurun_out/r04_w; O=$PWD/gpurun_out/r04_w
for WL in c2-64k c2; do
  echo "== $WL"
  KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_posdebug.so KAMD_POS_BEACON=1 KAMD_POS_PHASES=1 timeout 300 python bench.py --workload $WL --kernels-only --steps 3 --warmup 1 2>&1 | grep "pos phases" | tail -1 | tr ';' '\n'
done 2>&1 | tee $O/pos_phases.txt
