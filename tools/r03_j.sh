#!/bin/bash
# round 3, call i: where a position step's time goes (phase timers of the KAMD_POS_DEBUG build), c2 and c2-64k
mkdir -p gpurun_out/r03_j; O=$PWD/gpurun_out/r03_j
for w in c2 c2-64k; do
  KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_posdebug.so KAMD_POS_BEACON=1 KAMD_POS_PHASES=1 timeout 200 python tools/bench_multi.py $w "pos:" 3 > $O/phases_$w.txt 2> $O/phases_$w.err
  grep "pos phases" $O/phases_$w.err | tail -1 | cut -c1-900; cut -c1-300 $O/phases_$w.txt
done
