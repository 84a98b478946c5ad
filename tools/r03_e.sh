#!/bin/bash
# round 3, call e: per-kernel durations of the position-step path (rocprofv3 --kernel-trace --stats), c2 and c2-64k
mkdir -p gpurun_out/r03_e; O=$PWD/gpurun_out/r03_e
export TMPDIR=/tmp
cd /tmp
for w in c2 c2-64k; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python $GRAFT_REPO_ROOT/tools/bench_multi.py $w "pos:" 20 > $O/bench_$w.txt 2> $O/bench_$w.err
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv 2>/dev/null
  head -12 $O/kernel_stats_$w.csv | cut -c1-220
  rm -rf $O/prof_$w
done
