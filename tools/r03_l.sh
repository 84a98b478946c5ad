#!/bin/bash
# round 3, call l: further chunk words through the chain, item map by marking, records through LDS (early loads), 32-state ring
mkdir -p gpurun_out/r03_l; O=$PWD/gpurun_out/r03_l
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_posdebug.so KAMD_HANGDUMP=1 KAMD_POS_BEACON=1 KAMD_POS_STATS=1 timeout 120 python tools/quick_gpu.py 300 > $O/debug_small.txt 2>&1; echo "rc $?" >> $O/debug_small.txt
tail -4 $O/debug_small.txt | cut -c1-300
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 4000 > $O/check_c2.txt 2>&1; echo "rc $?" >> $O/check_c2.txt
tail -4 $O/check_c2.txt | cut -c1-300
if grep -q "bad 0 /" $O/check_c2.txt; then
  for w in c2 c2-64k; do
    KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_posdebug.so KAMD_POS_BEACON=1 KAMD_POS_PHASES=1 timeout 200 python tools/bench_multi.py $w "pos:" 3 > $O/phases_$w.txt 2> $O/phases_$w.err
    grep "pos phases" $O/phases_$w.err | tail -1 | cut -c1-1100
  done
  export TMPDIR=/tmp
  ( cd /tmp; for w in c2 c2-64k; do
    KAMD_POS_STATS=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python $GRAFT_REPO_ROOT/tools/bench_multi.py $w "pos:" 20 > $O/bench_$w.txt 2> $O/bench_$w.err
    f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_$w.csv 2>/dev/null
    head -6 $O/kernel_stats_$w.csv | cut -c1-60,150-260
    rm -rf $O/prof_$w
  done )
  KAMD_POS_STATS=1 timeout 200 python tools/bench_multi.py c2,c2-64k "pos:;pos-wps2:KAMD_WPS=2" 20 > $O/bench_multi.txt 2> $O/bench_multi.err
  cat $O/bench_multi.txt | cut -c1-330; grep "pos\]" $O/bench_multi.err | sort | uniq -c | cut -c1-300
fi
