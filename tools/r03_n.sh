#!/bin/bash
# round 3, call n: handed-over chunks carried on by k_pos_path itself (a real function reading the kernarg segment)
mkdir -p gpurun_out/r03_n; O=$PWD/gpurun_out/r03_n
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 4000 > $O/check_c2.txt 2>&1; echo "rc $?" >> $O/check_c2.txt
tail -3 $O/check_c2.txt | cut -c1-400
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_smallcaps.so KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/quick_gpu.py 300 > $O/check_smallcaps.txt 2>&1; echo "rc $?" >> $O/check_smallcaps.txt
tail -3 $O/check_smallcaps.txt | cut -c1-400
if grep -q "bad 0 /" $O/check_c2.txt; then
  KAMD_POS_STATS=1 timeout 300 python tools/bench_multi.py c2,c2-64k "pos:;nocont:KAMD_POS_CONT=0" 20 > $O/bench_multi.txt 2> $O/bench_multi.err
  cat $O/bench_multi.txt | cut -c1-330; grep "pos\]" $O/bench_multi.err | sort | uniq -c | cut -c1-400
  export TMPDIR=/tmp
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --kernels-only > $O/trace.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c2-64k.csv 2>/dev/null; rm -rf $O/prof
  head -8 $O/kernel_stats_c2-64k.csv | cut -c1-60,150-260
fi
