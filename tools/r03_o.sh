#!/bin/bash
# round 3, call o: ring-slot ownership fix; k_build_lattice with four chunks per wavefront
mkdir -p gpurun_out/r03_o; O=$PWD/gpurun_out/r03_o
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 4000 > $O/check_c2.txt 2>&1; echo "rc $?" >> $O/check_c2.txt
tail -2 $O/check_c2.txt | cut -c1-400
KAMD_LATTICE_GROUP=16 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 2000 > $O/check_c2_lat16.txt 2>&1; echo "rc $?" >> $O/check_c2_lat16.txt
tail -2 $O/check_c2_lat16.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_typo.py tests/test_gpu_cong.py -x -q -m gpu -k "pos" > $O/pytest_pos.txt 2>&1; tail -3 $O/pytest_pos.txt
if grep -q "bad 0 /" $O/check_c2.txt && grep -q "bad 0 /" $O/check_c2_lat16.txt; then
  timeout 300 python tools/bench_multi.py c2,c2-64k "default:;lat16:KAMD_LATTICE_GROUP=16;lat64:KAMD_LATTICE_GROUP=64" 20 > $O/bench_multi.txt 2> $O/bench_multi.err
  cat $O/bench_multi.txt | cut -c1-330
  export TMPDIR=/tmp
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --kernels-only > $O/trace.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c2-64k.csv 2>/dev/null; rm -rf $O/prof
  head -8 $O/kernel_stats_c2-64k.csv | cut -c1-60,150-260
fi
