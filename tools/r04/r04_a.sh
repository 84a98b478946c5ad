#!/bin/bash
# round 4, first call: k_lattice_wave on the MI355X -- lattice / token parity tests, then kernel statistics of c2-64k and c4-cong with the new and the old lattice kernel
mkdir -p gpurun_out/r04_a; O=$PWD/gpurun_out/r04_a; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py -m gpu -x -q --durations=8 > $O/pytest_gpu_lattice.txt 2>&1; echo "rc $?" >> $O/pytest_gpu_lattice.txt
tail -14 $O/pytest_gpu_lattice.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for WL in c2-64k c4-cong; do
  for MODE in 1 0; do
    KAMD_LATTICE_WAVE=$MODE timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --kernels-only > $O/trace_${WL}_wave$MODE.log 2>&1
    cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${WL}_wave$MODE.csv 2>/dev/null; rm -rf $O/trace
    echo "== $WL wave=$MODE"; head -9 $O/kernel_stats_${WL}_wave$MODE.csv | cut -c1-50,150-250
    tail -1 $O/trace_${WL}_wave$MODE.log | cut -c1-400
  done
done
cd $ROOT
KAMD_LATTICE_STATS=1 timeout 200 python bench.py --workload c4-cong --steps 2 --warmup 1 --kernels-only 2>&1 | grep "lattice wave" | tail -2
