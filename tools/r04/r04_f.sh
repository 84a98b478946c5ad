#!/bin/bash
# round 4: the lattice kernel writes candidate records and the position program itself -- parity, kernel statistics, bench lines
mkdir -p gpurun_out/r04_f; O=$PWD/gpurun_out/r04_f; ROOT=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py tests/test_gpu_cong.py -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -10 $O/pytest_gpu.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for WL in c2-64k c4-cong; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $ROOT/bench.py --workload $WL --steps 10 --warmup 3 --kernels-only > $O/trace_${WL}.log 2>&1
    cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${WL}.csv 2>/dev/null; rm -rf $O/trace
    echo "== $WL"; head -9 $O/kernel_stats_${WL}.csv | cut -c1-50,150-250
    tail -1 $O/trace_${WL}.log | cut -c1-700
done
