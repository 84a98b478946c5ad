#!/bin/bash
# round 4: cycles per phase of k_lattice_wave (LW_PROFILE build), rounds per chunk, kernel statistics
mkdir -p gpurun_out/r04_d; O=$PWD/gpurun_out/r04_d; ROOT=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lattices or tokens_bit or fuzzed or lattice_hbm" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for WL in c2-64k c4-cong; do
KAMD_LIB=$PWD/kiwi_amd/libkiwi_hip_lwprof.so KAMD_LATTICE_PROFILE=1 KAMD_LATTICE_STATS=1 timeout 300 python bench.py --workload $WL --steps 2 --warmup 1 --kernels-only 2>&1 | grep "lattice profile\|lattice wave" | tail -2 | tee -a $O/profile.txt
done
cd /tmp && export TMPDIR=/tmp
for WL in c2-64k c4-cong; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $ROOT/bench.py --workload $WL --steps 10 --warmup 3 --kernels-only > $O/trace_${WL}.log 2>&1
    cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${WL}.csv 2>/dev/null; rm -rf $O/trace
    echo "== $WL"; head -9 $O/kernel_stats_${WL}.csv | cut -c1-50,150-250
    tail -1 $O/trace_${WL}.log | cut -c1-300
done
