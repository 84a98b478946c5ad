#!/bin/bash
# round 4: whole GPU suite with k_lattice_wave, then the bench line of the default workload and of c4-cong
mkdir -p gpurun_out/r04_e; O=$PWD/gpurun_out/r04_e
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -14 $O/pytest_gpu.txt | cut -c1-200
timeout 600 python bench.py > $O/bench_c2-64k.json 2> $O/bench_c2-64k.err; cut -c1-1500 $O/bench_c2-64k.json
timeout 600 python bench.py --workload c4-cong --no-cpu-baseline > $O/bench_c4-cong.json 2> $O/bench_c4-cong.err; cut -c1-900 $O/bench_c4-cong.json
