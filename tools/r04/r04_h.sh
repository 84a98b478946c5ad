#!/bin/bash
# round 4: whole GPU suite, then the default bench line (every BASELINE config in config.also) and the measurement of the default workload
mkdir -p gpurun_out/r04_h; O=$PWD/gpurun_out/r04_h
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -14 $O/pytest_gpu.txt | cut -c1-200
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/bench_default.json; tail -3 $O/bench_default.err
