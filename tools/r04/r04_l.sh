#!/bin/bash
# round 4: default bench line after the state-arena / SkipBigram-model changes (device_bytes, c3-sbg whole corpus), GPU suite
mkdir -p gpurun_out/r04_l; O=$PWD/gpurun_out/r04_l
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt | cut -c1-200
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json; tail -3 $O/bench_default.err
