#!/bin/bash
# round 5, call x: the round's measurement on the final build and the order-4 'full' models -- whole GPU suite, then tools/measure_round.sh (default bench line with
# config.also, rocprofv3 kernel statistics, PMC passes / traffic of c2-64k)
mkdir -p gpurun_out/r05_x; O=$PWD/gpurun_out/r05_x
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee $O/pytest_gpu.txt
timeout 1500 bash tools/measure_round.sh r05_x c2-64k 2>&1 | tail -30 | cut -c1-700
