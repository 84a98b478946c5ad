#!/bin/bash
# round 5, call y: arena growth on the hardware (tests/test_gpu_sbg.py::test_state_arenas_grow_into_the_pool), then the measurement rounds of the two history workloads
# (bench line, rocprofv3 kernel statistics, PMC passes / HBM traffic): c3-sbg and c4-cong-global
mkdir -p gpurun_out/r05_y; O=$PWD/gpurun_out/r05_y
timeout 600 python -m pytest tests/test_gpu_sbg.py -m gpu -q -k "state_arenas" -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest_state_arenas.txt
timeout 900 bash tools/measure_round.sh r05_y c3-sbg 2>&1 | tail -12 | cut -c1-900
timeout 900 bash tools/measure_round.sh r05_y c4-cong-global 2>&1 | tail -12 | cut -c1-900
