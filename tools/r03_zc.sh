#!/bin/bash
# round 3, last call: GPU suite of the final build, smoke(), the default bench line and the c4-cong line
mkdir -p gpurun_out/r03_zc; O=$PWD/gpurun_out/r03_zc
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-250 $O/bench_default.json
bash tools/measure_round.sh r03_zc c4-cong | tail -4 | cut -c1-300
