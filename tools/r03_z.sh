#!/bin/bash
# round 3, final call: the whole GPU suite, smoke(), then the measurement of the default workload (bench line, kernel statistics, PMC passes)
mkdir -p gpurun_out/r03_z; O=$PWD/gpurun_out/r03_z
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -18 $O/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/measure_round.sh r03_z c2-64k
