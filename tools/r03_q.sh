#!/bin/bash
# round 3, call q: the whole GPU suite (what the driver runs at round end), then smoke()
mkdir -p gpurun_out/r03_q; O=$PWD/gpurun_out/r03_q
timeout 1300 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_gpu.txt 2>&1; echo "rc $?" >> $O/pytest_gpu.txt
tail -40 $O/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
