#!/bin/bash
# round 4: host share of the end-to-end rate (KAMD_HOST_TIMING), c5 with and without the position-step kernel, the dictionary scan after remembering terminals
mkdir -p gpurun_out/r04_j; O=$PWD/gpurun_out/r04_j
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lattices or tokens_bit or fuzzed" 2>&1 | tail -2
KAMD_HOST_TIMING=1 timeout 300 python tools/e2e_timing.py c2-64k > $O/e2e_timing.txt 2>&1; tail -24 $O/e2e_timing.txt
for PP in 1 0; do echo "== c5 KAMD_POS_PATH=$PP"; KAMD_POS_PATH=$PP timeout 300 python bench.py --workload c5 --kernels-only --steps 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}'; done
echo "== c2-64k"; timeout 300 python bench.py --workload c2-64k --kernels-only --steps 50 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}'
echo "== c4-cong"; timeout 300 python bench.py --workload c4-cong --kernels-only --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}'
