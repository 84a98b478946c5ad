#!/bin/bash
# round 4: k_finish_paths after the one-pass group count; lanes per chunk
mkdir -p gpurun_out/r04_m; O=$PWD/gpurun_out/r04_m
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tokens_bit or top_n" 2>&1 | tail -2
for ST in 0 2 8 16; do echo "== c2-64k KAMD_FINISH_STRIDE=$ST"; KAMD_FINISH_STRIDE=$ST timeout 300 python bench.py --workload c2-64k --kernels-only --steps 50 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"finish_ms": [0-9.]*'; done
for ST in 0 8 16; do echo "== c4-cong KAMD_FINISH_STRIDE=$ST"; KAMD_FINISH_STRIDE=$ST timeout 300 python bench.py --workload c4-cong --kernels-only --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"finish_ms": [0-9.]*'; done
