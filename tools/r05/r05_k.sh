#!/bin/bash
# round 5, call k: dialects on the MI355X -- every golden sentence through the low-level ABI, the C API suite (kiwi_init enabled_dialects, kiwi_analyze allowed_dialects / dialect_cost)
mkdir -p gpurun_out/r05_k; O=$PWD/gpurun_out/r05_k
timeout 900 python -m pytest tests/test_dialect.py tests/test_gpu_capi.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_dialect_capi.txt
