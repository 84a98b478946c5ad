#!/bin/bash
# round 4: the C API after the calling-thread diet (lines of a batch in one buffer, result handles in one block per part, short first batches)
mkdir -p gpurun_out/r04_z; O=$PWD/gpurun_out/r04_z
timeout 200 python -m pytest tests/test_gpu_capi.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest_capi.txt
for i in 1 2; do KAMD_CAPI_TIMING=1 timeout 60 tools/_build/capi_bench _data/full.raw _data/c2-64k.corpus.txt 8 1 2>&1 | tail -2 | cut -c1-220; done | tee $O/capi_bench.txt
