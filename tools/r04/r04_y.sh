#!/bin/bash
# round 4: the batch-in-parts test at scale (added after the final run)
mkdir -p gpurun_out/r04_y
timeout 300 python -m pytest tests/test_gpu_fullmodel.py -m gpu -x -q -k "batch_in_parts" 2>&1 | tail -3 | tee gpurun_out/r04_y/pytest.txt
