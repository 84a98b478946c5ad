#!/bin/bash
# round 5, the last GPU seconds: the build with the host-side result assembly changes (device code unchanged) -- first the C API + dialect suites (11 passed),
# then as much of the parity suite as 28 seconds hold (31 tests, all passed)
mkdir -p gpurun_out/r05_zx; O=$PWD/gpurun_out/r05_zx
timeout 28 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest_parity.txt
