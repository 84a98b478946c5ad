#!/bin/bash
# round 5, call h: child / context signatures in front of the Knlm bucket fetches (k_pos_path): parity families of the position steps, then 16- against 8-lane groups
mkdir -p gpurun_out/r05_h; O=$PWD/gpurun_out/r05_h
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pos or knlm or order" 2>&1 | tail -3 | tee $O/pytest_pos.txt
timeout 600 python tools/bench_multi.py c2-64k,c2 "g16:KAMD_POS_G=16;g8:KAMD_POS_G=8" 20 2>&1 | tee $O/bench_multi.txt | sed 's/"env.*"kernel_ms"/"kernel_ms"/' | cut -c1-260
