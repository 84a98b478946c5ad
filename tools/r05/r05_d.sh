#!/bin/bash
# round 5, call d: first hardware contact of k_pos_path<8, .> (eight chunks per wavefront): parity families that run the position steps, forced to 8-lane groups;
# then 16- against 8-lane groups on c2-64k, c2 and c4-cong
mkdir -p gpurun_out/r05_d; O=$PWD/gpurun_out/r05_d
KAMD_POS_G=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cong.py -m gpu -x -q -k "pos" 2>&1 | tail -3 | tee $O/pytest_pos_g8.txt
timeout 600 python tools/bench_multi.py c2-64k,c2,c4-cong "g16:KAMD_POS_G=16;g8:KAMD_POS_G=8" 20 2>&1 | tee $O/bench_multi.txt | cut -c1-330
