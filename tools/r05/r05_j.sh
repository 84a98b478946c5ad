#!/bin/bash
# round 5, call j: BASELINE config 3's corpus against the oracle -- the first 8192 sentences and the 64 heaviest of c3-sbg, top-3 --; how much of the SkipBigram
# state arenas a batch uses; the lattice / parity families after the engine changes
mkdir -p gpurun_out/r05_j; O=$PWD/gpurun_out/r05_j
timeout 900 python -m pytest tests/test_gpu_fullmodel.py -m gpu -x -q -k "c3_sbg_corpus" 2>&1 | tail -3 | tee $O/pytest_c3_sbg_8192.txt
KAMD_LATTICE_STATS=1 timeout 600 python bench.py --workload c3-sbg --kernels-only --steps 1 --warmup 1 > $O/bench_c3_sbg.json 2> $O/bench_c3_sbg.err; grep "state arenas" $O/bench_c3_sbg.err | tail -3; grep -o '"ms_per_step": [0-9.]*\|"device_bytes": [0-9]*' $O/bench_c3_sbg.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cong.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_parity_cong.txt
