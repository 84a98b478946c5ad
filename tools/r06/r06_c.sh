#!/bin/bash
# round 6, call c: form-trie edges as a hash (one load per automaton step) -- dictionary scan and typo lattice times, then the suites that walk the trie
mkdir -p gpurun_out/r06_c; O=$PWD/gpurun_out/r06_c
timeout 600 python tools/bench_multi.py c2-64k,c2,c5,c4-cong "edgehash:" 20 2>&1 | tee $O/bench_multi.txt | sed 's/"env.*"kernel_ms"/"kernel_ms"/' | cut -c1-260
timeout 1500 python -m pytest tests/test_gpu_typo.py tests/test_gpu_parity.py tests/test_gpu_capi.py -m gpu -q -x -p no:cacheprovider > $O/pytest_trie.txt 2>&1; tail -3 $O/pytest_trie.txt
