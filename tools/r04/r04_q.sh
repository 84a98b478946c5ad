#!/bin/bash
# round 4: best path only for single-chunk texts under top-1 (k_finish_paths, D2H), host stage timing inside the end-to-end loop
mkdir -p gpurun_out/r04_q; O=$PWD/gpurun_out/r04_q
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py tests/test_gpu_capi.py -m gpu -x -q -k "tokens_bit or fuzzed or whole_corpora or c2 or golden or order_4 or capi or client" 2>&1 | tail -3 | tee $O/pytest.txt
KAMD_HOST_TIMING=1 timeout 600 python bench.py --workload c2-64k --no-cpu-baseline --no-side-models --steps 100 > $O/bench_c2_64k.json 2> $O/bench_c2_64k.err
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"e2e": {[^}]*}\|"capi": {[^}]*}' $O/bench_c2_64k.json | head -8
grep "\[host\]" $O/bench_c2_64k.err | tail -40
