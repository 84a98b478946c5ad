#!/bin/bash
# round 6, call b: pretokenized spans on the MI355X (the new suite), the search suites that share the touched code (dialect blocklist bits, history-model arenas), a default bench line
mkdir -p gpurun_out/r06_b; O=$PWD/gpurun_out/r06_b
timeout 900 python -m pytest tests/test_gpu_pretokenized.py -m gpu -q -p no:cacheprovider > $O/pytest_pretokenized.txt 2>&1; tail -3 $O/pytest_pretokenized.txt
timeout 1500 python -m pytest tests/test_gpu_capi.py tests/test_dialect.py tests/test_gpu_sbg.py tests/test_gpu_cong_global.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest_touched.txt 2>&1; tail -3 $O/pytest_touched.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
