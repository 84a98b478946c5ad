#!/bin/bash
# Round-2 second GPU call: the whole GPU suite with durations, first bench lines for configs 3 (SkipBigram, top-3) and 5 (typo), e2e timing.
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -32 $OUT/pytest_gpu.txt
timeout 300 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-400 $OUT/bench_c2.json
timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-400 $OUT/bench_c5.json; tail -3 $OUT/bench_c5.err
timeout 200 python tools/e2e_timing.py c2 > $OUT/e2e_c2.txt 2>&1; cat $OUT/e2e_c2.txt
timeout 300 python tools/e2e_timing.py c2-64k > $OUT/e2e_c2_64k.txt 2>&1; cat $OUT/e2e_c2_64k.txt
timeout 600 python bench.py --workload c3-sbg --steps 2 --warmup 1 > $OUT/bench_c3_sbg.json 2> $OUT/bench_c3_sbg.err; cut -c1-600 $OUT/bench_c3_sbg.json; tail -3 $OUT/bench_c3_sbg.err
