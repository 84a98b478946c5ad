#!/bin/bash
# Round-2 third GPU call: the GPU suite on the rebuilt host pipeline, bench lines with the e2e block, host timing breakdown, SkipBigram at small batch.
TAG=${1:-r02c}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.txt
KAMD_HOST_TIMING=1 timeout 300 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-300 $OUT/bench_c2.json; python -c "import json;d=json.load(open('$OUT/bench_c2.json'));print(d['e2e']);print(d['cpu_baseline'])"; grep host $OUT/bench_c2.err | tail -12
KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c2-64k --steps 10 > $OUT/bench_c2_64k.json 2> $OUT/bench_c2_64k.err; python -c "import json;d=json.load(open('$OUT/bench_c2_64k.json'));print(d['value'], d['config']['kernel_ms']);print(d['e2e']);print(d['cpu_baseline'])"; grep host $OUT/bench_c2_64k.err | tail -6
timeout 200 python bench.py --workload c3-sbg --limit 1024 --steps 2 --warmup 1 > $OUT/bench_c3_sbg_1k.json 2> $OUT/bench_c3_sbg_1k.err; cut -c1-1500 $OUT/bench_c3_sbg_1k.json; tail -3 $OUT/bench_c3_sbg_1k.err
