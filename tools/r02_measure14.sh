#!/bin/bash
# Round-2 GPU call 16: typo graphs generated on the device (typo_graph_kernel.hip): the typo suite first, c5 / c2 bench lines, then the rest of the GPU suite.
TAG=${1:-r02p}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e',{}).get('value'), d.get('roofline',{}).get('frac'))"; }
timeout 300 python -m pytest tests/test_gpu_typo.py -m gpu -q -x > $OUT/pytest_gpu_typo.txt 2>&1; echo "pytest typo rc=$?"; tail -3 $OUT/pytest_gpu_typo.txt
timeout 150 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; show $OUT/bench_c5.json c5
KAMD_HOST_TIMING=1 timeout 100 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "\[host\]" | tail -12 > $OUT/host_timing_c5.txt; tail -6 $OUT/host_timing_c5.txt
timeout 240 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
timeout 420 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_typo.py > $OUT/pytest_gpu_rest.txt 2>&1; echo "pytest rest rc=$?"; tail -3 $OUT/pytest_gpu_rest.txt
