#!/bin/bash
# Round-2 GPU call 12: GPU suite (typo correction with CoNgram models, typo graphs generated on the host pool); c5 with its end-to-end rate; c2.
TAG=${1:-r02l}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e'), d.get('cpu_baseline',{}).get('value'), d.get('roofline',{}).get('frac'))"; }
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"; show $OUT/bench_c5.json c5; grep "\[host\]" $OUT/bench_c5.err | tail -12
timeout 300 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; show $OUT/bench_c2.json c2
