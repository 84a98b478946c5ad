#!/bin/bash
# Round-2 GPU call 9: c3-sbg proper (bench.py closes its staged batch before the end-to-end pass), then the whole GPU suite.
TAG=${1:-r02i}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d['config'].get('rerun_chunks'), d['config'].get('rerun_ms'), d.get('e2e'), d.get('cpu_baseline'), d.get('roofline',{}).get('frac'))"; }
KAMD_HOST_TIMING=1 timeout 1200 python bench.py --workload c3-sbg --steps 2 --warmup 1 > $OUT/bench_c3_sbg.json 2> $OUT/bench_c3_sbg.err; rc=$?; echo "c3-sbg rc=$rc"; grep -v "^\[host\] \(stage\|fetch\): \(text\|layout\|download\|post\)" $OUT/bench_c3_sbg.err | tail -6
[ $rc -eq 0 ] && show $OUT/bench_c3_sbg.json c3-sbg
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
