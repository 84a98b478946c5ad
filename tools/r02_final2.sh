#!/bin/bash
# Round-2 last GPU call: the CoNgram GPU file (character model, builder-style cong.mdl), the official c2 line with its CPU baseline, rocprofv3 kernel
# statistics of the same command, then as much of the remaining GPU suite as the budget allows.
TAG=${1:-r02r}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e',{}).get('value'), d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))"; }
timeout 120 python -m pytest tests/test_gpu_cong.py -m gpu -q -x > $OUT/pytest_gpu_cong.txt 2>&1; echo "pytest cong rc=$?"; tail -2 $OUT/pytest_gpu_cong.txt
timeout 150 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
(cd /tmp && export TMPDIR=/tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/rocprof_bench.log 2>&1); f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_c2.csv && head -8 $OUT/kernel_stats_c2.csv | cut -c1-160; rm -rf $OUT/prof
timeout 200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_cong.py --deselect tests/test_gpu_fullmodel.py > $OUT/pytest_gpu_rest.txt 2>&1; echo "pytest rest rc=$?"; tail -2 $OUT/pytest_gpu_rest.txt
