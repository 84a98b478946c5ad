#!/bin/bash
# Round-2 GPU call 11: GPU suite (top-N up to 16, dialect options accepted); c5 with the typo lattice kernel's context in registers
# (all methods inlined: LDS arrays addressed with ds_read / ds_write instead of flat accesses, 79 VGPRs).
TAG=${1:-r02k}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('cpu_baseline',{}).get('value'), d.get('roofline',{}).get('frac'))"; }
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"; show $OUT/bench_c5.json c5; grep "typo lattices" $OUT/bench_c5.err | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace5 -- python $ROOT/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace5.log 2>&1
cp $(find $OUT/trace5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c5.csv 2>/dev/null; head -5 $OUT/kernel_stats_c5.csv | cut -c1-220
rm -rf $OUT/trace5
