#!/bin/bash
# Round-2 GPU call 7: c3-sbg proper (64k sentences, SkipBigram, top-3; kamd_run now contains the capacity ladder), rocprofv3 kernel statistics of c5
# (wave-per-chunk typo lattice kernel) and c4-cong.
TAG=${1:-r02g}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d['config'].get('rerun_chunks'), d['config'].get('rerun_ms'), d.get('e2e'), d.get('cpu_baseline'), d.get('roofline',{}).get('frac'))"; }
KAMD_HOST_TIMING=1 timeout 900 python bench.py --workload c3-sbg --steps 2 --warmup 1 > $OUT/bench_c3_sbg.json 2> $OUT/bench_c3_sbg.err; rc=$?; echo "c3-sbg rc=$rc"; grep -v "^\[host\] \(stage\|fetch\): \(text\|layout\|download\|post\)" $OUT/bench_c3_sbg.err | tail -8
[ $rc -eq 0 ] && show $OUT/bench_c3_sbg.json c3-sbg
timeout 200 python bench.py --steps 30 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace5 -- python $ROOT/bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace5.log 2>&1
cp $(find $OUT/trace5 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c5.csv 2>/dev/null; head -8 $OUT/kernel_stats_c5.csv | cut -c1-220
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace4 -- python $ROOT/bench.py --workload c4-cong --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace4.log 2>&1
cp $(find $OUT/trace4 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c4_cong.csv 2>/dev/null; head -8 $OUT/kernel_stats_c4_cong.csv | cut -c1-220
rm -rf $OUT/trace5 $OUT/trace4
