#!/bin/bash
# Round-2 GPU call 4: GPU suite after the boundary / multi-GPU changes, SkipBigram with one chunk per wave, rocprofv3 kernel stats of bench c2.
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q -x --durations=6 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.txt
timeout 300 python bench.py --workload c3-sbg --limit 1024 --steps 2 --warmup 1 > $OUT/bench_c3_sbg_1k.json 2> $OUT/bench_c3_sbg_1k.err; echo "sbg1k rc=$?"; cut -c1-1800 $OUT/bench_c3_sbg_1k.json; tail -3 $OUT/bench_c3_sbg_1k.err
timeout 500 python bench.py --workload c3-sbg --limit 8192 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c3_sbg_8k.json 2> $OUT/bench_c3_sbg_8k.err; echo "sbg8k rc=$?"; cut -c1-1200 $OUT/bench_c3_sbg_8k.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o c2 -- python $ROOT/bench.py --no-cpu-baseline > $OUT/prof_c2.log 2>&1); echo "rocprof rc=$?"
find $OUT/prof_c2 -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
