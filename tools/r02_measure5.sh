#!/bin/bash
# Round-2 GPU call 6: GPU suite; c2 line (bench.py now refuses launches that skipped chunks); c5 with the wave-per-chunk typo lattice kernel
# (4 and 5 waves per SIMD, and the thread-per-chunk kernel for comparison); c3-sbg proper (64k sentences, top-3, state arenas x8); c4-cong.
TAG=${1:-r02f}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 700 python -m pytest tests -m gpu -q --durations=6 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.txt
timeout 300 python bench.py --steps 30 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; cut -c1-300 $OUT/bench_c2.json; tail -2 $OUT/bench_c2.err
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['config']['kernel_ms'], d.get('e2e'), d.get('cpu_baseline'), d.get('roofline',{}).get('frac'))"; }
KAMD_HOST_TIMING=1 timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"; show $OUT/bench_c5.json c5; grep "typo lattices" $OUT/bench_c5.err | tail -1
KAMD_TYPO_LATTICE_WPS=5 timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_wps5.json 2> $OUT/bench_c5_wps5.err; show $OUT/bench_c5_wps5.json c5-wps5
KAMD_LATTICE_LDS=0 timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_hbm.json 2> $OUT/bench_c5_hbm.err; show $OUT/bench_c5_hbm.json c5-hbm-kernel
KAMD_HOST_TIMING=1 timeout 600 python bench.py --workload c3-sbg --steps 2 --warmup 1 > $OUT/bench_c3_sbg.json 2> $OUT/bench_c3_sbg.err; rc=$?; echo "c3-sbg rc=$rc"; grep -v "^\[host\] \(stage\|fetch\): \(text\|layout\|download\|post\)" $OUT/bench_c3_sbg.err | tail -8
if [ $rc -ne 0 ]; then
  KAMD_HOST_TIMING=1 timeout 400 python bench.py --workload c3-sbg --limit 16384 --steps 2 --warmup 1 > $OUT/bench_c3_sbg_16k.json 2> $OUT/bench_c3_sbg_16k.err; echo "c3-sbg-16k rc=$?"; grep -v "^\[host\] \(stage\|fetch\): \(text\|layout\|download\|post\)" $OUT/bench_c3_sbg_16k.err | tail -8
  show $OUT/bench_c3_sbg_16k.json c3-sbg-16k
else
  show $OUT/bench_c3_sbg.json c3-sbg
fi
timeout 400 python bench.py --workload c4-cong --steps 5 --warmup 1 > $OUT/bench_c4_cong.json 2> $OUT/bench_c4_cong.err; echo "c4 rc=$?"; show $OUT/bench_c4_cong.json c4-cong; tail -2 $OUT/bench_c4_cong.err
