#!/bin/bash
# Round-2 GPU call 8: c3-sbg proper (64k sentences, SkipBigram, top-3); the lattice kernel with match records precomputed by all lanes: c2, c2-64k, c4-cong.
TAG=${1:-r02h}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d['config'].get('rerun_chunks'), d['config'].get('rerun_ms'), d.get('e2e'), d.get('cpu_baseline'), d.get('roofline',{}).get('frac'))"; }
KAMD_HOST_TIMING=1 timeout 900 python bench.py --workload c3-sbg --steps 2 --warmup 1 > $OUT/bench_c3_sbg.json 2> $OUT/bench_c3_sbg.err; rc=$?; echo "c3-sbg rc=$rc"; grep -v "^\[host\] \(stage\|fetch\): \(text\|layout\|download\|post\)" $OUT/bench_c3_sbg.err | tail -8
[ $rc -eq 0 ] && show $OUT/bench_c3_sbg.json c3-sbg
timeout 200 python bench.py --steps 30 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
timeout 200 python bench.py --workload c2-64k --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_64k.json 2> $OUT/bench_c2_64k.err; show $OUT/bench_c2_64k.json c2-64k
timeout 300 python bench.py --workload c4-cong --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c4_cong.json 2> $OUT/bench_c4_cong.err; show $OUT/bench_c4_cong.json c4-cong
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py -m gpu -q -x > $OUT/pytest_gpu_lattice.txt 2>&1; tail -3 $OUT/pytest_gpu_lattice.txt
