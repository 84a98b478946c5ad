#!/bin/bash
# Round-2 GPU call 10: GPU suite (blocklists, cong.mdl directories); the Knlm search kernel at 4 waves per SIMD (128 VGPRs, spills) on c2-64k.
TAG=${1:-r02j}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e',{}).get('value'))"; }
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
timeout 200 python bench.py --workload c2-64k --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_64k.json 2> $OUT/bench_c2_64k.err; show $OUT/bench_c2_64k.json c2-64k-default
KAMD_WPS=4 KAMD_GROUP_LANES=8 timeout 200 python bench.py --workload c2-64k --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_64k_g8w4.json 2> $OUT/bench_c2_64k_g8w4.err; show $OUT/bench_c2_64k_g8w4.json c2-64k-g8-wps4
KAMD_WPS=4 KAMD_GROUP_LANES=16 timeout 200 python bench.py --workload c2-64k --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_64k_g16w4.json 2> $OUT/bench_c2_64k_g16w4.err; show $OUT/bench_c2_64k_g16w4.json c2-64k-g16-wps4
KAMD_WPS=3 KAMD_GROUP_LANES=16 timeout 200 python bench.py --workload c2-64k --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_64k_g16w3.json 2> $OUT/bench_c2_64k_g16w3.err; show $OUT/bench_c2_64k_g16w3.json c2-64k-g16-wps3
KAMD_WPS=4 KAMD_GROUP_LANES=8 timeout 200 python bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c3_g8w4.json 2> $OUT/bench_c3_g8w4.err; show $OUT/bench_c3_g8w4.json c3-knlm-g8-wps4
timeout 200 python bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json c3-knlm-default
timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
