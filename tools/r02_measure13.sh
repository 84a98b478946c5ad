#!/bin/bash
# Round-2 GPU call 15: wave-parallel removeUnconnected (lattice_connect.hpp) in both lattice kernels: parity tests, phase split, c2 / c2-64k / c3 / c4-cong / c5.
TAG=${1:-r02o}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e',{}).get('value'))"; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py tests/test_gpu_cong.py tests/test_gpu_typo.py -m gpu -q -x > $OUT/pytest_gpu_lattice.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_lattice.txt
python tools/lattice_phases.py c2 c3 2>&1 | grep -v amdgpu.ids | tee $OUT/lattice_phases.txt
timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
timeout 200 python bench.py --workload c2-64k --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_c2_64k.json 2> $OUT/bench_c2_64k.err; show $OUT/bench_c2_64k.json c2-64k
timeout 200 python bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json c3-knlm
timeout 300 python bench.py --workload c4-cong --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c4_cong.json 2> $OUT/bench_c4_cong.err; show $OUT/bench_c4_cong.json c4-cong
timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; show $OUT/bench_c5.json c5
