#!/bin/bash
# Round-2 GPU call 17: model-file variants (history-transformed / quantised sj.knlm, cong.mdl with variable-length keys + 4-bit embeddings) on the device;
# c2 and c4-cong lines after lmProgress / congStep changed.
TAG=${1:-r02q}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e',{}).get('value'), d.get('roofline',{}).get('frac'))"; }
timeout 300 python -m pytest tests/test_gpu_cong.py tests/test_gpu_parity.py -m gpu -q -x -k "builder_writes_it or quantised_knlm or cong_tokens_bit_exact" > $OUT/pytest_gpu_model_files.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu_model_files.txt
timeout 150 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; show $OUT/bench_c2.json c2
timeout 200 python bench.py --workload c4-cong --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c4_cong.json 2> $OUT/bench_c4_cong.err; show $OUT/bench_c4_cong.json c4-cong
