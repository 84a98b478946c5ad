#!/bin/bash
# First GPU contact of the SkipBigram search kernel (kiwi_amd/csrc/viterbi_kernel_sbg.hip; bit-exact under lane emulation, never run
# on hardware when this was written -- DESIGN.md section 4).  Every step under its own timeout so that a hang costs seconds, not the box.
# usage (on the GPU box, e.g. through gpurun):  tools/sbg_bringup.sh <tag>
TAG=${1:-sbg}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export KAMD_EXPERIMENTAL_SBG=1
# 1. one short text, 16-lane groups; if this hangs, step 1b shows where every chunk stopped (host-side, non-intrusive)
timeout 60 python - > $OUT/first_contact.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from dataclasses import astuple
from kiwi_amd.api import KiwiAmd
from kiwi_amd.synth import SynthModel, SMALL_SBG_SPEC
from corpora import synthetic
import oraclelib
sm = SynthModel(SMALL_SBG_SPEC); os.makedirs("_data", exist_ok=True); path = "_data/small-sbg.raw"; sm.raw.save(path)
dev, orc = KiwiAmd(path), oraclelib.OracleKiwi(path)
norm = lambda res: [([astuple(t) for t in a[0]], a[1]) for a in res]
texts = synthetic(sm, 64, 301, min_jamo=5, max_jamo=60)
for top_n in (1, 3):
    got = dev.analyze_batch(texts, top_n=top_n).to_python()
    print("top", top_n, "mismatches", sum(norm(orc.analyze(s, top_n=top_n)) != norm(y) for s, y in zip(texts, got)), "of", len(texts), flush=True)
PY
echo "first contact: rc=$?"; tail -3 $OUT/first_contact.txt
if ! grep -q "mismatches" $OUT/first_contact.txt; then
  KAMD_HANGDUMP=1 timeout 60 python tools/quick_gpu.py 40 > $OUT/hangdump.txt 2>&1; tail -20 $OUT/hangdump.txt
  exit 1
fi
# 2. the gated parity suite
timeout 600 python -m pytest tests/test_gpu_sbg.py -m gpu -x -q > $OUT/pytest_gpu_sbg.txt 2>&1; tail -3 $OUT/pytest_gpu_sbg.txt
# 3. BASELINE config 3 proper (SkipBigram, top-3, 64k mixed sentences): a first number, however slow
timeout 900 python bench.py --workload c3-sbg --steps 3 --warmup 1 > $OUT/bench_c3_sbg.json 2> $OUT/bench_c3_sbg.err; cut -c1-600 $OUT/bench_c3_sbg.json
# 4. typo correction on the device (same status as the SkipBigram kernel: identical to the oracle under lane emulation, first time on hardware)
KAMD_EXPERIMENTAL_TYPO=1 timeout 600 python -m pytest tests/test_gpu_typo.py -m gpu -x -q > $OUT/pytest_gpu_typo.txt 2>&1; tail -3 $OUT/pytest_gpu_typo.txt
KAMD_EXPERIMENTAL_TYPO=1 timeout 600 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-600 $OUT/bench_c5.json
