#!/bin/bash
# Round-2 first GPU call: parity of the benchmarked 'full' model, then first hardware contact of the SkipBigram and typo kernels.
# Every step under its own timeout.  usage (through gpurun): tools/r02_bringup.sh <tag>
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests/test_gpu_fullmodel.py -m gpu -q > $OUT/pytest_fullmodel.txt 2>&1; echo "fullmodel rc=$?"; tail -5 $OUT/pytest_fullmodel.txt
export KAMD_EXPERIMENTAL_SBG=1
timeout 90 python - > $OUT/first_contact.txt 2>&1 <<'PY'
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
echo "sbg first contact: rc=$?"; tail -5 $OUT/first_contact.txt
if ! grep -q "mismatches" $OUT/first_contact.txt; then
  KAMD_HANGDUMP=1 timeout 60 python tools/quick_gpu.py 40 > $OUT/hangdump.txt 2>&1; tail -20 $OUT/hangdump.txt
else
  timeout 500 python -m pytest tests/test_gpu_sbg.py -m gpu -q > $OUT/pytest_gpu_sbg.txt 2>&1; echo "sbg rc=$?"; tail -8 $OUT/pytest_gpu_sbg.txt
fi
KAMD_EXPERIMENTAL_TYPO=1 timeout 500 python -m pytest tests/test_gpu_typo.py "tests/test_gpu_capi.py" -m gpu -q > $OUT/pytest_gpu_typo.txt 2>&1; echo "typo rc=$?"; tail -8 $OUT/pytest_gpu_typo.txt
