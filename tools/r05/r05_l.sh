#!/bin/bash
# round 5, call l: the global CoNgram model (window 7) on the MI355X -- golden analyses of the real reference, 2800 + 2800 random sentences vs the oracle (top-1 / top-3),
# long sentences, typo correction, kiwi_init's CONG_GLOBAL / LARGEST; the local-model suite beside it; then the c4-cong-global workload's device-resident rate
mkdir -p gpurun_out/r05_l; O=$PWD/gpurun_out/r05_l
timeout 1500 python -m pytest tests/test_gpu_cong_global.py tests/test_zzz_gpu_cong_global_probe.py tests/test_gpu_cong.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_cong_global.txt
KAMD_ARENA_STATS=1 timeout 600 python - > $O/side_c4_cong_global.json 2> $O/side_c4_cong_global.err <<'PY'
import json, bench
print(json.dumps(bench.side_measurement(None, "c4-cong-global", steps=3)))
PY
tail -c 1500 $O/side_c4_cong_global.json; tail -5 $O/side_c4_cong_global.err
