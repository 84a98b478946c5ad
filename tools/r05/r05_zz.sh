#!/bin/bash
# round 5, last calls: the GPU tests that need no 'full' model, on the final build (first run: C API, dialects, global CoNgram probe, smoke(); second: the built-model tests;
# third: eval_data, exact math, model files)
mkdir -p gpurun_out/r05_zz; O=$PWD/gpurun_out/r05_zz
true # (third run) timeout 150 python -m pytest tests/test_eval_data.py tests/test_exact_math.py tests/test_model_files.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_eval_math_files.txt
# fourth: the typo tests on the 'full' model
timeout 100 python -m pytest tests/test_gpu_zz_fullmodel_typo.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_fullmodel_typo.txt
