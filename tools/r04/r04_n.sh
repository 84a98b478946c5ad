#!/bin/bash
# round 4: the frequency-based unknown-form modes on the MI355X (k_unk_chr_freq, tanhf on the device), the capi config round trip
mkdir -p gpurun_out/r04_n; O=$PWD/gpurun_out/r04_n
timeout 900 python -m pytest tests/test_exact_math.py tests/test_gpu_cong.py tests/test_gpu_capi.py -m gpu -x -q --durations=5 -k "exact_math or character_model or substring_frequencies or config_roundtrip" > $O/pytest_f4.txt 2>&1; echo "rc $?" >> $O/pytest_f4.txt
tail -12 $O/pytest_f4.txt | cut -c1-300
