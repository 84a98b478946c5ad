#!/bin/bash
# round 4: BASELINE config 3 (SkipBigram, 65 536 mixed sentences, top-3) on the re-generated 'full-sbg' model (one reading of a homograph dominates, unknown NNG / NNP readings differ)
mkdir -p gpurun_out/r04_k; O=$PWD/gpurun_out/r04_k
timeout 900 python bench.py --workload c3-sbg --steps 3 --warmup 1 > $O/bench_c3-sbg.json 2> $O/bench_c3-sbg.err; cut -c1-1800 $O/bench_c3-sbg.json; tail -3 $O/bench_c3-sbg.err
timeout 600 python -m pytest tests/test_gpu_fullmodel.py -m gpu -x -q -k "c3_sbg" 2>&1 | tail -3
