#!/bin/bash
# round 5, call z: the default bench line on the round's final build (live-path compaction in the general search kernel came after call x)
mkdir -p gpurun_out/r05_z; O=$PWD/gpurun_out/r05_z
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json; tail -2 $O/bench_default.err
