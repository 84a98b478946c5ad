#!/bin/bash
# Builds everything (product library, its small-capacity build, oracle, emulator) and only then sends the tree to the GPU box: a stale
# libkiwi_hip_smallcaps.so costs a GPU call.  usage: tools/gpu_call.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/gpu_call_build.log 2>&1 || { tail -20 /tmp/gpu_call_build.log; exit 1; }
EX=""; for p in $(cat .gpurunignore); do EX="$EX --exclude=./$p"; done      # what gpurun leaves behind
sz=$(du -sm --exclude=./.git --exclude=./gpurun_out $EX . | cut -f1); echo "tree: ${sz} MiB"
[ "$sz" -lt 505 ] || { echo "tree too large for the 512 MiB snapshot limit"; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
