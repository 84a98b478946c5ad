#!/bin/bash
# A GPU call with a smaller snapshot: the paths named after the command are added to .gpurunignore for the duration of the call (the push of the full
# 410 MiB tree is charged: 90 - 130 s).  usage: tools/r05/gpu_lite.sh <timeout-seconds> '<command>' [path ...]
cd "$(dirname "$0")/../.."
T=$1; CMD=$2; shift 2
cp .gpurunignore /tmp/gpurunignore.saved
for p in "$@"; do echo "$p" >> .gpurunignore; done
/usr/local/graft/bin/gpurun --timeout "$T" -- "$CMD"; rc=$?
cp /tmp/gpurunignore.saved .gpurunignore
exit $rc
