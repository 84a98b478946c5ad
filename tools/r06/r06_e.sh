#!/bin/bash
# round 6, call e: the whole GPU suite on the build with pretokenized spans + the trie edge hash (one pytest process, no -x: every test reports)
mkdir -p gpurun_out/r06_e; O=$PWD/gpurun_out/r06_e
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
