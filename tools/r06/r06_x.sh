#!/bin/bash
# round 6, measurement round: bench line, rocprofv3 kernel statistics and PMC passes of every workload on the final build (tools/measure_round.sh)
for wl in "$@"; do bash tools/measure_round.sh r06_x $wl > gpurun_out/r06_x_$wl.log 2>&1; tail -12 gpurun_out/r06_x_$wl.log | cut -c1-300; done
