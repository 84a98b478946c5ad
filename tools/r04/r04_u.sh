#!/bin/bash
# round 4: measurement rounds of c5 and c4-cong on the final device build (kernel statistics, instruction counters, HBM traffic per kernel -> profiles/traffic.json)
bash tools/measure_round.sh r04_u c5 2>&1 | tail -12 | cut -c1-300
bash tools/measure_round.sh r04_u c4-cong 2>&1 | tail -14 | cut -c1-300
