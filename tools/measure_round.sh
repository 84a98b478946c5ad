#!/bin/bash
# Round measurement on the GPU box: parity tests, bench line, rocprofv3 kernel statistics, HBM traffic counters.
# usage: tools/measure_round.sh <tag>      (writes gpurun_out/<tag>/...; copy what is to be kept into profiles/)
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -2 $OUT/pytest_gpu.txt
timeout 300 python bench.py --steps 30 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# PMC passes: separate runs, no tracing flags
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_$n.log 2>&1
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_")}
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
