#!/bin/bash
# Round-2 final GPU call: whole GPU suite, smoke(), the default bench line, rocprofv3 kernel statistics + HBM traffic counters of the same command,
# and the c4-cong / c5 lines with their CPU baselines.
TAG=${1:-r02p}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', d['value'], d['ms_per_step'], d['config']['kernel_ms'], d.get('e2e',{}).get('value'), d.get('cpu_baseline',{}).get('value'), d.get('roofline',{}).get('frac'))"; }
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"; show $OUT/bench_c2.json c2
timeout 400 python bench.py --workload c4-cong --steps 5 --warmup 1 > $OUT/bench_c4_cong.json 2> $OUT/bench_c4_cong.err; show $OUT/bench_c4_cong.json c4-cong
timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; show $OUT/bench_c5.json c5
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c2.csv 2>/dev/null; head -8 $OUT/kernel_stats_c2.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
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
json.dump(summ, open(out + "/pmc_summary_c2.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
