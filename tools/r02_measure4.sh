#!/bin/bash
# Round-2 GPU call 5: the whole GPU suite (CoNgram kernel first time on hardware, model-directory loader, C client built against the reference
# header), bench lines for c2 / c5 / c4-cong / c3-sbg (small), rocprofv3 kernel statistics and HBM traffic counters of the c2 command.
TAG=${1:-r02e}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 120 python -m pytest tests/test_gpu_cong.py -m gpu -q -x > $OUT/pytest_gpu_cong.txt 2>&1; echo "cong rc=$?"; tail -6 $OUT/pytest_gpu_cong.txt
timeout 700 python -m pytest tests -m gpu -q --durations=6 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.txt
timeout 300 python bench.py --steps 30 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-300 $OUT/bench_c2.json
timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; python -c "import json;d=json.load(open('$OUT/bench_c5.json'));print('c5', d['value'], d['config']['kernel_ms'], d['cpu_baseline']['value'])"
timeout 400 python bench.py --workload c4-cong --steps 5 --warmup 1 > $OUT/bench_c4_cong.json 2> $OUT/bench_c4_cong.err; python -c "import json;d=json.load(open('$OUT/bench_c4_cong.json'));print('c4-cong', d['value'], d['config']['kernel_ms'], d.get('e2e'), d['cpu_baseline'], d['roofline']['frac'])"; tail -2 $OUT/bench_c4_cong.err
timeout 300 python bench.py --workload c3-sbg --limit 256 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_c3_sbg_256.json 2> $OUT/bench_c3_sbg_256.err; echo "sbg256 rc=$?"; cut -c1-900 $OUT/bench_c3_sbg_256.json
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
