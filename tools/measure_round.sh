#!/bin/bash
# Round measurement on the GPU box: the bench line, rocprofv3 kernel statistics and the PMC passes (HBM traffic, instruction counters) of the
# default bench workload -- each in its own run, counters never together with tracing.
# usage (through gpurun): tools/measure_round.sh <tag> [workload]     writes gpurun_out/<tag>/...; copy what is to be kept into profiles/
TAG=${1:-r01}; WL=${2:-c2-64k}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; cut -c1-600 $OUT/bench_$WL.json
cd /tmp && export TMPDIR=/tmp
timeout ${TO:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --kernels-only > $OUT/trace_$WL.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$WL.csv 2>/dev/null; rm -rf $OUT/trace
head -8 $OUT/kernel_stats_$WL.csv | cut -c1-60,150-260
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout ${TO:-400} rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -- python $ROOT/bench.py --workload $WL --steps 4 --warmup 1 --kernels-only > $OUT/pmc_$n.log 2>&1
done
python3 - $OUT $WL <<'PY'
import csv, sys, glob, collections, json
out, wl = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kamd::", "").replace("typok::", "").replace("congk::", "").replace("gk::", "").replace("sbgk::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_")}
json.dump(summ, open(out + f"/pmc_summary_{wl}.json", "w"), indent=1, sort_keys=True)
# profiles/traffic.json: HBM bytes per launch from FETCH_SIZE + WRITE_SIZE (KiB; raw values -- the accesses are scattered 16-64 B ones, see profiles/README.md)
tr = {"workload": wl, "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of `python bench.py --workload {wl} --steps 4 --warmup 1 --kernels-only`, no tracing flags (tools/measure_round.sh); counters in KiB, raw",
      "kernels": {k: {"fetch_size_kb": d.get("FETCH_SIZE"), "write_size_kb": d.get("WRITE_SIZE"), "hbm_bytes_per_launch": 1024.0 * (d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0))} for k, d in summ.items() if "FETCH_SIZE" in d}}
json.dump(tr, open(out + f"/traffic_{wl}.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $OUT/pmc_*/
