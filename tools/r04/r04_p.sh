#!/bin/bash
# round 4: SkipBigram mixture with the eight partner searches in lockstep (sbg_eval.hpp) -- c3-sbg whole corpus, then kernel statistics and instruction
# counters of the SkipBigram search on its first 8192 sentences
mkdir -p gpurun_out/r04_p; O=$PWD/gpurun_out/r04_p; ROOT=$PWD
timeout 300 python -m pytest tests/test_gpu_sbg.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --workload c3-sbg --kernels-only --steps 1 --warmup 1 > $O/bench_c3_sbg.json 2> $O/bench_c3_sbg.err; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"device_bytes": [0-9]*' $O/bench_c3_sbg.json | head -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $ROOT/bench.py --workload c3-sbg --limit 8192 --steps 2 --warmup 1 --kernels-only > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c3_sbg_8k.csv 2>/dev/null; rm -rf $O/trace
head -6 $O/kernel_stats_c3_sbg_8k.csv | cut -c1-60,150-260
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python $ROOT/bench.py --workload c3-sbg --limit 8192 --steps 1 --warmup 1 --kernels-only > $O/pmc_$n.log 2>&1
done
python3 - $O <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kamd::", "").replace("sbgk::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_")}
json.dump(summ, open(out + "/pmc_summary_c3_sbg_8k.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $O/pmc_*/
