#!/bin/bash
# round 3, call g: reachability bit sets in LDS; instruction counters of k_pos_path (rocprofv3 --pmc, its own run), c2
mkdir -p gpurun_out/r03_g; O=$PWD/gpurun_out/r03_g
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/quick_gpu.py 300 > $O/check_small.txt 2>&1; echo "rc $?" >> $O/check_small.txt; tail -3 $O/check_small.txt
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 4000 > $O/check_c2.txt 2>&1; echo "rc $?" >> $O/check_c2.txt; tail -3 $O/check_c2.txt
if grep -q "bad 0 /" $O/check_c2.txt; then
  export TMPDIR=/tmp; cd /tmp
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc -- python $GRAFT_REPO_ROOT/tools/bench_multi.py c2 "pos:" 4 > $O/pmc_bench.txt 2> $O/pmc_bench.err
  python3 - $O <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_")}
json.dump(summ, open(out + "/pmc_summary_c2.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
  rm -rf $O/pmc
  cd $GRAFT_REPO_ROOT
  timeout 200 python tools/bench_multi.py c2,c2-64k "pos:" 20 > $O/bench_multi.txt 2> $O/bench_multi.err; cut -c1-330 $O/bench_multi.txt
fi
