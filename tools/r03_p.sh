#!/bin/bash
# round 3, call p: per-key winners by segmented scans; back-off chain walk settled without the general loop (root starts, leaf edges), sparser edge hash
mkdir -p gpurun_out/r03_p; O=$PWD/gpurun_out/r03_p
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 4000 > $O/check_c2.txt 2>&1; echo "rc $?" >> $O/check_c2.txt
tail -2 $O/check_c2.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_typo.py tests/test_gpu_cong.py tests/test_gpu_parity.py -x -q -m gpu -k "pos" > $O/pytest_pos.txt 2>&1; tail -3 $O/pytest_pos.txt
if grep -q "bad 0 /" $O/check_c2.txt; then
  timeout 300 python tools/bench_multi.py c2,c2-64k "default:;wps2:KAMD_WPS=2" 20 > $O/bench_multi.txt 2> $O/bench_multi.err
  cat $O/bench_multi.txt | cut -c1-330
  export TMPDIR=/tmp
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --kernels-only > $O/trace.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c2-64k.csv 2>/dev/null; rm -rf $O/prof
  head -8 $O/kernel_stats_c2-64k.csv | cut -c1-60,150-260
  timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --kernels-only > $O/pmc.log 2>&1
  python3 - $O <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_pos")}
json.dump(summ, open(out + "/pmc_summary_c2-64k.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
  rm -rf $O/pmc
fi
