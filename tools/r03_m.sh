#!/bin/bash
# round 3, call m: lane reads by v_readlane, records fetched a step ahead in the two-waves build; instruction counters and HBM traffic at c2-64k;
# PC sampling of the search at c2-64k (line-table build)
mkdir -p gpurun_out/r03_m; O=$PWD/gpurun_out/r03_m
KAMD_POS_STATS=1 KAMD_HANGDUMP=1 timeout 120 python tools/pos_check.py c2 4000 > $O/check_c2.txt 2>&1; echo "rc $?" >> $O/check_c2.txt
tail -3 $O/check_c2.txt | cut -c1-300
if grep -q "bad 0 /" $O/check_c2.txt; then
  KAMD_POS_STATS=1 timeout 200 python tools/bench_multi.py c2,c2-64k "pos:;pos-wps2:KAMD_WPS=2" 20 > $O/bench_multi.txt 2> $O/bench_multi.err
  cat $O/bench_multi.txt | cut -c1-330
  export TMPDIR=/tmp
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --kernels-only > $O/trace.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c2-64k.csv 2>/dev/null; rm -rf $O/prof
  head -8 $O/kernel_stats_c2-64k.csv | cut -c1-60,150-260
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 150 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --kernels-only > $O/pmc_$n.log 2>&1
  done
  python3 - $O <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_")}
json.dump(summ, open(out + "/pmc_summary_c2-64k.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
  rm -rf $O/pmc_*/
  # PC sampling (beta): which source lines the waves of the search kernel are at
  rocprofv3 --list-avail 2>&1 | grep -i -B2 -A12 "pc.sampl" | head -60 > $O/pcs_avail.txt
  for m in "host_trap time 50" "stochastic cycles 1048576"; do
    set -- $m
    KAMD_LIB=$GRAFT_REPO_ROOT/kiwi_amd/libkiwi_hip_lines.so ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d $O/pcs_$1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --kernels-only > $O/pcs_$1.log 2>&1
    echo "pcs $1 rc $?"; tail -3 $O/pcs_$1.log | cut -c1-300
    python3 - $O/pcs_$1 $O/pcs_$1_summary.csv <<'PY'
import csv, sys, glob, collections
src, dst = sys.argv[1], sys.argv[2]
drop = ("timestamp", "exec", "correlation", "dispatch", "wave_in", "workgroup", "chiplet", "hw_id", "sample", "thread", "wave_id", "wave_count", "queue", "agent")
for f in glob.glob(src + "/**/*.csv", recursive=True):
    print("file", f)
    rd = csv.reader(open(f, newline=""))
    try: hdr = next(rd)
    except StopIteration: continue
    print("columns", hdr)
    if "pc_sampl" not in f: continue
    keep = [i for i, h in enumerate(hdr) if not any(d in h.lower() for d in drop)]
    cnt = collections.Counter(); n = 0
    for r in rd:
        if n < 3: print(r)
        n += 1
        cnt[tuple(r[i] for i in keep if i < len(r))] += 1
    print("samples", n, "distinct", len(cnt))
    with open(dst, "w", newline="") as o:
        w = csv.writer(o); w.writerow(["count"] + [hdr[i] for i in keep])
        for k, v in cnt.most_common(): w.writerow([v] + list(k))
PY
    rm -rf $O/pcs_$1
  done
fi
