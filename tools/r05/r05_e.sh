#!/bin/bash
# round 5, call e: where does k_pos_path wait?  texture-addresser / vector-L1 counters of the c2-64k search (16- and 8-lane groups)
mkdir -p gpurun_out/r05_e; O=$PWD/gpurun_out/r05_e; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\|TCC\|SQ\|SQC\|GRBM\)_[A-Za-z0-9_]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
for g in 16 8; do
for c in "TA_BUSY_avr TA_BUSY_max TA_BUSY_min GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  KAMD_POS_G=$g timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${g}_$n -- python $ROOT/bench.py --workload c2-64k --steps 3 --warmup 1 --kernels-only > $O/pmc_${g}_$n.log 2>&1 || tail -2 $O/pmc_${g}_$n.log
done
done
python3 - $O <<'PY'
import csv, sys, glob, collections, json, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    g = re.search(r"pmc_(\d+)_", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_pos")}
json.dump(summ, open(out + "/pmc_summary_pos.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $O/pmc_*/
