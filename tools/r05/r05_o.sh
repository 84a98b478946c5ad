#!/bin/bash
# round 5, call o: what bounds the SkipBigram search?  vector-memory / L1 / L2 / atomic counters of sbgk::k_best_path<64, 2> on the first 8192 sentences of c3-sbg
mkdir -p gpurun_out/r05_o; O=$PWD/gpurun_out/r05_o; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
cat > /tmp/drv.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import bench
d = bench.side_measurement(None, "c3-sbg", steps=1, limit=8192)
print({k: d[k] for k in ("value", "ms_per_step", "kernel_ms")})
PY
for c in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE TA_FLAT_ATOMIC_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCC_ATOMIC_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_FLAT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python /tmp/drv.py $ROOT > $O/pmc_$n.log 2>&1 || tail -2 $O/pmc_$n.log
done
python3 - $O <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: max(v) for c, v in d.items()} for k, d in agg.items() if "k_best_path" in k}     # (max: the 8192-sentence launch, not the 4096-sentence sample before it)
json.dump(summ, open(out + "/pmc_summary_sbg.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $O/pmc_*/
