#!/bin/bash
# round 4: phases of k_lattice_wave (KAMD_LATTICE_STOP) and its instruction counters
mkdir -p gpurun_out/r04_b; O=$PWD/gpurun_out/r04_b; ROOT=$PWD
timeout 600 python tools/lattice_phases.py c2-64k c4-cong > $O/lattice_phases.txt 2>&1; cat $O/lattice_phases.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_SCA"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -- python $ROOT/bench.py --workload c2-64k --steps 3 --warmup 1 --kernels-only > $O/pmc_$n.log 2>&1
done
python3 - $O <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kamd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("k_")}
json.dump(summ, open(out + "/pmc_summary_c2_64k.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(summ.items()):
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
rm -rf $O/pmc_*/
