#!/bin/bash
# round 4: phases of k_lattice_wave (KAMD_LATTICE_STOP), quick parity check, instruction counters
mkdir -p gpurun_out/r04_c; O=$PWD/gpurun_out/r04_c; ROOT=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lattices or tokens_bit or fuzzed" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/lattice_phases.py c2-64k c4-cong > $O/lattice_phases.txt 2>&1; cat $O/lattice_phases.txt
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
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
for k, d in sorted(agg.items()):
    if k.startswith("k_lattice") or k.startswith("k_expand"): print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
rm -rf $O/pmc_*/
