#!/bin/bash
# Developer aid: PMC counter passes of the bench workload (one rocprofv3 run per counter group; no tracing flags).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
cd /tmp && export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
i=0
while read -r grp; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 "$@" > $OUT/p$i.log 2>&1
done <<'GRP'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU
GRP
python3 - $OUT <<'PY'
import csv,sys,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-28:]; agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print("==",k)
    for c,v in sorted(agg[k].items()): print(f"  {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
