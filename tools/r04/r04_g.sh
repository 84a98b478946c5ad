#!/bin/bash
# round 4: kernel timeline of one step (gaps between the launches of the lattice stage)
mkdir -p gpurun_out/r04_g; O=$PWD/gpurun_out/r04_g; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for WL in c2-64k c4-cong; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $ROOT/bench.py --workload $WL --steps 3 --warmup 2 --kernels-only > $O/trace_$WL.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last k_dict_scan on
idx = max(i for i, r in enumerate(rows) if "k_dict_scan" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")[:40]
    print("%-42s start %8.1f us  dur %8.1f us  gap-after-prev-end %7.1f us  grid %s lds %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?"))))
    prev_end = max(prev_end, e)
PY
rm -rf $O/trace
done
