#!/bin/bash
# round 4: sub-batches again (lattice stage of sub-batch k+1 on stream A under the search of sub-batch k on stream B) now that the lattice kernel is
# latency-bound instead of scalar-issue-bound; the launch timeline of one c4-cong step
mkdir -p gpurun_out/r04_o; O=$PWD/gpurun_out/r04_o; ROOT=$PWD
for WL in c2-64k c4-cong; do for S in 1 2 3 4; do
  echo "== $WL KAMD_SUBBATCHES=$S"; KAMD_SUBBATCHES=$S timeout 300 python bench.py --workload $WL --kernels-only --steps $([ $WL = c2-64k ] && echo 60 || echo 10) 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' | head -2
done; done 2>&1 | tee $O/subbatches.txt
KAMD_SUBBATCHES=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullmodel.py -m gpu -x -q -k "tokens_bit or whole_corpora or top_n" 2>&1 | tail -3 | tee $O/pytest_subbatches2.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $ROOT/bench.py --workload c4-cong --steps 3 --warmup 2 --kernels-only > $O/trace_c4-cong.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - $f > $O/timeline_c4_cong.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_dict_scan" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kamd::", "")[:40]
    print("%-42s start %8.1f us  dur %8.1f us  end %8.1f us  grid %s lds %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r.get("Grid_Size_X", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?"))))
PY
rm -rf $O/trace
cat $O/timeline_c4_cong.txt | cut -c1-150
