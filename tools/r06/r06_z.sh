#!/bin/bash
# Final measurement of round 6 on the GPU box: the whole GPU suite + smoke, the default bench line (c2-64k with every config.also entry, e2e, kiwi_analyze_m),
# rocprofv3 kernel statistics of the default workload and of c5 (the typo lattice kernel is built for five wavefronts per SIMD since r06_g), c5's bench line.
# The search / lattice kernels are those of r06_x (PMC summaries and traffic.json of that run stay valid: this round's later commits are host-side).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG:-r06_z}; mkdir -p $OUT; cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
timeout 600 python bench.py --workload c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-300 $OUT/bench_c5.json
cd /tmp && export TMPDIR=/tmp
for WL in c2-64k c5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --kernels-only > $OUT/trace_$WL.log 2>&1
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$WL.csv 2>/dev/null; rm -rf $OUT/trace
  head -6 $OUT/kernel_stats_$WL.csv | cut -c1-60,150-260
done
