#!/bin/bash
# end-of-round validation + evidence: full GPU suite, smoke, default bench, SDXL-refiner bench (+ kernel-trace stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s24
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 700 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -n 2
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 1 $OUT/smoke.log
timeout 400 python bench.py --steps 8 --warmup 2 > $OUT/bench_default.log 2>&1; grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json; cut -c1-200 $OUT/bench_default.json
timeout 500 python bench.py --workload sdxl_refiner --steps 2 --warmup 1 > $OUT/refiner.log 2>&1; grep '^{' $OUT/refiner.log | tail -1 > $OUT/refiner.json; cut -c1-200 $OUT/refiner.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --workload sdxl_refiner --steps 1 --warmup 0 --lanes 2 --no-cpu-baseline --no-roofline > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/r03b_sdxl_refiner_kernel_stats.csv
rm -rf $OUT/prof
grep '^{' $OUT/prof.log | tail -1 | cut -c1-160
