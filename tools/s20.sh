#!/bin/bash
# round 3b evidence + validation: full GPU suite, smoke, default bench (4 lanes) with roofline / cpu baseline,
# kernel-trace stats of the lanes run, the prompt-set workload on 4 lanes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s20
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -n 1 $OUT/smoke.log
timeout 400 python bench.py --steps 8 --warmup 2 > $OUT/bench_default.log 2>&1; grep '^{' $OUT/bench_default.log | tail -1 > $OUT/bench_default.json; cut -c1-250 $OUT/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/prof.log 2>&1
grep '^{' $OUT/prof.log | tail -1 > $OUT/prof_bench_line.json; cut -c1-200 $OUT/prof_bench_line.json
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/r03b_bench_4lanes_kernel_stats.csv
rm -rf $OUT/prof
cd $R
timeout 400 python bench.py --workload lmd_v0.1 --prompts 100 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/lmd.log 2>&1; grep '^{' $OUT/lmd.log | tail -1 > $OUT/lmd.json; cut -c1-250 $OUT/lmd.json
