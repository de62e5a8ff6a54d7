#!/bin/bash
# GPU-box tool: rocprofv3 kernel-trace stats + the bench JSON line of the workloads DESIGN.md quotes, and the HBM-traffic
# PMC passes of the default command.  Everything lands under gpurun_out/$R/ (R = round tag, default r04; copied to
# profiles/ by hand):  gpurun --timeout 1500 -- 'bash tools/evidence.sh [r04]'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-r04}
OUT=$ROOT/gpurun_out/$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { # tag, bench args...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python $ROOT/bench.py "$@" > $OUT/$tag.log 2>&1
  grep '^{' $OUT/$tag.log | tail -1 > $OUT/${tag}_bench_line.json
  f=$(find $OUT/$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/$tag
  cut -c1-200 $OUT/${tag}_bench_line.json
}
prof ${R}b_bench_lanes1 --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline
prof ${R}b_bench_4lanes --steps 4 --warmup 1 --no-cpu-baseline --no-roofline
prof ${R}b_backward_guidance_sd21 --workload backward_guidance --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
TAG=$R/traffic bash $ROOT/tools/bench_traffic.sh > $OUT/traffic.log 2>&1
cp $OUT/traffic/bench_traffic_pmc.json $OUT/${R}_bench_traffic_pmc.json 2>/dev/null
rm -rf $OUT/traffic/FETCH_SIZE $OUT/traffic/WRITE_SIZE
ls -la $OUT
