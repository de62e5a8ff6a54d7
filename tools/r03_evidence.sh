#!/bin/bash
# GPU-box tool (round 3): rocprofv3 kernel-trace stats + the bench JSON line of every workload DESIGN.md quotes, and the
# HBM-traffic PMC passes of the default command.  Everything lands under gpurun_out/r03/ (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { # tag, bench args...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python $R/bench.py "$@" > $OUT/$tag.log 2>&1
  grep '^{' $OUT/$tag.log | tail -1 > $OUT/$tag.json
  f=$(find $OUT/$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/$tag
  cut -c1-200 $OUT/$tag.json
}
prof r03a_bench_4layouts --steps 1 --warmup 1 --no-cpu-baseline
prof r03_lmd_v0.1_100prompts --workload lmd_v0.1 --prompts 100 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
prof r03_backward_guidance_sd21 --workload backward_guidance --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
prof r03_bench_4layouts_sam --sam --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
TAG=r03/traffic bash $R/tools/bench_traffic.sh > $OUT/traffic.log 2>&1
cp $R/gpurun_out/r03/traffic/r02_bench_traffic_pmc.json $OUT/r03_bench_traffic_pmc.json 2>/dev/null
rm -rf $R/gpurun_out/r03/traffic/FETCH_SIZE $R/gpurun_out/r03/traffic/WRITE_SIZE
ls -la $OUT
