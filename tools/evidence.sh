#!/bin/bash
# GPU-box tool: rocprofv3 kernel-trace stats + the bench JSON line of the workloads DESIGN.md quotes, the HBM-traffic PMC
# passes of the default command, and the SQ / TCC counters of the 256x128 GEMM family on its heaviest shapes.  Everything
# lands under gpurun_out/$R/ (R = round tag, default r05; copied to profiles/ by hand):
#   gpurun --timeout 1500 -- 'bash tools/evidence.sh [r05]'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-r05}
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
prof ${R}_bench_lanes1 --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline
prof ${R}_bench_driver --steps 4 --warmup 1 --no-cpu-baseline --no-roofline
prof ${R}_lmd_lanes1 --workload lmd --lanes 1 --steps 1 --warmup 0 --no-cpu-baseline
TAG=$R/traffic bash $ROOT/tools/bench_traffic.sh > $OUT/traffic.log 2>&1
cp $OUT/traffic/bench_traffic_pmc.json $OUT/${R}_bench_traffic_pmc.json 2>/dev/null
rm -rf $OUT/traffic/FETCH_SIZE $OUT/traffic/WRITE_SIZE
# the 256x128 family (tile code 34) on its three heaviest benchmark shapes + the 4096x1280x1280 projection, one entry per shape
TAG=$R/pmc_gemm_256x128 TILES=34 SHAPES=geglu,plain PMC_BY_GRID=1 bash $ROOT/tools/pmc_gemm.sh > $OUT/pmc_gemm.log 2>&1
cp $ROOT/gpurun_out/$R/pmc_gemm_256x128/summary.json $OUT/${R}_gemm_256x128_pmc_summary.json 2>/dev/null
find $ROOT/gpurun_out/$R/pmc_gemm_256x128 -name '*.csv' -delete 2>/dev/null
ls -la $OUT
