#!/bin/bash
# concurrency probe: do two independent bench processes on ONE GPU finish 2x the work in < 2x the time?
mkdir -p gpurun_out/s13
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/s13/single.log 2>&1
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/s13/dual_a.log 2>&1 &
PA=$!
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/s13/dual_b.log 2>&1 &
PB=$!
wait $PA; wait $PB
tail -n 1 gpurun_out/s13/single.log | cut -c1-300
tail -n 1 gpurun_out/s13/dual_a.log | cut -c1-300
tail -n 1 gpurun_out/s13/dual_b.log | cut -c1-300
