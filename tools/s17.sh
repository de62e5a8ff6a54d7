#!/bin/bash
mkdir -p gpurun_out/s17
for L in 6 8; do
timeout 250 python bench.py --steps $((2*L)) --warmup 1 --lanes $L --no-cpu-baseline --no-roofline > gpurun_out/s17/lanes$L.log 2>&1; tail -n 1 gpurun_out/s17/lanes$L.log | cut -c1-230
done
timeout 250 python bench.py --workload backward_guidance --steps 8 --warmup 1 --lanes 4 --no-cpu-baseline --no-roofline > gpurun_out/s17/bg4.log 2>&1; tail -n 1 gpurun_out/s17/bg4.log | cut -c1-230
