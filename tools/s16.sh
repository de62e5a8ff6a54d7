#!/bin/bash
mkdir -p gpurun_out/s16
timeout 300 python -m pytest tests/test_lanes_gpu.py -q -rP > gpurun_out/s16/pytest_lanes.log 2>&1; echo "pytest rc=$?"
tail -n 3 gpurun_out/s16/pytest_lanes.log
for L in 2 3 4; do
timeout 200 python bench.py --steps $((2*L)) --warmup 1 --lanes $L --no-cpu-baseline --no-roofline > gpurun_out/s16/lanes$L.log 2>&1; tail -n 1 gpurun_out/s16/lanes$L.log | cut -c1-230
done
