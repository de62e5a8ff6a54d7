#!/bin/bash
mkdir -p gpurun_out/s14
timeout 300 python -m pytest tests/test_lanes_gpu.py -x -q -rP > gpurun_out/s14/pytest_lanes.log 2>&1; echo "pytest rc=$?"
tail -n 5 gpurun_out/s14/pytest_lanes.log
timeout 200 python bench.py --steps 4 --warmup 1 --lanes 1 --no-cpu-baseline --no-roofline > gpurun_out/s14/lanes1.log 2>&1; tail -n 1 gpurun_out/s14/lanes1.log | cut -c1-260
timeout 200 python bench.py --steps 4 --warmup 1 --lanes 2 --no-cpu-baseline --no-roofline > gpurun_out/s14/lanes2.log 2>&1; tail -n 1 gpurun_out/s14/lanes2.log | cut -c1-260
timeout 200 python bench.py --steps 6 --warmup 1 --lanes 3 --no-cpu-baseline --no-roofline > gpurun_out/s14/lanes3.log 2>&1; tail -n 1 gpurun_out/s14/lanes3.log | cut -c1-260
