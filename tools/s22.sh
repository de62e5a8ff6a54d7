#!/bin/bash
mkdir -p gpurun_out/s22
timeout 300 python bench.py --workload sdxl_refiner --config tiny_xl --steps 1 --warmup 0 --lanes 2 --layouts 2 --no-cpu-baseline > gpurun_out/s22/tiny.log 2>&1; tail -n 1 gpurun_out/s22/tiny.log | cut -c1-400
timeout 800 python bench.py --workload sdxl_refiner --steps 2 --warmup 1 --lanes 2 --no-cpu-baseline > gpurun_out/s22/refiner.log 2>&1; tail -n 3 gpurun_out/s22/refiner.log | cut -c1-1500
