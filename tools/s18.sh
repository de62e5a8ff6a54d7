#!/bin/bash
# is the interpreter lock what saturates 4 lanes?  two processes x 2 lanes vs one process x 4 lanes (s16: 1.63)
mkdir -p gpurun_out/s18
python bench.py --steps 8 --warmup 1 --lanes 2 --no-cpu-baseline --no-roofline > gpurun_out/s18/p2_a.log 2>&1 &
PA=$!
python bench.py --steps 8 --warmup 1 --lanes 2 --no-cpu-baseline --no-roofline > gpurun_out/s18/p2_b.log 2>&1 &
PB=$!
wait $PA; wait $PB
tail -n 1 gpurun_out/s18/p2_a.log | cut -c1-200
tail -n 1 gpurun_out/s18/p2_b.log | cut -c1-200
timeout 250 python bench.py --steps 8 --warmup 1 --lanes 4 --no-cpu-baseline --no-roofline > gpurun_out/s18/l4.log 2>&1; tail -n 1 gpurun_out/s18/l4.log | cut -c1-200
