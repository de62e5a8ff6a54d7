#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s19
mkdir -p $OUT
cd $R
cp llm-groundeddiffusion_amd/tuning_gfx950.json $OUT/tuning_gfx950_lanes_full.json
LGD_TUNE_STREAMS=4 LGD_TUNE_TOP=${TOP:-300} timeout 700 python tools/tune_gemm.py sd14_gligen $OUT/tuning_gfx950_lanes_full.json > $OUT/tune.log 2>&1 || true
tail -3 $OUT/tune.log
cp $OUT/tuning_gfx950_lanes_full.json llm-groundeddiffusion_amd/tuning_gfx950_lanes.json
LGD_TUNING_MODE=latency timeout 250 python bench.py --steps 8 --warmup 1 --lanes 4 --no-cpu-baseline --no-roofline > $OUT/l4_latency.log 2>&1; tail -n 1 $OUT/l4_latency.log | cut -c1-200
LGD_TUNING_MODE=throughput timeout 250 python bench.py --steps 8 --warmup 1 --lanes 4 --no-cpu-baseline --no-roofline > $OUT/l4_throughput.log 2>&1; tail -n 1 $OUT/l4_throughput.log | cut -c1-200
LGD_TUNING_MODE=throughput timeout 250 python bench.py --steps 2 --warmup 1 --lanes 1 --no-cpu-baseline --no-roofline > $OUT/l1_throughput.log 2>&1; tail -n 1 $OUT/l1_throughput.log | cut -c1-200
timeout 300 python -m pytest tests/test_bench_path_gpu.py -q -k tuned_gemm_mode > $OUT/pytest_modes.log 2>&1; tail -n 2 $OUT/pytest_modes.log
