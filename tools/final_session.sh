bash tools/gpu_session.sh tests
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2>&1; grep "^{" gpurun_out/bench_driver.log | tail -1 > gpurun_out/r04c_bench_driver_command_bench_line.json
python bench.py > gpurun_out/bench_default.log 2>&1; grep "^{" gpurun_out/bench_default.log | tail -1 > gpurun_out/r04c_bench_default_bench_line.json
python bench.py --workload lmd_v0.1 --prompts 100 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/bench_lmd.log 2>&1; grep "^{" gpurun_out/bench_lmd.log | tail -1 > gpurun_out/r04c_lmd_v0.1_100prompts_bench_line.json
python bench.py --workload sdxl_refiner --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_sdxl.log 2>&1; grep "^{" gpurun_out/bench_sdxl.log | tail -1 > gpurun_out/r04c_sdxl_refiner_bench_line.json
python bench.py --workload backward_guidance --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_bg.log 2>&1; grep "^{" gpurun_out/bench_bg.log | tail -1 > gpurun_out/r04c_backward_guidance_sd21_bench_line.json
for f in gpurun_out/r04c_*; do echo $f; cut -c1-160 $f; done
