#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s7
mkdir -p $OUT
cd $R
cp llm-groundeddiffusion_amd/tuning_gfx950.json $OUT/tuning_before.json
cp llm-groundeddiffusion_amd/tuning_gfx950.json $OUT/tuning_gfx950.json
LGD_TUNE_TOP=170 timeout 900 python tools/tune_gemm.py sd14_gligen $OUT/tuning_gfx950.json > $OUT/tune.log 2>&1 || true
tail -5 $OUT/tune.log
