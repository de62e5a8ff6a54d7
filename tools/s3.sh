#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s3
mkdir -p $OUT
cd $R
timeout 120 ./llm-groundeddiffusion_amd/build/ubench > $OUT/ubench.log 2>&1
head -10 $OUT/ubench.log
timeout 600 python -m pytest tests -m gpu -q -rP -k "teacher or vae or freed or reuses" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 400 python tools/sd21_grad.py > $OUT/sd21_grad.log 2>&1
cat $OUT/sd21_grad.log | tail -12
