#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s2
mkdir -p $OUT
cd $R
timeout 120 ./llm-groundeddiffusion_amd/build/ubench > $OUT/ubench.log 2>&1
head -5 $OUT/ubench.log
timeout 900 python -m pytest tests -m gpu -q -rP > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
