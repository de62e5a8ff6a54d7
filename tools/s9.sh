#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s9
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -rP -k "sd21_guidance or overall_stage or self32 or forced_rescale" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log; grep "\[gate\].*run b\|\[gate\].*sd21" $OUT/pytest.log | cut -c1-200
