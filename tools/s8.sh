#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s8
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -rP > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-600
