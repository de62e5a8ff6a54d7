#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s4
mkdir -p $OUT
cd $R
CFG="sd21:96" KEYSET="0" timeout 900 python tools/grad_probe2.py > $OUT/grad_probe3.log 2>&1
cat $OUT/grad_probe3.log | tail -30
