#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s11
mkdir -p $OUT
cd $R
timeout 200 python tools/attn_quick.py > $OUT/attn_ab2.log 2>&1
grep fwd $OUT/attn_ab2.log | cut -c1-240
