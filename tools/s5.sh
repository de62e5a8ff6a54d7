#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s5
mkdir -p $OUT
cd $R
timeout 120 python tools/attn_dbg.py > $OUT/dbg80.log 2>&1
grep -h "bad rows\|Error" $OUT/dbg80.log | cut -c1-300
