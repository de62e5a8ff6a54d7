#!/bin/bash
# Builds liblgd_hip.so with the ablation variants compiled in (-DLGD_GEMM_ABLATION, -DLGD_W4_ABLATION); tools only:
# rebuild with `python llm-groundeddiffusion_amd/build.py --force` afterwards.
set -e
P=/root/repo/llm-groundeddiffusion_amd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
/opt/rocm/bin/hipcc $F -DLGD_GEMM_ABLATION -c $P/csrc/gemm.hip -o $P/build/gemm.o &
/opt/rocm/bin/hipcc $F -DLGD_W4_ABLATION -c $P/csrc/attn_w4.hip -o $P/build/attn_w4.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/liblgd_hip.so $P/build/gemm.o $P/build/norm.o $P/build/attn.o $P/build/attn_w4.o $P/build/attn_bwd.o $P/build/misc.o $P/build/energy.o $P/build/boxdiff.o $P/build/sam.o
rm -f $P/liblgd_hip.so.stamp
echo "built ablation library"
