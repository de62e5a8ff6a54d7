#!/bin/bash
# Builds liblgd_hip.so with the GEMM ablation variants compiled in (-DLGD_GEMM_ABLATION); tools only.
set -e
P=/root/repo/llm-groundeddiffusion_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DLGD_GEMM_ABLATION -c $P/csrc/gemm.hip -o $P/build/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/liblgd_hip.so $P/build/gemm.o $P/build/norm.o $P/build/attn.o $P/build/attn_bwd.o $P/build/misc.o $P/build/energy.o
echo "built ablation library"
