#!/bin/bash
# GPU-box tool: cost of each ingredient of gemm_pipe_kernel by removal (library built with -DLGD_GEMM_ABLATION)
for abl in ${ABLS:-0 1 2 3 4 8 11}; do
  echo "== ABL=$abl (1: no DMA, 2: no LDS reads, 4: no MFMA, 8: no barrier, 16: no epilogue; sums combine)"
  LGD_GEMM_ABL=$abl TILES=${TILES:-33,34} SHAPES=${SHAPES:-conv} ROUNDS=3 timeout 120 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | head -${NSHAPES:-3}
done
