#!/bin/bash
# GPU-box tool: end-to-end A/B of kernel-variant switches with the unmodified benchmark command (4 lanes), alternating arms.
#   bash tools/ab_bench_options.sh "gn_fused=0" "attn_w4=0"      (each argument = one LGD_OPTIONS string compared with the default)
for rep in 1 2; do
  for opt in "" "$@"; do
    LGD_OPTIONS="$opt" python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>&1 | grep "^{" |
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('LGD_OPTIONS=\"$opt\"', j['value'], 'images/s', j['ms_per_step'], 'ms/step')"
  done
done
