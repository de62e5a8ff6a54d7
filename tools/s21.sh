#!/bin/bash
mkdir -p gpurun_out/s21
timeout 500 python -m pytest tests/test_sdxl_gpu.py -q -rP > gpurun_out/s21/pytest_sdxl.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^gate|FAILED|Error|error" gpurun_out/s21/pytest_sdxl.log | head -40
