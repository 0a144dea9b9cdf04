#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
TDR_AUDIT_ONLY=1 timeout 900 python -m pytest tests/test_embed_gpu.py tests/test_tsnekhorn_gpu.py tests/test_pacmap_gpu.py tests/test_configs_gpu.py -q  2>&1 | tail -25 > gpurun_out/r4d_tests.log; tail -12 gpurun_out/r4d_tests.log
cat gpurun_out/tolerance_audit.json
