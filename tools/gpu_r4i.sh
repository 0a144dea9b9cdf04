#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_distributed_gpu.py -q -x 2>&1 | tail -30 > gpurun_out/r4i_tests.log; tail -25 gpurun_out/r4i_tests.log
