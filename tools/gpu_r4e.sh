#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_f64_gpu.py tests/test_embed_gpu.py tests/test_affinity_gpu.py -q -k "float64 or sampler or permutation or perm or wider or two_half or sne or pacmap" 2>&1 | tail -40 > gpurun_out/r4e_tests.log; tail -30 gpurun_out/r4e_tests.log
