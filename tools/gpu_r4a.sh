#!/bin/bash
# round 4, call A: grouped schedule build -- tests + perf probe
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_umap_sched_gpu.py -q -k "grouped or group_order" 2>&1 | tail -15 > gpurun_out/r4a_tests.log; tail -5 gpurun_out/r4a_tests.log
timeout 400 python tools/sched_build2_perf.py > gpurun_out/r4a_build2.log 2>&1; grep "^{" gpurun_out/r4a_build2.log | cut -c1-200 || tail -20 gpurun_out/r4a_build2.log
