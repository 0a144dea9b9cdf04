#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4f_tests.log; tail -4 gpurun_out/r4f_tests.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-knn-variants > gpurun_out/r4f_bench.log 2>&1; grep "^{" gpurun_out/r4f_bench.log | cut -c1-700
bash tools/pmc_bench.sh r04 > gpurun_out/r4f_pmc.log 2>&1; tail -30 gpurun_out/r4f_pmc.log
