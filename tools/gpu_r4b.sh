#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_build2.sh 768 > gpurun_out/r4b_pmc.log 2>&1; tail -8 gpurun_out/r4b_pmc.log | cut -c1-900
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-knn-variants > gpurun_out/r4b_bench.log 2>&1; grep "^{" gpurun_out/r4b_bench.log | cut -c1-1500
