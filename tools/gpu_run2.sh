#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 1 --warmup 1 --n 200000 --no-cpu-baseline > gpurun_out/bench_200k.log 2>&1; tail -2 gpurun_out/bench_200k.log
