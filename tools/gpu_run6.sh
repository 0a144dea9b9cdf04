#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python tools/knn_ablate.py 400000 > gpurun_out/ablate.log 2>&1; cat gpurun_out/ablate.log | grep lib
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_1m.log 2>&1; tail -1 gpurun_out/bench_1m.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','knn_build_sec')}, d['roofline']['achieved'])"
