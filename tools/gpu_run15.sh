#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_1m.log 2>&1; tail -1 gpurun_out/bench_1m.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','knn_build_sec')}, d['roofline']['achieved'], d['cpu_baseline']['value'])"
timeout 300 python bench.py --steps 2 --warmup 1 --n 125000 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('125k', {k:d[k] for k in ('value','ms_per_step','knn_build_sec')})"
