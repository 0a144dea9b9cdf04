#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_1m.log 2>&1; tail -2 gpurun_out/bench_1m.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_bench -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
