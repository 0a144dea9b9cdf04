#!/bin/bash
# first GPU round: kNN parity + perf probe + rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -6 > gpurun_out/rocminfo.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python tools/knn_perf.py 100000 1000000 > gpurun_out/knn_perf.log 2>&1
echo "perf exit $?" >> gpurun_out/knn_perf.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_knn -- python $GRAFT_REPO_ROOT/tools/knn_perf.py 200000 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_knn -name "*stats*" | head; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/knn_perf.log
