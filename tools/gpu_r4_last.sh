#!/bin/bash
# last validation of round 4 after the non-temporal hints: schedule / distributed tests, default bench line, kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/last4; mkdir -p $O
timeout 400 python -m pytest tests/test_umap_sched_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 > $O/tests.log; cat $O/tests.log
timeout 400 python bench.py --steps 5 --warmup 1 > $O/bench.log 2>&1; grep "^{" $O/bench.log | cut -c1-260
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-knn-variants --no-configs > $O/prof.log 2>&1
cd $R; f=$(ls -t $O/prof/*/*kernel_stats.csv | head -1); cp "$f" $O/bench_kernel_stats.csv; head -5 $O/bench_kernel_stats.csv | cut -c1-150
