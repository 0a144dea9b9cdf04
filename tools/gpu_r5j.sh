#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5j; mkdir -p $O
timeout 600 python -m pytest tests/test_distributed_gpu.py -q -x -k "two_rank_sharded_path_on_one_gpu" 2>&1 | grep -v "socket.cpp" | tail -60 > $O/dist.log; grep -n "Error\|error\|assert\|raise\|File" $O/dist.log | tail -30 | cut -c1-220
for w in uniform structureless mixture; do timeout 200 python tools/knn_flat_search.py 1000000 $w 2>&1 | grep sec | tail -1; done
