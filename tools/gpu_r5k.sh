#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5k; mkdir -p $O
timeout 600 python tools/knn_flat_lab.py 1000000 scan 1:2,1:12,1:1,1:11,1:2,1:12 > $O/lab.json 2> $O/lab.err; grep scan_ $O/lab.json | cut -c1-160
for w in uniform structureless mixture; do timeout 200 python tools/knn_flat_search.py 1000000 $w 2>&1 | grep sec | tail -1; done
