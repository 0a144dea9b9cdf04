#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5q; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_flat_gpu.py -q -s 2>&1 | grep -v "^$" | tail -12 | cut -c1-220
for w in uniform structureless mixture; do timeout 200 python tools/knn_flat_search.py 1000000 $w 2>&1 | grep sec | tail -1; done
bash tools/pmc_configs_r5.sh 2>&1 | tail -30 | cut -c1-700
