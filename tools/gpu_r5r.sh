#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5r; mkdir -p $O
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_knn_flat_gpu.py -q -s 2>&1 | grep -v "^$" | tail -25 | cut -c1-250
