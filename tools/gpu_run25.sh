#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python tools/knn_ablate.py 512000 2>&1 | grep lib
timeout 600 python tools/knn_perf.py 1000000 2>&1 | grep tflops
