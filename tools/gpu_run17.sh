#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_knn_gpu.py tests/test_distributed_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/knn_perf.py 100000 300000 1000000 2>&1 | grep tflops
