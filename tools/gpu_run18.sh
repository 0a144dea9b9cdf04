#!/bin/bash
export TMPDIR=/tmp
for nw in 4 6 8; do echo "NW $nw"; TDR_KNN_NW=$nw timeout 900 python -m pytest tests/test_knn_gpu.py -m gpu -q -x 2>&1 | tail -1; TDR_KNN_NW=$nw timeout 600 python tools/knn_perf.py 400000 1000000 2>&1 | grep tflops | cut -c1-200; done
