#!/bin/bash
export TMPDIR=/tmp
for dp in 0 1 2; do for ds in 1 2; do echo "DEPHASE $dp SLEEPS $ds"; TDR_KNN_DEPHASE=$dp TDR_KNN_DEPHASE_SLEEPS=$ds timeout 600 python tools/knn_perf.py 512000 2>&1 | grep tflops | cut -c60-140; done; done
