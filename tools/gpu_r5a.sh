#!/bin/bash
# round 5, call A: new tests (threshold scan, IVF pin) + the threshold-scan lab
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_flat_gpu.py tests/test_knn_ivf_gpu.py -x -q 2>&1 | tail -25 > $O/tests.log; tail -8 $O/tests.log
timeout 600 python tools/knn_flat_lab.py 1000000 > $O/lab.json 2> $O/lab.err; cat $O/lab.json | cut -c1-600; tail -5 $O/lab.err
