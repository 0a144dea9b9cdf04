#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/knn_ablate.py 512000 2>&1 | grep lib
timeout 600 python tools/knn_ablate.py 1000000 2>&1 | grep lib
