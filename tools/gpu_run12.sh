#!/bin/bash
export TMPDIR=/tmp
for g in 0 1 2; do echo "GEOM $g"; TDR_UMAP_GEOM=$g timeout 600 python -m pytest tests/test_embed_gpu.py -m gpu -q -x -k "umap_three" 2>&1 | tail -1; TDR_UMAP_GEOM=$g timeout 600 python tools/umap_perf.py 2>&1 | grep -E "default|no_neg|random"; done
