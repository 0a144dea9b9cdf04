#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5x; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_flat_gpu.py tests/test_knn_screen_gpu.py -q 2>&1 | tail -5 | cut -c1-220
for w in uniform structureless mixture; do timeout 200 python tools/knn_flat_search.py 1000000 $w 2>&1 | grep sec | tail -1; done
python - <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from tests.conftest import gmm
from torchdr_amd.distance import pairwise_distances, base as dbase
X = gmm(1_000_000, 128, 1.0).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pairwise_distances(X, metric="sqeuclidean", k=30, exclude_diag=True, return_indices=True)
    torch.cuda.synchronize(); print({"scale1.0_sec": time.perf_counter() - t0, **{k: dbase.LAST_KNN.get(k) for k in ("path", "tile_bounds", "pruned", "flat_terms")}})
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_uniform -- python $R/tools/knn_flat_search.py 1000000 uniform > $O/trace_uniform.log 2>&1
f=$(ls -t $O/trace_uniform/*/*kernel_stats.csv | head -1); head -9 "$f" | cut -c1-160
