"""GPU probe: wall time of the headline fit (UMAP N = 1M D = 128 k = 30, 1000 iterations) without bench.py's context legs.

    python tools/fit_time.py [N] [reps]        env LIBPATH=<scratch .so>, RELABEL=0/1, GEOM=<int>, PREFETCH=0/1, EIGH=jacobi/library override the module defaults
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import torchdr_amd
from tests.conftest import gmm
from torchdr_amd.neighbor_embedding import umap as U

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if "RELABEL" in os.environ:
    U.RELABEL = os.environ["RELABEL"] == "1"
if "GEOM" in os.environ:
    U.SCHED_GEOM = int(os.environ["GEOM"])
if "NEGATIVES" in os.environ:
    U.NEGATIVES = os.environ["NEGATIVES"]
if "POOL_GEOM" in os.environ:
    U.POOL_GEOM = int(os.environ["POOL_GEOM"])
if "SCHED_STAGE" in os.environ:
    U.SCHED_STAGE = int(os.environ["SCHED_STAGE"])
from torchdr_amd import affinity_matcher as AM

if "PREFETCH" in os.environ:
    AM.PCA_PREFETCH = os.environ["PREFETCH"] == "1"
if "EIGH" in os.environ:
    AM.PCA_EIGH = os.environ["EIGH"]
if "LIBPATH" in os.environ:      # a scratch build of the library (measurement switches), path relative to the repo root
    from torchdr_amd import _lib as _L
    _L.LIB_PATH = os.path.join(ROOT, os.environ["LIBPATH"])
X = gmm(n, 128, 2.0).cuda()
ts = []
for r in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = torchdr_amd.UMAP(n_neighbors=30, max_iter=1000, random_state=r)
    Z = m.fit_transform(X)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"n": n, "negatives": U.NEGATIVES, "sched_stage": U.SCHED_STAGE, "pool_geom": U.POOL_GEOM, "prefetch": AM.PCA_PREFETCH, "eigh": AM.PCA_EIGH, "relabel": U.RELABEL, "geom": U.SCHED_GEOM, "relabelled": m.loop_order_ is not None,
                  "ms_per_fit": ts[1:], "finite": bool(torch.isfinite(Z).all())}))
