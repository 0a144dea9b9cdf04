"""GPU probe: which screening tier / path the exact search takes for several k at the headline size, and what it costs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = gmm(n, 128, 2.0).cuda()
for k in (15, 30, 45, 90):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"k": k, "sec": round(dt, 4), **{a: dbase.LAST_KNN.get(a) for a in ("path", "tier", "flagged", "pruned")}}), flush=True)
