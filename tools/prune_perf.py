"""GPU probe: cost of the cluster index and of the pruned two-stage search (N = 1M by default)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import base as dbase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
X = gmm(n, 128, scale).cuda()


def tic():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(2):
    P = dbase.PackedPoints(X)
    P.screen_image()
    t0 = tic()
    ci = dbase.ClusterIndex(P)
    t1 = tic()
    P._cluster_index = ci
    C, I = dbase.knn_packed(P, P, 30, "sqeuclidean", True)
    t2 = tic()
    C2, I2 = dbase.knn_packed(P, P, 30, "sqeuclidean", True)  # index and mapped image cached
    t3 = tic()
print(json.dumps({"n": n, "scale": scale, "clusters": ci.n_clusters, "n_img": ci.n_img, "index_build_ms": round((t1 - t0) * 1e3, 1),
                  "first_search_ms": round((t2 - t1) * 1e3, 1), "cached_search_ms": round((t3 - t2) * 1e3, 1),
                  "path": dbase.LAST_KNN}))
