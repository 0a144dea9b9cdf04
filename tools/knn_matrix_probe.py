"""GPU probe: exact kNN build over a matrix of shapes (bench generator): looks for sizes where the path taken is off."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance.base import PackedPoints

for n in (300_000, 1_000_000):
    for d in (32, 64, 128, 256):
        for scale in (1.0, 2.0, 5.0):
            X = gmm(n, d, scale).cuda()
            for k in (15, 30):
                for rep in range(2):
                    P = PackedPoints(X)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    dbase.knn_packed(P, P, k, "sqeuclidean", True)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                ci = getattr(P, "_cluster_index", None)
                print(json.dumps({"n": n, "d": d, "scale": scale, "k": k, "ms": round(dt * 1e3, 1), "path": dbase.LAST_KNN.get("path"),
                                  "tier": dbase.LAST_KNN.get("tier"), "flagged": dbase.LAST_KNN.get("flagged"),
                                  "balls": None if ci is None else ci.n_clusters}), flush=True)
            del X, P
