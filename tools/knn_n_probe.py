"""GPU probe: the exact kNN build (k = 30, D = 128, bench generator) over N: path, tier, flagged rows, balls, predicted scan share, ms."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance.base import PackedPoints

for n in [int(a) for a in sys.argv[1:]] or [500_000, 700_000, 1_000_000, 1_400_000, 2_000_000]:
    X = gmm(n, 128, 2.0).cuda()
    for rep in range(2):
        P = PackedPoints(X)
        dbase.PROFILE = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, I = dbase.knn_packed(P, P, 30, "sqeuclidean", True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ci = getattr(P, "_cluster_index", None)
    prof = [(round(a.elapsed_time(b), 2), nq, what) for a, b, nq, what in dbase.PROFILE]
    dbase.PROFILE = None
    out = {"n": n, "ms": round(dt * 1e3, 1), **{a: dbase.LAST_KNN.get(a) for a in ("path", "tier", "flagged", "pruned", "tier_candidates")},
           "balls": None if ci is None else ci.n_clusters, "launches": prof}
    if ci is not None:
        sizes = ci.tiles.float()
        out["tiles_per_ball"] = [round(float(sizes.mean()), 1), int(sizes.max()), int((sizes == 0).sum())]
        out["radius"] = [round(float(ci.radius.median()), 2), round(float(ci.radius.max()), 2)]
    print(json.dumps(out), flush=True)
    del X, P, C, I
