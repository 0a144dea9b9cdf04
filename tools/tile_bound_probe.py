"""GPU probe: the per-tile bound of the pruned search on data whose balls overlap (centre scale 1.0): ms, path, prediction."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import gmm
from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance.base import PackedPoints

for n, d, scale in [(int(a.split(",")[0]), int(a.split(",")[1]), float(a.split(",")[2])) for a in sys.argv[1:]] or ((300_000, 128, 1.0), (1_000_000, 128, 1.0)):
    X = gmm(n, d, scale).cuda()
    res = {}
    for tb in (False, True, "force"):
        with config.options(TILE_BOUNDS=tb):
            for rep in range(2):
                P = PackedPoints(X)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                C, I = dbase.knn_packed(P, P, 30, "sqeuclidean", True)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[tb] = (C, I)
        ci = getattr(P, "_cluster_index", None)
        print(json.dumps({"n": n, "d": d, "scale": scale, "tile_bounds": tb, "ms": round(dt * 1e3, 1), "path": dbase.LAST_KNN.get("path"),
                          "used": dbase.LAST_KNN.get("tile_bounds"), "tier": dbase.LAST_KNN.get("tier"), "flagged": dbase.LAST_KNN.get("flagged"),
                          "balls": None if ci is None else ci.n_clusters,
                          "pred@1/1.25/1.5/2": None if (ci is None or ci.tile_cdist is None) else [round(ci.scan_fraction_tiles(f * float(C[:, -1].max())), 3) for f in (1.0, 1.25, 1.5, 2.0)]}), flush=True)
    print("equal:", all(torch.equal(res[False][0], res[t][0]) and torch.equal(res[False][1], res[t][1]) for t in (True, "force")), flush=True)
    del X, P, res
