"""Lab: the heavy-tailed regime of tools/knn_regimes.py (3000 separated groups, Zipf(1.1) sizes, N = 1M x 128) under the default
dispatch, without the index refinement, and with pruning forced: time, path, flagged rows, predicted scan share, same rows?"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128, 30
g = torch.Generator().manual_seed(1)
nc = 3000
w = 1.0 / torch.arange(1, nc + 1, dtype=torch.float64) ** 1.1
lab = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
c = torch.randn(nc, d, generator=g) * 2.0
X = (c[lab] + 0.5 * torch.randn(n, d, generator=g)).float().cuda().contiguous()
ref = None
for name, opts in (("default", {}), ("no index refinement", {"REFINE_INDEX": False}), ("no refinement, prune forced", {"REFINE_INDEX": False, "PRUNE_MODE": "force"}),
                   ("refinement, tile bounds forced", {"TILE_BOUNDS": "force"})):
    with config.options(**opts):
        best = 1e9
        for _ in range(2):
            dbase.LAST_KNN.pop("index_refined", None); dbase.LAST_KNN.pop("index_radii", None)
            X2 = X.clone()          # a fresh block: no cached index
            torch.cuda.synchronize(); t0 = time.perf_counter()
            C, I = pairwise_distances(X2, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        L = dbase.LAST_KNN
        same = None if ref is None else bool(torch.equal(ref[0], C) and torch.equal(ref[1], I))
        if ref is None:
            ref = (C, I)
        print(json.dumps({"mode": name, "sec": round(best, 4), "path": L.get("path"), "tier": L.get("tier"), "flagged": L.get("flagged"), "tile_bounds": L.get("tile_bounds"),
                          "share": L.get("predicted_share"), "lists": L.get("lists"), "index_refined": L.get("index_refined"), "index_radii": L.get("index_radii"), "pilot_tau": L.get("pilot_tau"), "same_as_default": same}), flush=True)
