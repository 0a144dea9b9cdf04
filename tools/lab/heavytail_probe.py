import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances
n, d, k = 1_000_000, 128, 30
g = torch.Generator().manual_seed(1)
nc = 3000
w = 1.0 / torch.arange(1, nc + 1, dtype=torch.float64) ** 1.1
lab = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
c = torch.randn(nc, d, generator=g) * 2.0
X = (c[lab] + 0.5 * torch.randn(n, d, generator=g)).float().cuda().contiguous()
ref = None
for name, opts in (("default", {}), ("prune forced", {"PRUNE_MODE": "force"}), ("tile bounds forced", {"TILE_BOUNDS": "force"}),
                   ("prune forced, sorted lists", {"PRUNE_MODE": "force", "PRUNED_LISTS": "sorted"})):
    with config.options(**opts):
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        L = dbase.LAST_KNN
        same = None if ref is None else bool(torch.equal(ref[0], C) and torch.equal(ref[1], I))
        if ref is None: ref = (C, I)
        print(json.dumps({"mode": name, "sec": round(best, 4), "path": L.get("path"), "tier": L.get("tier"), "flagged": L.get("flagged"), "tile_bounds": L.get("tile_bounds"),
                          "share": L.get("predicted_share"), "lists": L.get("lists"), "same_as_default": same}), flush=True)
