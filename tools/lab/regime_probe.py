"""Lab: what the kNN chooser sees on one regime of tools/knn_regimes.py (python tools/lab/regime_probe.py "integer-valued features")."""
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

name = sys.argv[1]
src = open(os.path.join(ROOT, "tools", "knn_regimes.py")).read()
head = src[: src.index("for name in (")]
ns = {"__name__": "probe", "__file__": os.path.join(ROOT, "tools", "knn_regimes.py")}
sys.argv = [sys.argv[0]]
exec(compile(head, "knn_regimes_head", "exec"), ns)
X = ns["data"](name).float().cuda().contiguous()
for label, opts in (("default", {}), ("prune forced", {"PRUNE_MODE": "force"}), ("tile bounds forced", {"TILE_BOUNDS": "force"})):
    with config.options(**opts):
        best = 1e9
        for _ in range(2):
            for k_ in ("index_refined", "index_radii", "pilot_tau", "predicted_share", "lists"):
                dbase.LAST_KNN.pop(k_, None)
            X2 = X.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pairwise_distances(X2, metric="sqeuclidean", k=30, exclude_diag=True, return_indices=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(json.dumps({"mode": label, "sec": round(best, 4), **{k_: v for k_, v in dbase.LAST_KNN.items() if not torch.is_tensor(v)}}, default=str), flush=True)
