"""The exact kNN build across shapes and data regimes: the default dispatch of round 6 (lazy pilots; the pruned scan with lazy
candidate buffers where the predicted scan share is <= distance/base.py:_LAZY_MAX_SHARE, sorted lists above it; PRUNED_LISTS=lazy
in the environment forces the buffers everywhere: the calibration run) against the sorted lists of rounds 2-5 everywhere
(tdr_knn_screen_clustered_lists(0): also the pilots' form) -- one JSON line per case: path, tier, flagged rows, predicted share,
list form taken, best-of-3 wall time of pairwise_distances(k), and whether both returned the same rows bit for bit.

    python tools/knn_lists_matrix.py > profiles/r06_knn_lists_matrix.jsonl
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from torchdr_amd import _lib
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

L = _lib.lib()


def heavy_tail(n, d, seed=7):
    g = torch.Generator().manual_seed(seed)
    nc = 3000
    w = 1.0 / torch.arange(1, nc + 1, dtype=torch.float64) ** 1.1
    lab = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
    c = torch.randn(nc, d, generator=g) * 2.0
    return c[lab] + 0.5 * torch.randn(n, d, generator=g)


CASES = [
    ("headline mixture", lambda: gmm(1_000_000, 128, 2.0), 30),
    ("mixture, k = 15", lambda: gmm(1_000_000, 128, 2.0), 15),
    ("mixture, k = 60", lambda: gmm(1_000_000, 128, 2.0), 60),
    ("mixture, centre scale 1.6", lambda: gmm(1_000_000, 128, 1.6), 30),
    ("mixture, centre scale 1.3 (tile bounds)", lambda: gmm(1_000_000, 128, 1.3), 30),
    ("mixture, centre scale 1.15 (tile bounds)", lambda: gmm(1_000_000, 128, 1.15), 30),
    ("mixture, centre scale 1.0 (tile bounds)", lambda: gmm(1_000_000, 128, 1.0), 30),
    ("mixture, D = 64", lambda: gmm(1_000_000, 64, 2.0), 30),
    ("mixture, D = 64, centre scale 1.0 (tile bounds)", lambda: gmm(1_000_000, 64, 1.0), 30),
    ("mixture, D = 256, k = 15", lambda: gmm(1_000_000, 256, 2.0), 15),
    ("mixture, D = 256, centre scale 5, k = 15", lambda: gmm(1_000_000, 256, 5.0), 15),
    ("mixture, N = 300k", lambda: gmm(300_000, 128, 2.0), 30),
    ("mixture, N = 700k", lambda: gmm(700_000, 128, 2.0), 30),
    ("mixture, N = 2M", lambda: gmm(2_000_000, 128, 2.0), 30),
    ("heavy-tailed cluster sizes", lambda: heavy_tail(1_000_000, 128), 30),
]

only = os.environ.get("CASES")
for name, make, k in CASES:
    if only and only not in name:
        continue
    X = make().float().cuda().contiguous()
    rec = {"case": name, "n": int(X.shape[0]), "d": int(X.shape[1]), "k": k}
    outs = {}
    if os.environ.get("PRUNED_LISTS"):
        dbase.PRUNED_LISTS = os.environ["PRUNED_LISTS"]
    for mode, label in ((1, "default_dispatch"), (0, "sorted_lists")):
        prev = L.tdr_knn_screen_clustered_lists(mode)
        try:
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            LK = dbase.LAST_KNN
            rec[label] = {"ms": round(best * 1e3, 2), "path": LK.get("path"), "tier": LK.get("tier"), "tile_bounds": LK.get("tile_bounds"),
                          "flagged_rows": LK.get("flagged"), "predicted_share": LK.get("predicted_share"), "lists": LK.get("lists")}
            outs[mode] = (C, I)
        finally:
            L.tdr_knn_screen_clustered_lists(prev)
    rec["same_rows_bit_for_bit"] = bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
    rec["speedup"] = round(rec["sorted_lists"]["ms"] / rec["default_dispatch"]["ms"], 2)
    print(json.dumps(rec), flush=True)
    del X, outs, C, I
    torch.cuda.empty_cache()
