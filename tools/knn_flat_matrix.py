"""GPU probe: the unpruned exact search away from the shape the threshold scan was tuned on (N = 1M, D = 128) -- wall seconds of
pairwise_distances with FLAT_SCAN on / off (threshold scan / list-keeping kernel), best of 2, and equality of the two results.

    python tools/knn_flat_matrix.py [set] > gpurun_out/knn_flat_matrix.jsonl        set 1 (default): sizes / D / k; set 2: data regimes
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

CASES = [  # (n_db, n_q (0 = self search), D, k, data)
    (200_000, 0, 128, 30, "structureless"),
    (500_000, 0, 128, 30, "structureless"),
    (1_000_000, 0, 32, 30, "structureless"),
    (1_000_000, 0, 64, 30, "structureless"),
    (1_000_000, 0, 256, 30, "structureless"),
    (1_000_000, 0, 128, 5, "uniform"),
    (1_000_000, 0, 128, 100, "uniform"),
    (2_000_000, 0, 128, 15, "uniform"),
    (1_000_000, 200_000, 128, 30, "structureless"),
    (300_000, 0, 100, 30, "uniform"),
]


CASES2 = [  # data regimes at N = 1M, D = 128, k = 30 (pruning off where the index would prune: the unpruned path is what is measured)
    (1_000_000, 0, 128, 30, "gmm0.5"),
    (1_000_000, 0, 128, 30, "gmm0.85"),
    (1_000_000, 0, 128, 30, "gmm1.0"),
    (1_000_000, 0, 128, 30, "gmm6.0"),
    (1_000_000, 0, 128, 30, "lowdim8"),
    (1_000_000, 0, 128, 30, "sorted2.0"),
    (1_000_000, 0, 16, 30, "uniform"),
    (999_983, 0, 128, 15, "uniform"),
    (1_000_000, 40_000, 128, 30, "structureless"),
]
METRIC = {"gmm6.0": "euclidean"}


def make(data, m, d, seed):
    torch.manual_seed(seed)
    if data == "uniform":
        return torch.randn(m, d)
    if data == "structureless":
        return gmm(m, d, 0.0)
    if data.startswith("gmm"):
        return gmm(m, d, float(data[3:]), seed=seed)
    if data.startswith("sorted"):          # rows grouped by class: the strided visiting order is what keeps the buffers from flooding
        X = gmm(m, d, float(data[6:]), seed=seed)
        return X[torch.argsort(X[:, 0])]
    if data.startswith("lowdim"):          # an r-dimensional subspace + small noise: many candidates inside a query's error band
        r = int(data[6:])
        return torch.randn(m, r) @ torch.randn(r, d) + 0.01 * torch.randn(m, d)
    raise ValueError(data)


def run(X, Y, k, flat, metric="sqeuclidean"):
    best, out = None, None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with config.options(FLAT_SCAN=flat, PRUNE_MODE="0" if SET2 else "auto"):
            out = pairwise_distances(X, Y, metric=metric, k=k, exclude_diag=Y is None, return_indices=True)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best, out, {k_: dbase.LAST_KNN.get(k_) for k_ in ("path", "tier", "flat_terms", "flagged", "pruned")}


SET2 = len(sys.argv) > 1 and sys.argv[1] == "2"
for n, nq, d, k, data in (CASES2 if SET2 else CASES):
    X = make(data, n, d, 42).cuda()
    Q = make(data, nq, d, 7).cuda() if nq else None
    a, b = (Q, X) if nq else (X, None)
    metric = METRIC.get(data, "sqeuclidean")
    t1, (c1, i1), info1 = run(a, b, k, True, metric)
    t0, (c0, i0), info0 = run(a, b, k, False, metric)
    print(json.dumps({"n_db": n, "n_q": nq or n, "D": d, "k": k, "data": data, "metric": metric, "threshold_scan_sec": round(t1, 4),
                      "list_kernel_sec": round(t0, 4), "speedup": round(t0 / t1, 3), "equal": bool(torch.equal(c1, c0) and torch.equal(i1, i0)),
                      "threshold_scan": info1, "list_kernel": info0}), flush=True)
    del X, Q, c1, i1, c0, i0
    torch.cuda.empty_cache()
