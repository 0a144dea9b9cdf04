"""GPU probe: the unpruned exact search away from the shape the threshold scan was tuned on (N = 1M, D = 128) -- wall seconds of
pairwise_distances with FLAT_SCAN on / off (threshold scan / list-keeping kernel), best of 2, and equality of the two results.

    python tools/knn_flat_matrix.py > gpurun_out/knn_flat_matrix.jsonl
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


def run(X, Y, k, flat):
    best, out = None, None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with config.options(FLAT_SCAN=flat):
            out = pairwise_distances(X, Y, metric="sqeuclidean", k=k, exclude_diag=Y is None, return_indices=True)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    return best, out, {k_: dbase.LAST_KNN.get(k_) for k_ in ("path", "tier", "flat_terms", "flagged", "pruned")}


for n, nq, d, k, data in CASES:
    torch.manual_seed(42)
    gen = (lambda m: torch.randn(m, d)) if data == "uniform" else (lambda m: gmm(m, d, 0.0))
    X = gen(n).cuda()
    Q = None
    if nq:
        torch.manual_seed(7)
        Q = gen(nq).cuda()
    a, b = (Q, X) if nq else (X, None)
    t1, (c1, i1), info1 = run(a, b, k, True)
    t0, (c0, i0), info0 = run(a, b, k, False)
    print(json.dumps({"n_db": n, "n_q": nq or n, "D": d, "k": k, "data": data, "threshold_scan_sec": round(t1, 4),
                      "list_kernel_sec": round(t0, 4), "speedup": round(t0 / t1, 3), "equal": bool(torch.equal(c1, c0) and torch.equal(i1, i0)),
                      "threshold_scan": info1, "list_kernel": info0}), flush=True)
    del X, Q, c1, i1, c0, i0
    torch.cuda.empty_cache()
