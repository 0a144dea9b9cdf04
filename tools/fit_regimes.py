"""Whole-fit robustness pass at N = 1M over the data regimes of the quality gates (tests/conftest.regime_data: separated blobs,
overlapping blobs, swiss roll in 50 dimensions, heavy-tailed cluster sizes) plus exact duplicates and integer-valued features:
wall time, finiteness, the longest symmetrised row, neighbourhood preservation on a 20k subsample.

    python tools/fit_regimes.py [N] > profiles/r06_fit_regimes.jsonl
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import torchdr_amd
from tests.conftest import regime_data
from torchdr_amd.distance import base as dbase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for name in ("gmm2", "overlap", "swiss", "heavytail", "duplicates", "integers"):
    if name == "duplicates":
        X, _ = regime_data("gmm2", n, seed=5)
        X[::100] = X[1::100][: X[::100].shape[0]]          # 1 % exact duplicates
    elif name == "integers":
        X, _ = regime_data("gmm2", n, seed=6)
        X = (X * 2).round()                                # ties everywhere
    else:
        X, _ = regime_data(name, n)
    X = X.float().cuda().contiguous()
    rec = {"regime": name, "n": int(X.shape[0]), "d": int(X.shape[1])}
    ts = []
    for r in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = torchdr_amd.UMAP(n_neighbors=30, max_iter=1000, random_state=r)
        keep = {}
        orig = m.clear_memory

        def grab(m=m, keep=keep, orig=orig):
            rp = m._csr.rowptr
            keep["max_deg"] = int((rp[1:] - rp[:-1]).max())
            keep["nnz"] = int(m._csr.nnz)
            orig()

        m.clear_memory = grab
        Z = m.fit_transform(X)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    rec.update(fit_ms=round(min(ts), 1), finite=bool(torch.isfinite(Z).all()), rows_out=int(Z.shape[0]), max_degree=keep.get("max_deg"),
               nnz=keep.get("nnz"), knn_path=dbase.LAST_KNN.get("path"), knn_lists=dbase.LAST_KNN.get("lists"))
    sub = torch.randperm(X.shape[0], generator=torch.Generator().manual_seed(0))[:20000].cuda()
    from torchdr_amd.eval import neighborhood_preservation

    rec["neighbourhood_preservation_20k_subsample"] = round(float(neighborhood_preservation(X[sub], Z[sub].float(), K=15)), 4)
    print(json.dumps(rec), flush=True)
    del X, Z, m
    torch.cuda.empty_cache()
