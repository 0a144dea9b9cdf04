"""One real-data-shaped pass through the kNN chooser: exact kNN (k = 30) at N = 1M x 128 on data the tuning never
saw -- heavy-tailed cluster sizes, a 2-d manifold (swiss roll) rotated into 128 dimensions, 1 % exact duplicates, integer-valued
features (masses of exactly tied distances), the benchmark's own mixture as the control -- recording the path / tier the dispatcher
took, the flagged rows, the time, and a bit-for-bit check of sampled rows against the CPU oracle (oracle/knn_oracle.c: the
reference's arithmetic, distance/torch.py:82-122).  Also the UMAP fit on each (both negative samplers) with the embedding's
neighbourhood preservation.

    python tools/knn_regimes.py [N] > profiles/r06_knn_regimes.jsonl
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import oracle
import torchdr_amd
from oracle.ref_torch import canonical_rows
from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances
from torchdr_amd.eval import neighborhood_preservation

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 128, 30


def data(name):
    g = torch.Generator().manual_seed(7)
    if name == "mixture (control)":
        nc = 1000
        c = torch.randn(nc, d, generator=g) * 2.0
        return c[torch.arange(n) % nc] + 0.5 * torch.randn(n, d, generator=g)
    if name == "heavy-tailed cluster sizes":
        nc = 3000
        w = 1.0 / torch.arange(1, nc + 1, dtype=torch.float64) ** 1.1
        lab = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
        c = torch.randn(nc, d, generator=g) * 2.0
        return c[lab] + 0.5 * torch.randn(n, d, generator=g)
    if name == "swiss roll in 128-d":
        t = 1.5 * torch.pi * (1 + 2 * torch.rand(n, generator=g))
        h = 21 * torch.rand(n, generator=g)
        P = torch.stack([t * torch.cos(t), h, t * torch.sin(t)], 1)
        Q, _ = torch.linalg.qr(torch.randn(d, d, generator=g))
        return P @ Q[:3] + 0.05 * torch.randn(n, d, generator=g)
    if name == "1 % exact duplicates":
        X = data("mixture (control)")
        src = torch.randint(0, n, (n // 100,), generator=g)
        dst = torch.randint(0, n, (n // 100,), generator=g)
        X[dst] = X[src]
        return X
    if name == "integer-valued features":
        return torch.randint(0, 4, (n, d), generator=g).float()
    raise ValueError(name)


torchdr_amd.UMAP(n_neighbors=k, max_iter=40, random_state=0).fit_transform(data("mixture (control)").float().cuda()[:200_000].contiguous())   # warm-up
for name in ("mixture (control)", "heavy-tailed cluster sizes", "swiss roll in 128-d", "1 % exact duplicates", "integer-valued features"):
    Xc = data(name).float().contiguous()
    X = Xc.cuda()
    rec = {"regime": name, "n": n, "d": d, "k": k}
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    L = dbase.LAST_KNN
    rec.update({"knn_sec": best, "path": L.get("path"), "tier": L.get("tier"), "pruned": L.get("pruned"), "tile_bounds": L.get("tile_bounds"),
                "threshold_scan_terms": L.get("flat_terms"), "flagged_rows": L.get("flagged"), "tier_candidates": L.get("tier_candidates")})
    # sampled rows against the CPU oracle (k + 2 without exclusion, the query's own index dropped; ties: canonical order, and
    # only rows whose k-th and (k+1)-th distances differ have an unambiguous index set)
    rows = torch.randperm(n, generator=torch.Generator().manual_seed(2))[:192].sort().values
    Cf, If = oracle.knn(Xc[rows].contiguous(), k + 2, "sqeuclidean", False, Y=Xc)
    # duplicates of the query are legitimate neighbours at distance ~0: drop exactly ONE entry equal to the query's index
    notself = If != rows[:, None].to(If.dtype)
    pick = notself & (notself.long().cumsum(1) <= k + 1)
    Co, Io = Cf[pick].view(-1, k + 1), If[pick].view(-1, k + 1)
    Cg, Ig = C.cpu()[rows], I.cpu()[rows]
    clear = Co[:, k] > Co[:, k - 1]
    _, Io_c = canonical_rows(Co[:, :k], Io[:, :k])
    _, Ig_c = canonical_rows(Cg, Ig)
    rec["oracle_rows"] = int(rows.numel())
    rec["distances_bit_equal"] = bool(torch.equal(Cg, Co[:, :k]))
    rec["rows_with_unambiguous_top_k"] = float(clear.float().mean())
    rec["indices_equal_on_unambiguous_rows"] = bool(torch.equal(Ig_c[clear], Io_c[clear].to(Ig_c.dtype)))
    del C, I
    if name != "integer-valued features":      # (UMAP removes duplicate rows itself; a 4-valued lattice has no neighbourhoods to embed)
        for mode in ("pool", "iid"):
            with config.options(NEGATIVES=mode):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                Z = torchdr_amd.UMAP(n_neighbors=k, max_iter=1000, random_state=0).fit_transform(X)
                torch.cuda.synchronize()
                rec[f"umap_fit_sec_{mode}"] = time.perf_counter() - t0
            sub = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:200_000].cuda()
            rec[f"umap_neighborhood_preservation_K15_200k_subsample_{mode}"] = float(neighborhood_preservation(X[sub], Z[sub], K=15))
            del Z
    print(json.dumps(rec), flush=True)
    del X
    torch.cuda.empty_cache()
