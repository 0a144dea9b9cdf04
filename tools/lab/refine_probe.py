"""Lab: does the index refinement cost where it cannot help?  Mixtures whose groups have DIFFERENT widths (radii vary by the data's
nature, not because of strays), a mixture with 0.5 % far outliers, one wide Gaussian: kNN (k = 30) with and without REFINE_INDEX."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from torchdr_amd import config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

n, d = 1_000_000, 64
g = torch.Generator().manual_seed(2)


def mix(sigmas, scale=2.0, groups=1000):
    c = torch.randn(groups, d, generator=g) * scale
    lab = torch.arange(n) % groups
    sg = torch.tensor(sigmas)[lab % len(sigmas)]
    return c[lab] + sg[:, None] * torch.randn(n, d, generator=g)


cases = {
    "widths 0.25 / 0.5 / 1.5": lambda: mix([0.25, 0.5, 1.5]),
    "widths 0.3 .. 1.2 (8 values)": lambda: mix([0.3, 0.4, 0.5, 0.6, 0.7, 0.9, 1.0, 1.2]),
    "equal widths + 0.5 % far outliers": lambda: torch.cat([mix([0.5])[: n - n // 200], torch.randn(n // 200, d, generator=g) * 6.0]),
    "one wide Gaussian": lambda: torch.randn(n, d, generator=g) * 2.0,
}
for name, make in cases.items():
    X = make().float().cuda().contiguous()
    rec = {"case": name}
    ref = None
    # the two modes ALTERNATE (no_refine, refine, no_refine, refine ...: the first searches after empty_cache() pay for fresh
    # allocations whichever mode they run in -- an earlier version of this script charged that to the mode it ran first)
    for label, opts in (("no_refine", {"REFINE_INDEX": False}), ("refine", {}), ("no_refine", {"REFINE_INDEX": False}), ("refine", {})):
        with config.options(**opts):
            best = rec.get(label, {}).get("ms", 1e9) / 1e3
            for _ in range(2):
                for k_ in ("index_refined", "index_radii", "pilot_tau", "predicted_share", "lists"):
                    dbase.LAST_KNN.pop(k_, None)
                X2 = X.clone()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                C, I = pairwise_distances(X2, metric="sqeuclidean", k=30, exclude_diag=True, return_indices=True)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            L = dbase.LAST_KNN
            rec[label] = {"ms": round(best * 1e3, 1), "path": L.get("path"), "refined": L.get("index_refined"), "radii": L.get("index_radii"), "share": L.get("predicted_share"),
                          "lists": L.get("lists"), "tau": (L.get("pilot_tau") or [None])[0]}
            if ref is None:
                ref = (C, I)
            else:
                rec["same_rows"] = rec.get("same_rows", True) and bool(torch.equal(ref[0], C) and torch.equal(ref[1], I))
    print(json.dumps(rec), flush=True)
    del X
    torch.cuda.empty_cache()
