"""Manhattan kNN timing on one MI355X: tile pass, exact-order re-evaluation, total (N x D x k from argv)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.conftest import gmm  # noqa: E402
from torchdr_amd.distance import base as dbase  # noqa: E402
from torchdr_amd.distance import pairwise_distances  # noqa: E402

n, d, k = (int(a) for a in (sys.argv[1:4] or (100_000, 128, 30)))
X = gmm(n, d, 2.0, seed=42).cuda()


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


t_all, (C, I) = timed(lambda: pairwise_distances(X, metric="manhattan", k=k, exclude_diag=True, return_indices=True))
flagged = dbase.LAST_KNN["flagged"]
t_tile, _ = timed(lambda: dbase._knn_general(X, X, k + 8, "manhattan", True))
t_blk, _ = timed(lambda: dbase._l1_block(X[:4096], X[:65536]))
pairs = 4096 * min(65536, n) * d
print(f"N={n} D={d} k={k}: total {t_all * 1e3:.1f} ms (tile pass {t_tile * 1e3:.1f} ms, flagged rows {flagged}); "
      f"L1 block 4096x{min(65536, n)}: {t_blk * 1e3:.2f} ms = {pairs / t_blk / 1e12:.2f} T pair-features/s")
