"""CPU model of the rounds of the scheduled UMAP gradient launch: how many 16-item rounds a wavefront (16 rows) runs when the
rows of a 64-row workgroup are taken in order and when they are dealt by active count (csrc/tdr_umap_sched.hip, geom bit 6).

A row's items in one slice = its fired edges with column in the slice + its share of 5 negatives per fired edge (binomial
split over the slices).  Firing counts come from a real graph: UMAP affinity of a Gaussian mixture through the CPU oracle
(oracle/ref_torch.py), per-edge epochs_per_sample, the recurrence of umap.py:243-247 for a few hundred iterations.

    python tools/round_model.py [N] [iterations]      (CPU only; N = 20000 takes ~1 min)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle.ref_torch as R
from tests.conftest import gmm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
S, k, max_iter = 2, 30, 1000
X = gmm(n, 32, 2.0, seed=3)
C, I = R.knn_chunked(X, k)
P = R.umap_affinity(C, k)[-1] if isinstance(R.umap_affinity(C, k), tuple) else R.umap_affinity(C, k)
vals, idx = R.symmetrize_sparse(P, I.long())
vals, idx = vals.numpy(), idx.numpy()
mask = idx >= 0
w = vals[mask]
rows = np.repeat(np.arange(n), mask.sum(1))
cols = idx[mask]
eps_per = np.where(w > 0, w.max() / np.maximum(w, 1e-30), np.inf).astype(np.float32)
keep = eps_per <= max_iter          # umap.py:215-234: edges that would fire less than once are dropped
rows, cols, eps_per = rows[keep], cols[keep], eps_per[keep]
nxt = eps_per.copy()
rng = np.random.default_rng(0)
own_slice = cols * S // n
# "local": the loop in cluster order -- practically every fired edge stays in the row's own slice
stats = {"plain": [], "dealt": [], "work": [], "plain_local": [], "dealt_local": [], "dealt_pool256": [], "dealt_round8": [],
         "dealt_exact_key": []}
row_slice = np.arange(n) * S // n
for t in range(iters):
    act = nxt <= t + 1
    nxt[act] += eps_per[act]
    if t < iters // 2:
        continue                       # let the counters spread out first
    fired = np.bincount(rows[act], minlength=n)
    n_use = np.minimum(5 * fired, 5 * k)
    for s in range(S):
        npos = np.bincount(rows[act & (own_slice == s)], minlength=n)
        nneg = rng.binomial(n_use, 1.0 / S) if s == 0 else n_use - nneg0
        if s == 0:
            nneg0 = nneg
        rounds = -(-(npos + nneg) // 16)
        m = n // 64 * 64
        blocks = rounds[:m].reshape(-1, 64)
        key = fired[:m].reshape(-1, 64)
        plain = blocks.reshape(-1, 4, 16).max(2)
        order = np.argsort(key, axis=1, kind="stable")
        dealt = np.take_along_axis(blocks, order, 1).reshape(-1, 4, 16).max(2)
        stats["plain"].append(plain.mean())
        stats["dealt"].append(dealt.mean())
        stats["work"].append(((npos + nneg)[:m] / 16.0).mean())
        # variants (what a next step could buy): a pool of 256 rows, rounds of 8 items (in units of 16), the exact item count as key
        m4 = n // 256 * 256
        it = (npos + nneg)
        o4 = np.argsort(fired[:m4].reshape(-1, 256), axis=1, kind="stable")
        stats["dealt_pool256"].append(np.take_along_axis(rounds[:m4].reshape(-1, 256), o4, 1).reshape(-1, 16, 16).max(2).mean())
        r8 = (-(-it // 8))[:m].reshape(-1, 64)
        stats["dealt_round8"].append(np.take_along_axis(r8, order, 1).reshape(-1, 4, 16).max(2).mean() / 2.0)
        oe = np.argsort(it[:m].reshape(-1, 64), axis=1, kind="stable")
        stats["dealt_exact_key"].append(np.take_along_axis(blocks, oe, 1).reshape(-1, 4, 16).max(2).mean())
        loc = -(-(np.where(row_slice == s, fired, 0) + nneg) // 16)[:m].reshape(-1, 64)
        stats["plain_local"].append(loc.reshape(-1, 4, 16).max(2).mean())
        stats["dealt_local"].append(np.take_along_axis(loc, order, 1).reshape(-1, 4, 16).max(2).mean())
out = {k_: float(np.mean(v)) for k_, v in stats.items()}
out.update(n=n, edges=int(rows.size), mean_fired_per_row=float(np.mean(fired)), std_fired_per_row=float(np.std(fired)))
print(json.dumps(out))
