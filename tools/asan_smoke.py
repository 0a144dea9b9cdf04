"""Smoke-size exercise of the ASAN build (tools/build_asan.sh): UMAP through the scheduled loop (1 and 2 L2 slices, a
partial last window) and the two-stage exact kNN search, plain and cluster-pruned, against the one-stage kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torchdr_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "libtdr_asan.so")
import torch

import torchdr_amd
from tests.conftest import gmm
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances
from torchdr_amd.neighbor_embedding import umap as U

X = gmm(6000, 24, 2.0, seed=1).cuda()
dbase.SCREEN_MODE = "0"
Ce, Ie = pairwise_distances(X, metric="sqeuclidean", k=12, exclude_diag=True, return_indices=True)
for prune in ("0", "force"):
    dbase.SCREEN_MODE, dbase.PRUNE_MODE = "force", prune
    C, I = pairwise_distances(X, metric="sqeuclidean", k=12, exclude_diag=True, return_indices=True)
    assert torch.equal(C, Ce) and torch.equal(I, Ie), prune
    print("two-stage search, prune =", prune, "ok:", dbase.LAST_KNN["path"], flush=True)
for slices in (1, 2):
    U.SCHED_SLICES = slices
    Z = torchdr_amd.UMAP(n_neighbors=12, max_iter=70, random_state=0).fit_transform(X)
    assert bool(torch.isfinite(Z).all())
    print("scheduled UMAP loop,", slices, "slice(s) ok", flush=True)
torch.cuda.synchronize()
print("ASAN smoke finished without a report")
