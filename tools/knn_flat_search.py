"""One whole unpruned search through the threshold scan, for a kernel trace: argv = n set(k) [flat 0|1]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "uniform"
flat = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
torch.manual_seed(42)
if which == "uniform":
    X, k = torch.randn(n, 128).cuda(), 15
elif which == "structureless":
    X, k = gmm(n, 128, 0.0).cuda(), 30
else:
    X, k = gmm(n, 128, 2.0).cuda(), 30
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with config.options(PRUNE_MODE="0" if which == "mixture" else "auto", FLAT_SCAN=flat):
        C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
    torch.cuda.synchronize()
    print({"sec": time.perf_counter() - t0, **{k_: v for k_, v in dbase.LAST_KNN.items() if k_ in ("path", "tier", "flat_terms", "flagged")}}, flush=True)
