"""Two launches of the threshold-scan kernel over the whole database (N = 1M, D = 128, one term, thresholds = exact k-th
distance + band) for rocprofv3 --pmc passes (tools/pmc_flat.sh).  argv: n terms"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from torchdr_amd import _lib, config
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
terms = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d, k = 128, 30
X = gmm(n, d, 2.0).cuda()
with config.options(PRUNE_MODE="auto", FLAT_SCAN=False):      # the pruned search: no threshold-scan launches of its own
    C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
L = _lib.lib()
P = dbase.PackedPoints(X)
q16, y16, meta = dbase._screen_operands(P, P)
tau = (C[:, -1] + 2.6).contiguous()
cap = 256
buf = torch.empty((n, cap), dtype=torch.int64, device="cuda")
cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
n_tiles = (n + 31) // 32
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.tdr_knn_flat_scan_f32(_lib.ptr(q16), n, 0, _lib.ptr(y16), n, d, terms, 1, 0, n_tiles, 1, _lib.ptr(meta), _lib.ptr(tau),
                                       _lib.ptr(buf), _lib.ptr(cnt), cap, _lib.stream_ptr()), "scan")
    e1.record()
    torch.cuda.synchronize()
    print({"scan_ms": e0.elapsed_time(e1), "mean_appended": float(cnt.float().mean())}, flush=True)
