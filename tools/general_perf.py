"""GPU probe: kNN for D > 256 on MNIST-shaped problems -- the K-chunked MFMA scan (tdr_knn_wide_f32) against the
library-GEMM + running top-k form of the same search."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import base as dbase
from torchdr_amd.distance import pairwise_distances

for n, d, k in ((70000, 784, 15), (200000, 512, 30), (20000, 2048, 30)):
    X = gmm(n, d, 2.0).cuda()
    for wide in (True, False):
        dbase.WIDE_SCAN = wide
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps({"n": n, "d": d, "k": k, "path": dbase.LAST_KNN["path"], "sec": round(dt, 3),
                          "tflops": round(2.0 * n * n * d / dt / 1e12, 1)}), flush=True)
    dbase.WIDE_SCAN = True
