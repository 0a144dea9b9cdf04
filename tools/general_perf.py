"""GPU probe: the general-D kNN path (library GEMM + running top-k) on an MNIST-shaped problem."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import pairwise_distances

for n, d, k in ((70000, 784, 15), (200000, 512, 30)):
    X = gmm(n, d, 2.0).cuda()
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        C, I = pairwise_distances(X, metric="sqeuclidean", k=k, exclude_diag=True, return_indices=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"n": n, "d": d, "k": k, "sec": round(dt, 3), "tflops": round(2.0 * n * n * d / dt / 1e12, 1)}), flush=True)
