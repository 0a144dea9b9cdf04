"""COSNE per-iteration timing on one MI355X (all-pairs float64 kernel dominates)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.conftest import gmm  # noqa: E402
import torchdr_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
X = gmm(n, 50, 2.0, seed=1).cuda()
for it in (5, 5, iters):          # the first fit pays the one-time warm-up
    m = torchdr_amd.COSNE(perplexity=30, max_iter=it, lr=0.05, random_state=0, check_interval=10 ** 9)
    torch.cuda.synchronize()
    t = time.perf_counter()
    m.fit_transform(X)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    if it == 5:
        t5 = dt
per_iter = (dt - t5) / (iters - 5)
print(f"COSNE N={n}: {per_iter * 1e3:.2f} ms / iteration = {n * n / per_iter / 1e9:.1f} G pairs/s (float64); "
      f"fit with {iters} iterations {dt:.2f} s")
