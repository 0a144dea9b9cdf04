"""Per-iteration timing of the SNE / InfoTSNE gradient paths (GPU box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import torchdr_amd  # noqa: E402
from bench import gmm  # noqa: E402

for cls, n, kw in ((torchdr_amd.SNE, 50_000, dict(perplexity=30, lr=10.0)), (torchdr_amd.InfoTSNE, 1_000_000, dict(perplexity=30)),
                   (torchdr_amd.InfoTSNE, 100_000, dict(perplexity=30))):
    X = gmm(n, 64, 2.0).cuda()
    for it in (20, 120):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cls(max_iter=it, random_state=0, **kw).fit_transform(X)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it == 20:
            base = dt
    print(f"{cls.__name__} n={n}: {(dt - base) / 100 * 1e3:.2f} ms/iter (fit_transform {it} it: {dt:.2f}s)", flush=True)
