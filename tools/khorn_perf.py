"""TSNEkhorn at BASELINE config C5's size (N = 200k, D = 64): ms per Sinkhorn pass on the embedding, per adjoint mat-vec,
per force scan (plain and unrolled) and per training step.  `python tools/khorn_perf.py [n]`"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests.conftest import gmm  # noqa: E402
import torchdr_amd as t  # noqa: E402
from torchdr_amd.affinity.entropic import sinkhorn_student_adjoint, sinkhorn_student_dual  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
X = gmm(n, 64, 2.0).cuda()
out = {"n": n}
Z = torch.randn(n, 2, device="cuda") * 5
out["sinkhorn_5_passes_ms"] = timed(lambda: sinkhorn_student_dual(Z, None, 5, 0.0, True))
rec = []
sinkhorn_student_dual(Z, None, 5, 0.0, True, record=rec)
g = torch.full((n,), -2.0 / n, device="cuda")
out["adjoint_5_matvecs_ms"] = timed(lambda: sinkhorn_student_adjoint(Z, rec, g, True))
for unroll in (False, True):
    t.TSNEkhorn(perplexity=30, max_iter=2, max_iter_affinity_in=3, init="normal", init_scaling=1.0, lr=1.0, optimizer="SGD",
                optimizer_kwargs=None, min_grad_norm=0.0, random_state=0, unrolling=unroll).fit_transform(X)    # warm-up
    m = t.TSNEkhorn(perplexity=30, max_iter=4, max_iter_affinity_in=3, init="normal", init_scaling=1.0, lr=1.0, optimizer="SGD",
                    optimizer_kwargs=None, min_grad_norm=0.0, random_state=0, unrolling=unroll)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.fit_transform(X)
    torch.cuda.synchronize()
    a = time.perf_counter() - t0
    m = t.TSNEkhorn(perplexity=30, max_iter=12, max_iter_affinity_in=3, init="normal", init_scaling=1.0, lr=1.0, optimizer="SGD",
                    optimizer_kwargs=None, min_grad_norm=0.0, random_state=0, unrolling=unroll)
    t0 = time.perf_counter()
    m.fit_transform(X)
    torch.cuda.synchronize()
    b = time.perf_counter() - t0
    out["step_ms_unrolled" if unroll else "step_ms"] = (b - a) / 8 * 1e3
print(json.dumps(out))
