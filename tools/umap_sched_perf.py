"""GPU probe: scheduled UMAP loop kernels on the N = 1M graph -- schedule build time, gradient passes per lane
geometry / slice count, against the per-step kernel (tools/umap_perf.py measures that one alone).

    python tools/umap_sched_perf.py [N] [geoms, e.g. 0,1,2] [slices, e.g. 1,2,4]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from tests.test_umap_sched_gpu import Sched, layout, prepare
from torchdr_amd import _lib
from torchdr_amd.affinity import UMAPAffinity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
geoms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3").split(",")]
slices = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2").split(",")]
ITERS = int(os.environ.get("ITERS", "32"))
X = gmm(n, 128, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
del X
eps_per, nxt0 = prepare(csr.vals, 1000)
cols = csr.cols
if os.environ.get("LAYOUT", "1") == "1":
    cols, eps_per = layout(csr.rowptr, csr.cols, eps_per)
    nxt0 = eps_per.clone()
Z = (torch.randn(n, 2, device="cuda") * 5).contiguous()
print(json.dumps({"n": n, "nnz": csr.nnz, "mean_deg": csr.nnz / n}), flush=True)


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return sum(ts[: max(1, len(ts) // 2)]) / max(1, len(ts) // 2)


for S in slices:
    sc = Sched(csr.rowptr, cols, eps_per, n, 32, S)
    nxt = nxt0.clone()
    for t0 in (0, 32, 64):          # a steady-state window
        sc.build(nxt, t0, 32)
    snap = nxt.clone()

    def rebuild():
        nxt.copy_(snap)
        sc.build(nxt, 96, 32)

    ms_build = timed(rebuild, 5) - timed(lambda: nxt.copy_(snap), 5)
    cap = int(sc.blk_base[-1].item())
    used = int(sc.records(32)[1].sum())
    print(json.dumps({"slices": S, "build_ms": ms_build, "build_ms_per_iter": ms_build / 32, "list_capacity": cap,
                      "list_used": used}), flush=True)
    for geom in geoms:
        it = [0]

        def step():
            t = it[0] % 32
            it[0] += 1
            sc.grad(Z, t, 96 + t, 1.577, 0.895, 150, neg=None, seed=1234, geom=geom)

        print(json.dumps({"slices": S, "geom": geom, "grad_ms": timed(step, ITERS)}), flush=True)
    # no negatives: the positive items alone
    def step_pos():
        sc.grad(Z, 3, 99, 1.577, 0.895, 150, neg=None, seed=1234, geom=0, neg_rate=0)

    print(json.dumps({"slices": S, "variant": "no_negatives", "grad_ms": timed(step_pos, 16)}), flush=True)
    del sc
