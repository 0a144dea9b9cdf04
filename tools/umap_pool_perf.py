"""GPU probe: the pool-sampled gradient kernel (csrc/tdr_umap_pool.hip) on the N = 1M graph in the production (cluster-sorted)
numbering -- per geometry, with and without negatives -- next to the i.i.d. kernel's joint two-slice launch and the schedule
builds they need (one slice / two slices).

    python tools/umap_pool_perf.py [N] [geoms, e.g. 0,1,2,3,4,5]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from tests.test_umap_pool_gpu import pool_grad
from tests.test_umap_sched_gpu import Sched, layout, prepare
from torchdr_amd import _lib
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.distance.base import ClusterIndex, PackedPoints

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
geoms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5").split(",")]
X = gmm(n, 128, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
rowptr_, cols_, vals_ = csr.rowptr, csr.cols, csr.vals
perm = ClusterIndex(PackedPoints(X)).perm.to(torch.int64)
inv = torch.empty(n, dtype=torch.int64, device="cuda")
inv[perm] = torch.arange(n, device="cuda")
deg = (rowptr_[1:] - rowptr_[:-1])[perm]
rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
rowptr[1:] = deg.cumsum(0)
erow = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
src = rowptr_[perm][erow] + (torch.arange(erow.numel(), device="cuda") - rowptr[erow])
cols = inv[cols_[src].to(torch.int64)].to(torch.int32).contiguous()
vals = vals_[src].contiguous()
del perm, inv, deg, erow, src, X, csr, rowptr_, cols_, vals_
eps_per, _ = prepare(vals, 1000)
cols, eps_per = layout(rowptr, cols, eps_per)
nxt0 = eps_per.clone()
Z = (torch.randn(n, 2, device="cuda") * 5).contiguous()
print(json.dumps({"n": n, "nnz": int(cols.numel())}), flush=True)


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


for S in (1, 2):
    sc = Sched(rowptr, cols, eps_per, n, 32, S)
    nxt = nxt0.clone()
    for t0 in (0, 32, 64):
        sc.build(nxt, t0, 32)
    snap = nxt.clone()

    def rebuild():
        nxt.copy_(snap)
        sc.build(nxt, 96, 32)

    ms_build = timed(rebuild, 5) - timed(lambda: nxt.copy_(snap), 5)
    print(json.dumps({"slices": S, "row_chunk_build_ms": ms_build}), flush=True)
    it = [0]
    if S == 1:
        for geom in geoms:
            def step():
                t = it[0] % 32
                it[0] += 1
                pool_grad(sc, Z, t, 96 + t, 1.577, 0.895, 150, 1234, geom=geom)

            def step_pos():
                pool_grad(sc, Z, 3, 99, 1.577, 0.895, 150, 1234, geom=geom, neg_rate=0)

            print(json.dumps({"pool_geom": geom, "grad_ms": timed(step, 32), "no_negatives_ms": timed(step_pos, 16)}), flush=True)
    else:
        def step():
            t = it[0] % 32
            it[0] += 1
            sc.grad(Z, t, 96 + t, 1.577, 0.895, 150, neg=None, seed=1234, geom=16)

        print(json.dumps({"iid_slices": S, "geom": 16, "grad_ms": timed(step, 32)}), flush=True)
    del sc
