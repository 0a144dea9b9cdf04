"""GPU probe: the pool gradient launch on a SHARD of the N-point graph (rows [c0, c0 + n / W) of the cluster-sorted order), per
block-split form (geom 17 = one workgroup per 1024-row block, 18 / 20 / 24 = 2 / 4 / 8 workgroups with one row per lane).

    python tools/pool_split_perf.py [N] [D] [worlds, e.g. 1,2,4,8]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tests.conftest import gmm
from tests.test_umap_pool_gpu import pool_grad
from tests.test_umap_sched_gpu import GroupSched, Sched, layout, prepare
from torchdr_amd.affinity import UMAPAffinity
from torchdr_amd.distance.base import ClusterIndex, PackedPoints

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
worlds = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
X = gmm(n, d, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
perm = ClusterIndex(PackedPoints(X)).perm.to(torch.int64)
inv = torch.empty(n, dtype=torch.int64, device="cuda")
inv[perm] = torch.arange(n, device="cuda")
deg = (csr.rowptr[1:] - csr.rowptr[:-1])[perm]
rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
rowptr[1:] = deg.cumsum(0)
erow = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
src = csr.rowptr[perm][erow] + (torch.arange(erow.numel(), device="cuda") - rowptr[erow])
cols = inv[csr.cols[src].to(torch.int64)].to(torch.int32).contiguous()
vals = csr.vals[src].contiguous()
del perm, inv, deg, erow, src, X, csr
eps_per, _ = prepare(vals, 1000)
cols, eps_per = layout(rowptr, cols, eps_per)
Z = (torch.randn(n, 2, device="cuda") * 5).contiguous()


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


for W in worlds:
    rows = n // W
    c0 = (W // 2) * rows
    c1 = c0 + rows
    e0, e1 = int(rowptr[c0]), int(rowptr[c1])
    sc = Sched((rowptr[c0:c1 + 1] - rowptr[c0]).contiguous(), cols[e0:e1].contiguous(), eps_per[e0:e1].contiguous(), n, 32, 1, row0=c0)
    nxt = eps_per[e0:e1].clone()
    for t0 in (0, 32, 64, 96):
        sc.build(nxt, t0, 32)
    out = {"n": n, "d": d, "world": W, "rows": rows, "blocks": (rows + 1023) // 1024}
    ref = None
    for geom in (17, 18, 20, 24, 0):
        it = [0]

        def step():
            t = it[0] % 32
            it[0] += 1
            return pool_grad(sc, Z, t, 96 + t, 1.577, 0.895, 150, 1234, geom=geom)

        g = pool_grad(sc, Z, 5, 101, 1.577, 0.895, 150, 1234, geom=geom)
        if ref is None:
            ref = g
        assert torch.equal(g, ref), geom
        out[f"geom{geom}_ms"] = round(timed(step, 32), 4)
    # the grouped schedule build of one window of 32 iterations on the same shard (what the loop runs every 32 iterations)
    gs = GroupSched(sc.rowptr, sc.cols, sc.eps_per, n, 32, 1, row0=c0)
    ng = gs.to_group(eps_per[e0:e1].contiguous())
    for t0 in (0, 32, 64):
        gs.build(ng, t0, 32)
    snap = ng.clone()

    def rebuild():
        ng.copy_(snap)
        gs.build(ng, 96, 32)

    out["group_build_ms"] = round(timed(rebuild, 7) - timed(lambda: ng.copy_(snap), 7), 4)
    out["list_entries_per_window"] = int(gs.blk_base[-1].item())
    print(json.dumps(out), flush=True)
    del sc, nxt, gs, ng, snap
