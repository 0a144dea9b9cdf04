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
ABL_GEOM = int(os.environ.get("ABL_GEOM", "0"))
geoms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5,6").split(",")]
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
erow = torch.repeat_interleave(torch.arange(n, device="cuda"), rowptr[1:] - rowptr[:-1])
far = (cols.to(torch.int64) - erow).abs()
same_blk = (cols.to(torch.int64) // 512) == (erow // 512)
hot = eps_per < 8.0       # edges that fire at least every 8th iteration: most of the firings
print(json.dumps({"n": n, "nnz": int(cols.numel()), "edges_same_512_block": float(same_blk.float().mean()),
                  "hot_edges_same_512_block": float(same_blk[hot].float().mean()), "hot_share": float(hot.float().mean()),
                  "hot_within_1024": float((far[hot] < 1024).float().mean()), "hot_within_4096": float((far[hot] < 4096).float().mean())}), flush=True)
del erow, far, same_blk, hot


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


ONLY = os.environ.get("PERF_ONLY", "0") == "1"     # counters: the production launch of the first geometry only
for S in (1,) if ONLY else (1, 2):
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

            if ONLY:
                print(json.dumps({"pool_geom": geom, "grad_ms": timed(step, 32)}), flush=True)
                break
            print(json.dumps({"pool_geom": geom, "grad_ms": timed(step, 32), "no_negatives_ms": timed(step_pos, 16)}), flush=True)
        if ONLY:
            break
        # ablations through the instrumented instance (1 no pool staging, 2 no attraction, 4 no negatives, 8 rows not sorted,
        # 16 no gathers, 32 no list reads)
        L = _lib.lib()
        gbuf = torch.empty((n, 2), device="cuda")

        def dbg(ab, times=None, tl=3):
            _lib.check(L.tdr_umap_pool_grad_debug_f32(_lib.ptr(Z), 2, n, 0, n, _lib.ptr(sc.list), _lib.ptr(sc.hdr), tl, 1.577, 0.895, 96 + tl, 5, 150,
                                                      1234, _lib.ptr(gbuf), ABL_GEOM, ab, _lib.ptr(times), _lib.stream_ptr()), "pool_grad_debug")

        for ab, what in ((0, "instrumented instance, nothing off"), (1, "no staging"), (2, "no attraction"), (4, "no negatives"), (8, "unsorted"),
                         (6, "fixed part only"), (7, "fixed part without staging"), (5, "attraction only, no staging"),
                         (5 + 16, "attraction only, no staging, no gathers"), (5 + 32, "attraction only, no staging, no list reads"),
                         (5 + 48, "attraction only, no staging, no list reads, no gathers")):
            it2 = [0]

            def step_ab():
                it2[0] += 1
                dbg(ab, tl=it2[0] % 32)

            print(json.dumps({"geom": ABL_GEOM, "ablate": ab, "what": what, "grad_ms": timed(step_ab, 32)}), flush=True)
        # phase time stamps of every wavefront (shader clock; the counters of different XCDs are not aligned: durations only, and
        # "since the block's first stamp")
        NWV = {4: 4, 6: 16}.get(ABL_GEOM, 8)
        for ab in (0, 4, 2):
            times = torch.zeros((4096 * NWV, 8), dtype=torch.int64, device="cuda")
            dbg(ab, times)
            torch.cuda.synchronize()
            T = times.cpu().view(4096, NWV, 8)
            T = T[T[:, 0, 0] != 0].double()
            b0 = T[:, :, 0].min(1, keepdim=True).values
            q = lambda x: [round(float(v)) for v in torch.quantile(x.flatten(), torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64))]
            per_wave = {f"wave{w}": {"pass0": q(T[:, w, 4] - T[:, w, 3]), "pass1": q(T[:, w, 5] - T[:, w, 4]), "end_since_block_start": q(T[:, w, 5] - b0[:, 0])}
                        for w in range(NWV)}
            print(json.dumps({"geom": ABL_GEOM, "ablate": ab, "blocks": int(T.shape[0]), "cycles_p10_p50_p90": {
                "first_barrier_wait": q(T[:, :, 2] - T[:, :, 1]), "sort": q(T[:, :, 3] - T[:, :, 2]),
                "block_duration": q(T[:, :, 5].max(1).values - b0[:, 0]), **per_wave}}), flush=True)
    else:
        def step():
            t = it[0] % 32
            it[0] += 1
            sc.grad(Z, t, 96 + t, 1.577, 0.895, 150, neg=None, seed=1234, geom=16)

        print(json.dumps({"iid_slices": S, "geom": 16, "grad_ms": timed(step, 32)}), flush=True)
    del sc
