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
geoms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3").split(",")]   # + 16: joint slice launch
slices = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2").split(",")]
ITERS = int(os.environ.get("ITERS", "32"))
X = gmm(n, 128, 2.0).cuda()
csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
rowptr_, cols_, vals_ = csr.rowptr, csr.cols, csr.vals
if os.environ.get("RELABEL", "0") == "1":
    # points renumbered in the cluster-sorted order of the kNN stage's index: a row's neighbours get nearby numbers
    from torchdr_amd.distance.base import ClusterIndex, PackedPoints

    perm = ClusterIndex(PackedPoints(X)).perm.to(torch.int64)    # the production numbering (members of a cluster by ascending row)
    assert perm.numel() == n
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    inv[perm] = torch.arange(n, device="cuda")
    deg = (rowptr_[1:] - rowptr_[:-1])[perm]
    new_rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    new_rowptr[1:] = deg.cumsum(0)
    erow = torch.repeat_interleave(torch.arange(n, device="cuda"), deg)
    src = rowptr_[perm][erow] + (torch.arange(erow.numel(), device="cuda") - new_rowptr[erow])
    cols_ = inv[cols_[src].to(torch.int64)].to(torch.int32).contiguous()
    vals_ = vals_[src].contiguous()
    rowptr_ = new_rowptr
    del perm, inv, deg, erow, src
    far = (cols_.to(torch.int64) - torch.repeat_interleave(torch.arange(n, device="cuda"), rowptr_[1:] - rowptr_[:-1])).abs()
    print(json.dumps({"relabel": True, "edges_within_1024": float((far < 1024).float().mean()),
                      "edges_within_4096": float((far < 4096).float().mean())}), flush=True)
    del far


class _G:
    pass


csr_ = _G()
csr_.rowptr, csr_.cols, csr_.vals, csr_.nnz = rowptr_, cols_, vals_, csr.nnz
csr = csr_
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

# ---- host cost of the loop object: wall time of the tdr_umap_loop_run CALL (no synchronisation) per iteration --------
import ctypes
import time

L = _lib.lib()
S = slices[0]
sc = Sched(csr.rowptr, cols, eps_per, n, 32, S)
T = 320
lr = torch.linspace(1.0, 0.0, T + 1)[:T].contiguous().cuda()
nxt = nxt0.clone()
grad, norm2 = torch.empty((n, 2), device="cuda"), torch.zeros(64, device="cuda")
flag, scratch = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(16, dtype=torch.int32, device="cuda")
d = _lib.UmapLoopDesc()
d.Z, d.nc, d.n_total, d.row0, d.n_rows = _lib.ptr(Z), 2, n, 0, n
d.rowptr, d.cols, d.eps_per, d.next = _lib.ptr(csr.rowptr), _lib.ptr(cols), _lib.ptr(eps_per), _lib.ptr(nxt)
d.blk_base, d.list, d.hdr, d.err = _lib.ptr(sc.blk_base), _lib.ptr(sc.list), _lib.ptr(sc.hdr), _lib.ptr(sc.err)
d.acc, d.grad, d.mom_buf = _lib.ptr(sc.acc), _lib.ptr(grad), None
d.a, d.b, d.neg_rate, d.n_negatives, d.seed = 1.577, 0.895, 5, 150, 77
d.exag, d.rep, d.eps, d.n_slices, d.block_iters = 1.0, 1.0, 1e-3, S, 32
d.lr_table, d.max_iter, d.momentum, d.first_iter, d.check_interval = _lib.ptr(lr), T, 0.0, 0, 50
d.norm2, d.snap, d.nan_flag, d.scratch, d.gather, d.gather_ctx, d.geom = _lib.ptr(norm2), None, _lib.ptr(flag), _lib.ptr(scratch), None, None, 0
h = ctypes.c_void_p()
_lib.check(L.tdr_umap_loop_create(ctypes.byref(h), ctypes.byref(d)), "create")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for mode, name in ((1, "graph"), (0, "launches")):
        _lib.check(L.tdr_umap_loop_run(h, 0, 32, mode, _lib.stream_ptr()), "warm")   # capture / first use
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(L.tdr_umap_loop_run(h, 32, 288, mode, _lib.stream_ptr()), "run")
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(json.dumps({"loop_object": name, "host_us_per_iteration": t_host / 288 * 1e6, "gpu_ms_per_iteration": t_all / 288 * 1e3}),
              flush=True)
L.tdr_umap_loop_destroy(h)
