"""GPU probe: where the time of the schedule build (umap_sched_build_kernel, one window of 32 iterations at N = 1M) goes.
Each variant is a scratch build of the library with parts of the kernel switched off (tools/build_ablate.sh; bit 0 stop after
phase 1, bit 1 no counting atomics, bit 2 no row records, bit 3 no list stores, bit 4 no phase 2), timed in its own process.

    bash tools/build_ablate.sh "1 3 4 8 16"     # in the build container
    gpurun -- 'python tools/sched_build_ablate.py'
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from torchdr_amd import _lib

    v = int(sys.argv[2])
    if v:
        _lib.LIB_PATH = os.path.join(ROOT, "tools", "scratch", f"libtdr_ab{v}.so")
    import torch

    from tests.conftest import gmm
    from tests.test_umap_sched_gpu import Sched, layout, prepare
    from torchdr_amd.affinity import UMAPAffinity
    from torchdr_amd.distance.base import ClusterIndex, PackedPoints

    n = 1_000_000
    X = gmm(n, 128, 2.0).cuda()
    csr = UMAPAffinity(n_neighbors=30, max_iter=100)(X, return_csr=True)
    ci = ClusterIndex(PackedPoints(X))       # production numbering: cluster-sorted order
    perm, inv = ci.perm, ci.inv
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum((csr.rowptr[1:] - csr.rowptr[:-1])[perm.long()], 0, out=rowptr[1:])
    cols, vals = torch.empty_like(csr.cols), torch.empty_like(csr.vals)
    _lib.check(_lib.lib().tdr_csr_permute_f32(_lib.ptr(csr.rowptr), _lib.ptr(csr.cols), _lib.ptr(csr.vals), n, _lib.ptr(perm), _lib.ptr(inv),
                                              _lib.ptr(rowptr), _lib.ptr(cols), _lib.ptr(vals), _lib.stream_ptr()), "permute")
    del X, ci
    eps_per, _ = prepare(vals, 1000)
    cols, eps_per = layout(rowptr, cols, eps_per)
    nxt0 = eps_per.clone()
    sc = Sched(rowptr, cols, eps_per, n, 32, 2)
    nxt = nxt0.clone()
    for t0 in (0, 32, 64):
        sc.build(nxt, t0, 32)
    snap = nxt.clone()
    ts = []
    for _ in range(7):
        nxt.copy_(snap)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.build(nxt, 96, 32)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(json.dumps({"ablate": v, "build_ms": sum(ts[:4]) / 4}), flush=True)
else:
    for v in [0] + [int(x) for x in (sys.argv[1:] or ["1", "3", "4", "8", "16"])]:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(v)], capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else json.dumps({"ablate": v, "error": out.stderr[-400:]}), flush=True)
